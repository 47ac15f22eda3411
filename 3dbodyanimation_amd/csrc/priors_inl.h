// priors_inl.h — pose prior, shape prior and temporal residuals, executed by EXTRA workgroups of the
// k_frame_resjac launch (one per tile of 16 frames), so an evaluation sweep needs no second stream, no
// fork/join events and no extra launches (a cross-stream fork/join was measured at ~8 us per sweep).
//
//   pose prior   PosePriorAAAnalytic::Evaluate (include/Sim3BA.h:263-315): L2 r = beta_p x, or the GMM
//                max-mixture residual of ark::GaussianMixture::residual (uses at :280,288)
//   shape prior  ShapePriorL2Analytic::Evaluate (include/Sim3BA.h:331-343): r = beta_s w
//   temporal     Vec3DiffCost (include/MultiFrameBA.h:20-28,121-142): r = lambda (a_f - a_{f+1}) on
//                rootT, rootAA, then joints 1..23
// The GMM whitening  T_k = s (X - mu_k) L_k  ([16 x 69].[69 x 69] per component) is a dense contraction and
// runs on the f64 matrix cores: wave k of the 8-wave workgroup takes component k, 5 column
// tiles x 18 k-steps of v_mfma_f64_16x16x4_f64 each with the 16 frames on the MFMA row index; L_k is stored
// in MFMA B-fragment order at upload (fully coalesced 16-byte loads, six k-steps in flight at a time); the
// 16 pose vectors go through LDS.  |T_k|^2 per frame is a 16-lane butterfly; the component is picked across
// the waves through LDS (first minimum, as the sequential reference loop) and only the winner writes its rows.
#pragma once
#include "bodyfit_device.h"

namespace bodyfit {

typedef __attribute__((ext_vector_type(4))) double prior_d4;
constexpr int kPriorTileF = 16;   // frames per prior workgroup (MFMA M)
constexpr int kPriorNT = 5;       // column tiles of 16 (69 -> 80)
constexpr int kPriorKS = 18;      // k-steps of 4 (69 -> 72)

// Written for the 512-thread (8-wave) workgroups of k_frame_resjac: one mixture component per wave (K <= 8).
__device__ inline void prior_block(const PriorArgs& A, int tile, const double* __restrict__ params, double* sm) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NT = 512;
  const int f0 = tile * kPriorTileF, F = A.F;
  const int npose = kFrameParams, D = npose - 7;
  double* sx = sm;                       // [16][72]
  double* sval = sm + kPriorTileF * 72;  // [8][16]
  double pcost = 0.0;                    // 1/2 sum of squares of the rows this thread writes (folded shared-beta reduction)

  if (A.beta_pose > 0.0 && A.r_prior) {
    if (!A.has_gmm) {
      for (int i = tid; i < kPriorTileF * D; i += NT) {
        const int f = f0 + i / D, c = i % D;
        if (f < F) { const double v = A.beta_pose * params[(size_t)f * npose + 7 + c]; A.r_prior[(size_t)f * D + c] = v; pcost += 0.5 * v * v; }
      }
      if (A.comp && tid < kPriorTileF && f0 + tid < F) A.comp[f0 + tid] = 0;
    } else {
      const DevGmm& g = A.g;
      const int m = lane & 15, kk = lane >> 4;  // MFMA: A[i = m][k = kk], B[k = kk][j = m]
      {
        // 16 x 72 pose values: fixed 3 predicated passes, all loads in flight
        double xv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int idx = tid + u * NT, fr = idx / 72, c = idx % 72;
          xv[u] = (idx < kPriorTileF * 72 && c < D && f0 + fr < F) ? params[(size_t)(f0 + fr) * npose + 7 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int idx = tid + u * NT;
          if (idx < kPriorTileF * 72) sx[idx] = xv[u];
        }
      }
      __syncthreads();
      const int k = wave;                 // one mixture component per wave
      prior_d4 acc[kPriorNT];
      double nlw = 0.0;
#pragma unroll
      for (int nt = 0; nt < kPriorNT; ++nt) acc[nt] = prior_d4{0.0, 0.0, 0.0, 0.0};
      if (k < g.K) {
        nlw = g.neg_log_w[k];
        const double2* Lf = reinterpret_cast<const double2*>(g.prec_frag) + (size_t)k * kPriorKS * 3 * 64 + lane;
#pragma unroll
        for (int third = 0; third < 3; ++third) {   // 6 k-steps of fragments in flight (72 VGPRs): the kernel stays
          double2 b[6][3];                          // under 128 VGPRs so a prior workgroup co-resides with a frame one
          double mu[6];
          // the factor is LOWER triangular (B[r][c] = L[r][c], zero for c > r): column tile nt only meets the k-steps with
          // 4 ks + 3 >= 16 nt, i.e. ks >= 4 nt — 50 of the 90 tile products (and their fragment loads) remain; all the
          // conditions below are compile-time after unrolling
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const int ks = third * 6 + s, r = 4 * ks + kk;
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
              if (ks >= 4 * (2 * pr)) b[s][pr] = Lf[(size_t)(ks * 3 + pr) * 64];
            mu[s] = (r < D) ? g.mean[(size_t)k * D + r] : 0.0;
          }
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const int ks = third * 6 + s, r = 4 * ks + kk;
            const double a = (r < D) ? sx[m * 72 + r] - mu[s] : 0.0;
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][0].x, acc[0], 0, 0, 0);
            if (ks >= 4) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][0].y, acc[1], 0, 0, 0);
            if (ks >= 8) acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][1].x, acc[2], 0, 0, 0);
            if (ks >= 12) acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][1].y, acc[3], 0, 0, 0);
            if (ks >= 16) acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][2].x, acc[4], 0, 0, 0);
          }
        }
        // D layout (f64): column = lane & 15, frame row = (lane >> 4) + 4 q
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double sacc = 0.0;
#pragma unroll
          for (int nt = 0; nt < kPriorNT; ++nt) {
            acc[nt][q] *= g.resid_scale;
            sacc += acc[nt][q] * acc[nt][q];
          }
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) sacc += __shfl_xor(sacc, off, 64);
          if (m == 0) sval[k * kPriorTileF + kk + 4 * q] = sacc + nlw;
        }
      }
      __syncthreads();
      if (k < g.K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = kk + 4 * q, f = f0 + row;
          int best = 0;
          double bv = sval[row];
          for (int k2 = 1; k2 < g.K; ++k2) {
            const double v2 = sval[k2 * kPriorTileF + row];
            if (v2 < bv) { bv = v2; best = k2; }
          }
          if (best == k && f < F) {
            double* o = A.r_prior + (size_t)f * (D + 1);
#pragma unroll
            for (int nt = 0; nt < kPriorNT; ++nt) {
              const int c = 16 * nt + m;
              if (c < D) { const double v = A.beta_pose * acc[nt][q]; o[c] = v; pcost += 0.5 * v * v; }
            }
            if (m == 0) {
              const double v = A.beta_pose * sqrt(nlw);
              o[D] = v; pcost += 0.5 * v * v;
              if (A.comp) A.comp[f] = k;
            }
          }
        }
      }
    }
  }
  if (A.beta_shape > 0.0 && A.r_shape && A.beta) {
    if (A.beta_stride > 0) {
      for (int i = tid; i < kPriorTileF * A.nS; i += NT) {
        const int f = f0 + i / A.nS, c = i % A.nS;
        if (f < F) { const double v = A.beta_shape * A.beta[(size_t)f * A.beta_stride + c]; A.r_shape[(size_t)f * A.nS + c] = v; pcost += 0.5 * v * v; }
      }
    } else if (tile == 0) {
      for (int i = tid; i < A.nS; i += NT) { const double v = A.beta_shape * A.beta[i]; A.r_shape[i] = v; pcost += 0.5 * v * v; }
    }
  }
  if (A.lambda_t > 0.0 && A.r_temporal) {
    const int T = 6 + D;
    for (int i = tid; i < kPriorTileF * T; i += NT) {
      const int f = f0 + i / T, c = i % T;
      if (f < A.n_pairs) {
        const int src = (c < 3) ? (4 + c) : (c < 6 ? (1 + (c - 3)) : (7 + (c - 6)));
        const double v = A.lambda_t * (params[(size_t)f * npose + src] - params[(size_t)(f + 1) * npose + src]);
        A.r_temporal[(size_t)f * T + c] = v; pcost += 0.5 * v * v;
      }
    }
  }
  if (A.plain_cost) {   // this tile's share of 1/2 |r|^2 over the prior / shape / temporal rows, fixed-order block sum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pcost += __shfl_xor(pcost, off, 64);
    __syncthreads();
    if (lane == 0) sval[wave] = pcost;
    __syncthreads();
    if (tid == 0) {
      double v = 0.0;
      for (int w = 0; w < NT / 64; ++w) v += sval[w];
      store_f64_through(A.plain_cost + (size_t)tile * kReducePartial + fold_slot_cost(1), v);   // (may be summed inside the launch: fold_tail)
    }
  }
}

}  // namespace bodyfit
