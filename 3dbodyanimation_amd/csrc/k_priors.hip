// k_priors.hip — pose prior, shape prior and temporal residuals for all frames, f64.
//
//   pose prior   PosePriorAAAnalytic::Evaluate (include/Sim3BA.h:263-315): L2 r = beta_p x, or the GMM
//                max-mixture residual of ark::GaussianMixture::residual (uses at :280,288)
//   shape prior  ShapePriorL2Analytic::Evaluate (include/Sim3BA.h:331-343): r = beta_s w
//   temporal     Vec3DiffCost (include/MultiFrameBA.h:20-28,121-142): r = lambda (a_f - a_{f+1}) on
//                rootT, rootAA, then joints 1..23
// One workgroup per tile of 16 frames.  The GMM whitening  T_k = (X - mu_k) L_k  ([16 x 69].[69 x 69] per
// component) is a dense contraction, so it runs on the f64 matrix cores: wave k of the workgroup owns
// component k, 5 column tiles x 18 k-steps of v_mfma_f64_16x16x4_f64 with the 16 frames on the MFMA row
// index; L_k is read once per (frame tile, component), 128-B row segments, half of it in flight at a
// time.  |T_k|^2 per frame is a 16-lane butterfly; the mixture component is picked across the waves
// through 1 KiB of LDS and only the winning wave writes its rows (no recompute, no atomics).
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(4))) double d4;
constexpr int kTileF = 16;           // frames per workgroup (MFMA M)
constexpr int kNT = 5;               // column tiles of 16 (69 -> 80)
constexpr int kKS = 18;              // k-steps of 4 (69 -> 72)
constexpr int kMaxComp = 8;            // one wave per component, 512 threads

__global__ __launch_bounds__(512) void k_priors(int F, int nJ, int nS, int beta_stride, const double* __restrict__ params,
                                                  const double* __restrict__ beta, double beta_pose, DevGmm g,
                                                  int has_gmm, double beta_shape, double lambda_t, int n_pairs,
                                                  double* __restrict__ r_prior, double* __restrict__ r_shape,
                                                  double* __restrict__ r_temporal, int* __restrict__ comp_out) {
  __shared__ double sval[kMaxComp * kTileF];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int f0 = blockIdx.x * kTileF;
  const int npose = 7 + 3 * (nJ - 1);
  const int D = 3 * (nJ - 1);

  if (beta_pose > 0.0 && r_prior) {
    if (!has_gmm) {
      for (int i = tid; i < kTileF * D; i += blockDim.x) {
        const int f = f0 + i / D, c = i % D;
        if (f < F) r_prior[(size_t)f * D + c] = beta_pose * params[(size_t)f * npose + 7 + c];
      }
      if (comp_out && tid < kTileF && f0 + tid < F) comp_out[f0 + tid] = 0;
    } else {
      const int k = wave;                       // component owned by this wave (blockDim = 64 K)
      const int m = lane & 15, kk = lane >> 4;  // MFMA: A[i = m][k = kk], B[k = kk][j = m]
      const int fa = f0 + m;
      const double* xa = params + (size_t)min(fa, F - 1) * npose + 7;
      const double* mu = g.mean + (size_t)k * D;
      const double* L = g.prec_cho + (size_t)k * D * D;
      d4 acc[kNT];
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) acc[nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        double a[kKS / 2], b[kKS / 2][kNT];
#pragma unroll
        for (int s = 0; s < kKS / 2; ++s) {
          const int r = 4 * (half * (kKS / 2) + s) + kk;
          const bool rv = r < D;
          a[s] = (rv && fa < F) ? xa[r] - mu[r] : 0.0;
#pragma unroll
          for (int nt = 0; nt < kNT; ++nt) {
            const int c = 16 * nt + m;
            b[s][nt] = (rv && c < D) ? L[(size_t)r * D + c] : 0.0;
          }
        }
#pragma unroll
        for (int s = 0; s < kKS / 2; ++s)
#pragma unroll
          for (int nt = 0; nt < kNT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s][nt], acc[nt], 0, 0, 0);
      }
      // D layout (f64): column = lane & 15, frame row = (lane >> 4) + 4 q
      const double nlw = g.neg_log_w[k];
      double sq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double sacc = 0.0;
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
          acc[nt][q] *= g.resid_scale;
          sacc += acc[nt][q] * acc[nt][q];
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) sacc += __shfl_xor(sacc, off, 64);
        sq[q] = sacc + nlw;
      }
      if (m == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sval[k * kTileF + kk + 4 * q] = sq[q];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = kk + 4 * q, f = f0 + row;
        int best = 0;
        double bv = sval[row];
        for (int k2 = 1; k2 < g.K; ++k2) {
          const double v2 = sval[k2 * kTileF + row];
          if (v2 < bv) { bv = v2; best = k2; }
        }
        if (best == k && f < F) {
          double* o = r_prior + (size_t)f * (D + 1);
#pragma unroll
          for (int nt = 0; nt < kNT; ++nt) {
            const int c = 16 * nt + m;
            if (c < D) o[c] = beta_pose * acc[nt][q];
          }
          if (m == 0) {
            o[D] = beta_pose * sqrt(nlw);
            if (comp_out) comp_out[f] = k;
          }
        }
      }
    }
  }
  if (beta_shape > 0.0 && r_shape && beta) {
    if (beta_stride > 0) {
      for (int i = tid; i < kTileF * nS; i += blockDim.x) {
        const int f = f0 + i / nS, c = i % nS;
        if (f < F) r_shape[(size_t)f * nS + c] = beta_shape * beta[(size_t)f * beta_stride + c];
      }
    } else if (blockIdx.x == 0) {
      for (int i = tid; i < nS; i += blockDim.x) r_shape[i] = beta_shape * beta[i];
    }
  }
  if (lambda_t > 0.0 && r_temporal) {
    const int T = 6 + D;
    for (int i = tid; i < kTileF * T; i += blockDim.x) {
      const int f = f0 + i / T, c = i % T;
      if (f < n_pairs) {
        const int src = (c < 3) ? (4 + c) : (c < 6 ? (1 + (c - 3)) : (7 + (c - 6)));
        r_temporal[(size_t)f * T + c] = lambda_t * (params[(size_t)f * npose + src] - params[(size_t)(f + 1) * npose + src]);
      }
    }
  }
}

}  // namespace

void launch_priors(const DevProblem& P, int nJ, int nS, const double* d_params, const double* d_beta,
                   double beta_pose, const DevGmm* gmm, double beta_shape, double lambda_t, int n_pairs,
                   double* d_r_prior, double* d_r_shape, double* d_r_temporal, int* d_comp, hipStream_t s) {
  if (P.F <= 0) return;
  DevGmm g{};
  if (gmm) g = *gmm;
  const int nblk = (P.F + kTileF - 1) / kTileF;
  const int threads = gmm ? 64 * g.K : 256;
  hipLaunchKernelGGL(k_priors, dim3(nblk), dim3(threads), 0, s, P.F, nJ, nS, P.beta_stride, d_params, d_beta,
                     beta_pose, g, gmm ? 1 : 0, beta_shape, lambda_t, n_pairs, d_r_prior, d_r_shape, d_r_temporal,
                     d_comp);
}

}  // namespace bodyfit
