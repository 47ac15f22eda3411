// k_priors.hip — pose prior, shape prior and temporal residuals for all frames, f64.
//
//   pose prior   PosePriorAAAnalytic::Evaluate (include/Sim3BA.h:263-315): L2 r = beta_p x, or the GMM
//                max-mixture residual of ark::GaussianMixture::residual (uses at :280,288)
//   shape prior  ShapePriorL2Analytic::Evaluate (include/Sim3BA.h:331-343): r = beta_s w
//   temporal     Vec3DiffCost (include/MultiFrameBA.h:20-28,121-142): r = lambda (a_f - a_{f+1}) on
//                rootT, rootAA, then joints 1..23
// The GMM whitening  T_k = (X - mu_k) L_k  ([16 x 69].[69 x 69] per component) is a dense contraction, so it
// runs on the f64 matrix cores: one single-wave workgroup per (tile of 16 frames, component) -- the units
// spread over the CUs the mesh kernel leaves idle -- 5 column tiles x 18 k-steps of v_mfma_f64_16x16x4_f64 with the 16 frames on the MFMA row
// index; L_k is stored in MFMA B-fragment order at upload, so it is read once per (frame tile, component)
// with fully coalesced 16-byte loads, all 54 of them in flight together; the 16 pose vectors and the
// component means go through LDS.  |T_k|^2 per frame is a 16-lane butterfly; k_gmm_select then picks the
// mixture component per frame (first minimum, as the sequential reference loop) and copies its rows.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(4))) double d4;
constexpr int kTileF = 16;           // frames per workgroup (MFMA M)
constexpr int kNT = 5;               // column tiles of 16 (69 -> 80)
constexpr int kKS = 18;              // k-steps of 4 (69 -> 72)

// GMM whitening: one wavefront per (tile of 16 frames, mixture component).
//   scratch_T [F][K][72]  scaled whitened residual T_k = scale (x - mu_k) L_k  (columns >= 69 unused)
//   scratch_v [F][K]      |T_k|^2 - log w'_k
__global__ __launch_bounds__(64) void k_gmm_whiten(int F, int nJ, const double* __restrict__ params, DevGmm g,
                                                    double* __restrict__ scratch_T, double* __restrict__ scratch_v) {
  __shared__ double sx[kTileF * 72];
  const int lane = threadIdx.x;
  const int f0 = blockIdx.x * kTileF, k = blockIdx.y;
  const int npose = 7 + 3 * (nJ - 1);
  const int D = 3 * (nJ - 1);
  const int m = lane & 15, kk = lane >> 4;  // MFMA: A[i = m][k = kk], B[k = kk][j = m]
  // L_k in B-fragment order: 54 fully coalesced 16-byte loads per lane, all in flight together
  const double2* Lf = reinterpret_cast<const double2*>(g.prec_frag) + (size_t)k * kKS * 3 * 64 + lane;
  double2 b[kKS][3];
#pragma unroll
  for (int s = 0; s < kKS; ++s)
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) b[s][pr] = Lf[(size_t)(s * 3 + pr) * 64];
  // the 16 frames' pose vectors minus the component mean, through LDS.  Fixed trip count (16 x 72 / 64 = 18
  // fully unrolled, predicated iterations) so every load is in flight at once: a runtime-bounded loop here
  // serialises one memory round trip per iteration.
  {
    double xv[18], mv[18];
#pragma unroll
    for (int it = 0; it < 18; ++it) {
      const int idx = it * 64 + lane, fr = idx / 72, c = idx % 72;
      const int f = f0 + fr;
      const bool ok = c < D && f < F;
      xv[it] = ok ? params[(size_t)f * npose + 7 + c] : 0.0;
      mv[it] = (c < D) ? g.mean[(size_t)k * D + c] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 18; ++it) sx[it * 64 + lane] = xv[it] - mv[it];
  }
  __syncthreads();
  d4 acc[kNT];
#pragma unroll
  for (int nt = 0; nt < kNT; ++nt) acc[nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < kKS; ++s) {
    const int r = 4 * s + kk;
    const double a = (r < D) ? sx[m * 72 + r] : 0.0;
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][0].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][0].y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][1].x, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][1].y, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][2].x, acc[4], 0, 0, 0);
  }
  // D layout (f64): column = lane & 15, frame row = (lane >> 4) + 4 q
  const double nlw = g.neg_log_w[k];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = f0 + kk + 4 * q;
    double sacc = 0.0;
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
      acc[nt][q] *= g.resid_scale;
      sacc += acc[nt][q] * acc[nt][q];
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) sacc += __shfl_xor(sacc, off, 64);
    if (f < F) {
      double* o = scratch_T + ((size_t)f * g.K + k) * 72;
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt)
        if (16 * nt + m < 72) o[16 * nt + m] = acc[nt][q];
      if (m == 0) scratch_v[(size_t)f * g.K + k] = sacc + nlw;
    }
  }
}

// Max-mixture selection (ark::GaussianMixture::residual's argmin, first minimum wins): one wave per frame.
__global__ __launch_bounds__(256) void k_gmm_select(int F, int D, DevGmm g, double beta_pose,
                                                     const double* __restrict__ scratch_T,
                                                     const double* __restrict__ scratch_v, double* __restrict__ r_prior,
                                                     int* __restrict__ comp_out) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= F) return;
  int best = 0;
  double bv = scratch_v[(size_t)f * g.K];
  for (int k2 = 1; k2 < g.K; ++k2) {
    const double v2 = scratch_v[(size_t)f * g.K + k2];
    if (v2 < bv) { bv = v2; best = k2; }
  }
  const double* T = scratch_T + ((size_t)f * g.K + best) * 72;
  double* o = r_prior + (size_t)f * (D + 1);
  for (int c = lane; c < D; c += 64) o[c] = beta_pose * T[c];
  if (lane == 0) {
    o[D] = beta_pose * sqrt(g.neg_log_w[best]);
    if (comp_out) comp_out[f] = best;
  }
}

// L2 pose prior, shape prior and temporal rows: one workgroup per 16 frames, plain streaming.
__global__ __launch_bounds__(256) void k_priors(int F, int nJ, int nS, int beta_stride, const double* __restrict__ params,
                                                 const double* __restrict__ beta, double beta_pose_l2, double beta_shape,
                                                 double lambda_t, int n_pairs, double* __restrict__ r_prior,
                                                 double* __restrict__ r_shape, double* __restrict__ r_temporal,
                                                 int* __restrict__ comp_out) {
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * kTileF;
  const int npose = 7 + 3 * (nJ - 1);
  const int D = 3 * (nJ - 1);
  if (beta_pose_l2 > 0.0 && r_prior) {
    for (int i = tid; i < kTileF * D; i += blockDim.x) {
      const int f = f0 + i / D, c = i % D;
      if (f < F) r_prior[(size_t)f * D + c] = beta_pose_l2 * params[(size_t)f * npose + 7 + c];
    }
    if (comp_out && tid < kTileF && f0 + tid < F) comp_out[f0 + tid] = 0;
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
                   double* d_r_prior, double* d_r_shape, double* d_r_temporal, int* d_comp, double* d_gmm_T,
                   double* d_gmm_v, hipStream_t s) {
  if (P.F <= 0) return;
  const int nblk = (P.F + kTileF - 1) / kTileF;
  const bool use_gmm = gmm && beta_pose > 0.0;
  if (use_gmm) {
    hipLaunchKernelGGL(k_gmm_whiten, dim3(nblk, gmm->K), dim3(64), 0, s, P.F, nJ, d_params, *gmm, d_gmm_T, d_gmm_v);
    hipLaunchKernelGGL(k_gmm_select, dim3((P.F + 3) / 4), dim3(256), 0, s, P.F, 3 * (nJ - 1), *gmm, beta_pose, d_gmm_T,
                       d_gmm_v, d_r_prior, d_comp);
  }
  const double bp_l2 = use_gmm ? 0.0 : beta_pose;
  if (bp_l2 > 0.0 || beta_shape > 0.0 || lambda_t > 0.0)
    hipLaunchKernelGGL(k_priors, dim3(nblk), dim3(256), 0, s, P.F, nJ, nS, P.beta_stride, d_params, d_beta, bp_l2,
                       beta_shape, lambda_t, n_pairs, d_r_prior, d_r_shape, d_r_temporal, d_comp);
}

}  // namespace bodyfit
