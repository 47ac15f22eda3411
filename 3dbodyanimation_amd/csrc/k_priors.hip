// k_priors.hip — pose prior, shape prior and temporal residuals for all frames, f64.
//
//   pose prior   PosePriorAAAnalytic::Evaluate (include/Sim3BA.h:263-315): L2 r = beta_p x, or the GMM
//                max-mixture residual of ark::GaussianMixture::residual (uses at :280,288)
//   shape prior  ShapePriorL2Analytic::Evaluate (include/Sim3BA.h:331-343): r = beta_s w
//   temporal     Vec3DiffCost (include/MultiFrameBA.h:20-28,121-142): r = lambda (a_f - a_{f+1}) on
//                rootT, rootAA, then joints 1..23
// One wavefront per frame.  The GMM sweep reads each 69x69 Cholesky factor row-wise with the lanes on
// consecutive columns (coalesced) and picks the component by a wave reduction of |L^T d|^2.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

__device__ inline double wave_sum_all(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(64) void k_priors(int F, int nJ, int nS, int beta_stride, const double* __restrict__ params,
                                                const double* __restrict__ beta, double beta_pose, DevGmm g,
                                                int has_gmm, double beta_shape, double lambda_t, int n_pairs,
                                                double* __restrict__ r_prior, double* __restrict__ r_shape,
                                                double* __restrict__ r_temporal, int* __restrict__ comp_out) {
  __shared__ double sd[128];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int npose = 7 + 3 * (nJ - 1);
  const int D = 3 * (nJ - 1);
  const double* x = params + (size_t)f * npose + 7;

  if (beta_pose > 0.0 && r_prior) {
    if (!has_gmm) {
      for (int i = lane; i < D; i += 64) r_prior[(size_t)f * D + i] = beta_pose * x[i];
      if (comp_out && lane == 0) comp_out[f] = 0;
    } else {
      // two column slots per lane: c0 = lane, c1 = lane + 64 (D = 69)
      double best0 = 0, best1 = 0, bestv = 1.0 / 0.0, bestc = 0;
      int bestk = 0;
      for (int k = 0; k < g.K; ++k) {
        for (int i = lane; i < D; i += 64) sd[i] = x[i] - g.mean[(size_t)k * D + i];
        __syncthreads();
        const double* L = g.prec_cho + (size_t)k * D * D;
        double t0 = 0, t1 = 0;
        const int c0 = lane, c1 = lane + 64;
        for (int r = 0; r < D; ++r) {
          const double dr = sd[r];
          if (c0 <= r && c0 < D) t0 += L[(size_t)r * D + c0] * dr;
          if (c1 <= r && c1 < D) t1 += L[(size_t)r * D + c1] * dr;
        }
        t0 *= g.resid_scale;
        t1 *= g.resid_scale;
        const double sq = wave_sum_all(t0 * t0 + t1 * t1);
        const double val = sq + g.neg_log_w[k];
        if (val < bestv) {
          bestv = val; bestk = k; best0 = t0; best1 = t1; bestc = sqrt(g.neg_log_w[k]);
        }
        __syncthreads();
      }
      double* o = r_prior + (size_t)f * (D + 1);
      if (lane < D) o[lane] = beta_pose * best0;
      if (lane + 64 < D) o[lane + 64] = beta_pose * best1;
      if (lane == 0) {
        o[D] = beta_pose * bestc;
        if (comp_out) comp_out[f] = bestk;
      }
    }
  }
  if (beta_shape > 0.0 && r_shape && beta) {
    if (beta_stride > 0) {
      for (int i = lane; i < nS; i += 64) r_shape[(size_t)f * nS + i] = beta_shape * beta[(size_t)f * beta_stride + i];
    } else if (f == 0) {
      for (int i = lane; i < nS; i += 64) r_shape[i] = beta_shape * beta[i];
    }
  }
  if (lambda_t > 0.0 && r_temporal && f < n_pairs) {
    const double* a = params + (size_t)f * npose;
    const double* b = params + (size_t)(f + 1) * npose;
    for (int i = lane; i < 6 + D; i += 64) {
      const int src = (i < 3) ? (4 + i) : (i < 6 ? (1 + (i - 3)) : (7 + (i - 6)));
      r_temporal[(size_t)f * (6 + D) + i] = lambda_t * (a[src] - b[src]);
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
  hipLaunchKernelGGL(k_priors, dim3(P.F), dim3(64), 0, s, P.F, nJ, nS, P.beta_stride, d_params, d_beta, beta_pose, g,
                     gmm ? 1 : 0, beta_shape, lambda_t, n_pairs, d_r_prior, d_r_shape, d_r_temporal, d_comp);
}

}  // namespace bodyfit
