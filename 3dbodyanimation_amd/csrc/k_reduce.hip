// k_reduce.hip — (1) shared-shape reduction of one evaluation, (2) the joint regressor.
//
// (1) For frame-sharded multi-frame solves (include/MultiFrameBA.h:67-118: one beta block shared by
//     every reprojection block) each GPU reduces, over its local rows,
//        out[0]      cost      = sum_kp 1/2 rho(|r_kp|^2) + 1/2 |r_rest|^2     (HuberLoss on reprojection
//                                blocks only, include/MultiFrameBA.h:64,102,110,117,128)
//        out[1..10]  g_beta    = sum_kp rho' J_beta^T r   (+ shape-prior rows)
//        out[11..65] H_bb (upper, row-major) = sum_kp rho' J_beta^T J_beta (+ beta_s^2 I)
//     The 66 doubles are what the ranks all-reduce over xGMI.  Two deterministic stages (per-block
//     partials in fixed order, then one block) so the result does not depend on atomics order.
// (2) k_regress: out[j][c] = sum_v reg[j][v] x[v][c]  (initialJointPos, jointShapeReg) by wave reductions.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

constexpr int kRedBlocks = 64;
constexpr int kRedThreads = 128;  // 66 live entries

__device__ inline void tri_index(int e, int& a, int& b) {  // e in [0,55) -> (a<=b) upper, row-major
  int row = 0, rem = e;
  while (rem >= 10 - row) { rem -= 10 - row; ++row; }
  a = row; b = row + rem;
}

__global__ __launch_bounds__(kRedThreads) void k_reduce_stage1(int K, int ncols, int npose, int nS, int total_rows,
                                                                const double* __restrict__ r,
                                                                const double* __restrict__ J, double delta,
                                                                int shape_row0, int shape_rows, double beta_shape,
                                                                double* __restrict__ partials) {
  const int e = threadIdx.x;
  const int nb = gridDim.x, b = blockIdx.x;
  const bool has_beta = (ncols > npose) && J;
  int ia = 0, ib = 0;
  if (e >= 11 && e < 66) tri_index(e - 11, ia, ib);
  double acc = 0.0;
  const double d2 = delta * delta;
  // keypoint blocks, contiguous chunk per block
  const int per = (K + nb - 1) / nb;
  const int k0 = b * per, k1 = min(K, k0 + per);
  for (int k = k0; k < k1; ++k) {
    const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
    const double sq = r0 * r0 + r1 * r1;
    double rho = sq, rho1 = 1.0;
    if (delta > 0.0 && sq > d2) {
      const double rt = sqrt(sq);
      rho = 2.0 * delta * rt - d2;
      rho1 = delta / rt;
    }
    if (e == 0) {
      acc += 0.5 * rho;
    } else if (has_beta && e < 66) {
      const double* j0 = J + (size_t)(2 * k) * ncols + npose;
      const double* j1 = j0 + ncols;
      if (e < 11) {
        if (e - 1 < nS) acc += rho1 * (j0[e - 1] * r0 + j1[e - 1] * r1);
      } else if (ib < nS) {
        acc += rho1 * (j0[ia] * j0[ib] + j1[ia] * j1[ib]);
      }
    }
  }
  // remaining rows (priors, temporal): plain least squares
  if (e == 0) {
    const int rows = total_rows - 2 * K;
    const int rper = (rows + nb - 1) / nb;
    const int q0 = 2 * K + b * rper, q1 = min(total_rows, q0 + rper);
    for (int q = q0; q < q1; ++q) acc += 0.5 * r[q] * r[q];
  }
  // shared shape prior rows: J = beta_s I
  if (b == 0 && shape_rows > 0 && e >= 1 && e < 66) {
    if (e < 11) {
      if (e - 1 < shape_rows) acc += beta_shape * r[shape_row0 + e - 1];
    } else if (ia == ib && ia < shape_rows) {
      acc += beta_shape * beta_shape;
    }
  }
  if (e < 66) partials[(size_t)b * 66 + e] = acc;
}

__global__ __launch_bounds__(kRedThreads) void k_reduce_stage2(int nb, const double* __restrict__ partials,
                                                                double* __restrict__ out) {
  const int e = threadIdx.x;
  if (e >= 66) return;
  double acc = 0.0;
  for (int b = 0; b < nb; ++b) acc += partials[(size_t)b * 66 + e];
  out[e] = acc;
}

__global__ __launch_bounds__(256) void k_regress(int V, int ncol, const double* __restrict__ reg,
                                                  const double* __restrict__ x, double* __restrict__ out) {
  __shared__ double sw[4];
  const int j = blockIdx.x, c = blockIdx.y;
  double acc = 0.0;
  for (int v = threadIdx.x; v < V; v += 256) acc += reg[(size_t)j * V + v] * x[(size_t)v * ncol + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[(size_t)j * ncol + c] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

}  // namespace

int reduce_partials_doubles() { return kRedBlocks * 66; }

void launch_reduce_shared_ex(int K, int ncols, int npose, int nS, int total_rows, const double* d_r,
                             const double* d_J, double huber_delta, int shape_row0, int shape_rows,
                             double beta_shape, double* d_partials, double* d_out66, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_stage1, dim3(kRedBlocks), dim3(kRedThreads), 0, s, K, ncols, npose, nS, total_rows, d_r,
                     d_J, huber_delta, shape_row0, shape_rows, beta_shape, d_partials);
  hipLaunchKernelGGL(k_reduce_stage2, dim3(1), dim3(kRedThreads), 0, s, kRedBlocks, d_partials, d_out66);
}

void launch_regress(int nJ, int V, int ncol, const double* d_reg, const double* d_x, double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_regress, dim3(nJ, ncol), dim3(256), 0, s, V, ncol, d_reg, d_x, d_out);
}

}  // namespace bodyfit
