// k_reduce.hip — (1) shared-shape reduction of one evaluation, (2) the joint regressor.
//
// (1) For frame-sharded multi-frame solves (include/MultiFrameBA.h:67-118: one beta block shared by
//     every reprojection block) each GPU reduces, over its local rows,
//        out[0]      cost      = sum_kp 1/2 rho(|r_kp|^2) + 1/2 |r_rest|^2     (HuberLoss on reprojection
//                                blocks only, include/MultiFrameBA.h:64,102,110,117,128)
//        out[1..10]  g_beta    = sum_kp rho' J_beta^T r   (+ shape-prior rows)
//        out[11..65] H_bb (upper, row-major) = sum_kp rho' J_beta^T J_beta (+ beta_s^2 I)
//     The 66 doubles are what the ranks all-reduce over xGMI.  H_bb and g_beta are one Gram product of the
//     robustified [J_beta | r] rows on the f64 matrix cores; two deterministic stages (per-wave partials,
//     then one workgroup summing them in fixed order), no atomics.
// (2) k_regress: out[j][c] = sum_v reg[j][v] x[v][c]  (initialJointPos, jointShapeReg) by wave reductions.
#include <algorithm>

#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(4))) double d4;
constexpr int kRedWavesMax = 4096;  // stage-1 wavefronts (one per workgroup), sized by the launch: 16 MFMA steps each
constexpr int kRedSteps = 16;       // 4-row MFMA steps per stage-1 wave: all their loads are in flight together
constexpr int kPartial = 256 + 2;   // 16x16 Gram tile + [huber cost, plain cost] per stage-1 wave

__device__ inline double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Stage 1.  The reprojection rows, robustified by sqrt(rho'), form Jhat = [sqrt(rho') J_beta | sqrt(rho') r]
// ([2K x 11]).  Its Gram matrix Jhat^T Jhat holds H_bb (10x10), g_beta (column 10) in one symmetric product,
// so it runs on the f64 matrix cores: v_mfma_f64_16x16x4_f64 with the SAME register as A and B operand
// (A[i][k] = Jhat[row k][i], B[k][j] = Jhat[row k][j]), four rows per instruction, rows dealt to 256 waves.
__global__ __launch_bounds__(64) void k_reduce_stage1(int K, int ncols, int npose, int nS, int total_rows,
                                                       const double* __restrict__ r, const double* __restrict__ J,
                                                       double delta, double* __restrict__ partials) {
  const int lane = threadIdx.x, w = blockIdx.x, nw = gridDim.x;
  const int col = lane & 15, kk = lane >> 4;
  const bool has_beta = (ncols > npose) && J;
  const double d2 = delta * delta;
  const int nrows = 2 * K;
  const int steps = (nrows + 3) / 4;
  const int per = (steps + nw - 1) / nw;                      // <= kRedSteps unless the launch was capped
  const int s0 = w * per, s1 = min(steps, s0 + per);
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  double hub = 0.0;
  if (has_beta) {
    for (int sb = s0; sb < s1; sb += kRedSteps) {
      // kRedSteps independent steps: residual pair + Jacobian entry of each are requested before any is used (a
      // runtime-bounded loop of dependent round trips is what made this kernel 10x slower than its traffic)
      double rr0[kRedSteps], rr1[kRedSteps], jv[kRedSteps];
#pragma unroll
      for (int u = 0; u < kRedSteps; ++u) {
        const int row = 4 * (sb + u) + kk;
        const bool on = sb + u < s1 && row < nrows;
        const int k = on ? (row >> 1) : 0;
        rr0[u] = on ? r[2 * (size_t)k] : 0.0;
        rr1[u] = on ? r[2 * (size_t)k + 1] : 0.0;
        jv[u] = (on && col < nS) ? J[(size_t)row * ncols + npose + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kRedSteps; ++u) {
        const int row = 4 * (sb + u) + kk;
        const double sq = rr0[u] * rr0[u] + rr1[u] * rr1[u];
        const double rho1 = (delta > 0.0 && sq > d2) ? delta / sqrt(sq) : 1.0;
        const double sw = sqrt(rho1);
        double v = 0.0;
        if (sb + u < s1 && row < nrows) {
          if (col < nS) v = sw * jv[u];
          else if (col == 10) v = sw * ((row & 1) ? rr1[u] : rr0[u]);
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
      }
    }
  }
  // Huber cost of the keypoints and plain least squares of the remaining rows: lanes over rows
  {
    const int kper = (K + nw - 1) / nw;
    const int k0 = w * kper, k1 = min(K, k0 + kper);
    for (int k = k0 + lane; k < k1; k += 64) {
      const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
      const double sq = r0 * r0 + r1 * r1;
      hub += 0.5 * ((delta > 0.0 && sq > d2) ? 2.0 * delta * sqrt(sq) - d2 : sq);
    }
    const int rows = total_rows - nrows;
    const int rper = (rows + nw - 1) / nw;
    const int q0 = nrows + w * rper, q1 = min(total_rows, q0 + rper);
    double pl = 0.0;
    for (int q = q0 + lane; q < q1; q += 4 * 64) {   // four loads in flight per pass
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (q + 64 * u < q1) ? r[q + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) pl += 0.5 * v[u] * v[u];
    }
    hub = wave_sum64(hub);
    pl = wave_sum64(pl);
    if (lane == 0) { partials[(size_t)w * kPartial + 256] = hub; partials[(size_t)w * kPartial + 257] = pl; }
  }
  // D layout (f64 16x16): column = lane & 15, row = (lane >> 4) + 4 q
#pragma unroll
  for (int q = 0; q < 4; ++q) partials[(size_t)w * kPartial + (kk + 4 * q) * 16 + col] = acc[q];
}

// Stage 2: fixed-order sum of the per-wave partials (deterministic), then pack [cost, g(10), upper H(55)].
// 16 slices per entry; a slice's partials are requested in fixed-trip batches of 16 (all in flight), because a
// runtime-bounded loop of loads pays one L2 round trip per iteration.
constexpr int kSlices = 16;
// compact: the partials are the sweep's per-frame rows (bodyfit_device.h kFoldEntries: 67 leading slots), not Gram tiles
__global__ __launch_bounds__(1024) void k_reduce_stage2(int nw, const double* __restrict__ partials, int shape_row0,
                                                         int shape_rows, double beta_shape,
                                                         const double* __restrict__ r, double* __restrict__ out, int compact) {
  __shared__ double sred[kSlices][kPartial];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kSlices * kPartial; idx += 1024) {
    const int slice = idx / kPartial, e = idx % kPartial;
    double tot = 0.0;
    for (int w0 = slice; w0 < nw; w0 += kSlices * 16) {
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int w = w0 + kSlices * u;
        a[u] = (w < nw) ? partials[(size_t)w * kPartial + e] : 0.0;
      }
      tot += (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
             (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
    }
    sred[slice][e] = tot;
  }
  __syncthreads();
  if (tid < 66) {
    auto T = [&](int e) {
      double v = 0.0;
#pragma unroll
      for (int sl = 0; sl < kSlices; ++sl) v += sred[sl][e];
      return v;
    };
    auto G = [&](int i, int j) { return T(compact ? fold_slot_gram(i, j) : i * 16 + j); };
    double v;
    if (tid == 0) {
      v = compact ? T(fold_slot_cost(0)) + T(fold_slot_cost(1)) : T(256) + T(257);
    } else if (tid < 11) {
      v = G(tid - 1, 10);
      if (tid - 1 < shape_rows) v += beta_shape * r[shape_row0 + tid - 1];     // shared shape prior, J = beta_s I
    } else {
      int row = 0, rem = tid - 11;
      while (rem >= 10 - row) { rem -= 10 - row; ++row; }
      v = G(row, row + rem);
      if (rem == 0 && row < shape_rows) v += beta_shape * beta_shape;
    }
    out[tid] = v;
  }
}

// Stage 2a (large problems only): the same fixed-order sum spread over 17 workgroups (16 entries each, 64 slices per
// entry, one batch of loads per thread for up to 1024 stage-1 waves); the totals go through partials[0..kPartial) of a
// second buffer and k_reduce_stage2 then packs them with nw = 1.
// rows_begin < rows_end (folded path only): the last workgroup's threads that own no entry also sum 1/2 r^2 over the
// non-reprojection rows [rows_begin, rows_end) into entry 257.
__global__ __launch_bounds__(1024) void k_reduce_stage2a(int nw, const double* __restrict__ partials,
                                                          double* __restrict__ totals, const double* __restrict__ r,
                                                          int rows_begin, int rows_end) {
  __shared__ double sred[64][17];
  const int tid = threadIdx.x, el = tid & 15, slice = tid >> 4;
  const int e = blockIdx.x * 16 + el;
  double tot = 0.0;
  if (rows_begin < rows_end && e >= kPartial) {
    // 14 idle entry columns x 64 slices = 896 threads over the rows, 8 loads in flight per pass
    const int t = (el - (kPartial & 15)) * 64 + slice;
    for (int q = rows_begin + t; q < rows_end; q += 896 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (q + 896 * u < rows_end) ? r[q + 896 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) tot += 0.5 * v[u] * v[u];
    }
  }
  if (e < kPartial) {
    for (int w0 = slice; w0 < nw; w0 += 64 * 16) {
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int w = w0 + 64 * u;
        a[u] = (w < nw) ? partials[(size_t)w * kPartial + e] : 0.0;
      }
      tot += (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
             (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
    }
  }
  sred[slice][el] = tot;
  __syncthreads();
  if (tid < 16 && e < kPartial) {
    double v = 0.0;
    for (int sl = 0; sl < 64; ++sl) v += sred[sl][tid];
    if (e == 257 && rows_begin < rows_end)
      for (int c = (kPartial & 15); c < 16; ++c)
        for (int sl = 0; sl < 64; ++sl) v += sred[sl][c];
    totals[e] = v;
  }
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

int reduce_partials_doubles() { return (kRedWavesMax + 1) * kPartial; }   // + one row of totals (stage 2a)

void launch_reduce_shared_ex(int K, int ncols, int npose, int nS, int total_rows, const double* d_r,
                             const double* d_J, double huber_delta, int shape_row0, int shape_rows,
                             double beta_shape, double* d_partials, double* d_out66, hipStream_t s) {
  const int steps = (2 * K + 3) / 4;
  const int nw = std::min(kRedWavesMax, std::max(64, (steps + kRedSteps - 1) / kRedSteps));
  BODYFIT_LAUNCH(k_reduce_stage1, dim3(nw), dim3(64), 0, s, K, ncols, npose, nS, total_rows, d_r, d_J,
                     huber_delta, d_partials);
  if (nw > 256) {   // many partials: sum them on 17 workgroups first (a single workgroup pays ~20 serial L2 round trips)
    double* totals = d_partials + (size_t)kRedWavesMax * kPartial;
    BODYFIT_LAUNCH(k_reduce_stage2a, dim3((kPartial + 15) / 16), dim3(1024), 0, s, nw, d_partials, totals, d_r, 0, 0);
    BODYFIT_LAUNCH(k_reduce_stage2, dim3(1), dim3(1024), 0, s, 1, totals, shape_row0, shape_rows, beta_shape, d_r,
                       d_out66, 0);
  } else {
    BODYFIT_LAUNCH(k_reduce_stage2, dim3(1), dim3(1024), 0, s, nw, d_partials, shape_row0, shape_rows, beta_shape,
                       d_r, d_out66, 0);
  }
}

// Folded path: the sweep already left one partial per frame and per prior tile (d_frame_partials, [F][258], compact: the
// first kFoldEntries slots; the rest stay zero from allocation); sum them on 17 workgroups, then pack.
void launch_reduce_frames(int F, const double* d_r, int shape_row0, int shape_rows,
                          double beta_shape, const double* d_frame_partials, double* d_scratch, double* d_out66,
                          hipStream_t s) {
  static_assert(kPartial == kReducePartial, "partial layout");
  if (F <= 256) {   // small shard: one workgroup sums the partials (one batch of loads) and packs
    BODYFIT_LAUNCH(k_reduce_stage2, dim3(1), dim3(1024), 0, s, F, d_frame_partials, shape_row0, shape_rows, beta_shape,
                       d_r, d_out66, 1);
    return;
  }
  BODYFIT_LAUNCH(k_reduce_stage2a, dim3((kPartial + 15) / 16), dim3(1024), 0, s, F, d_frame_partials, d_scratch, d_r, 0, 0);
  BODYFIT_LAUNCH(k_reduce_stage2, dim3(1), dim3(1024), 0, s, 1, d_scratch, shape_row0, shape_rows, beta_shape, d_r,
                     d_out66, 1);
}
// ---- Jacobian packing for the host-pointer path (bodyfit_evaluate_batch without a caller's Jacobian buffer: the cached sweep
//      that bodyfit_evaluate_block serves ceres::CostFunction::Evaluate calls from).  A reprojection block's Jacobian is dense
//      in its 86 columns only on paper: a keypoint moves with its kinematic ancestors, so most of the 23 joint blocks are
//      structurally zero (include/Sim3BA.h:173-207: the chain walk touches the ancestors only).  What crosses PCIe is the
//      column blocks a probe sweep found non-zero: per keypoint [present columns of row 0 | of row 1]. ---------------------
namespace {
// Blocks beyond the K keypoints copy the residual vector and the GMM components (r_out / comp_out may be null): with the
// outputs in device-addressable page-locked host memory this ONE kernel is the whole way down of a cached sweep.
__global__ __launch_bounds__(128) void k_pack_jacobian(int K, int ncols, int njb, const double* __restrict__ J,
                                                        const unsigned* __restrict__ mask, const unsigned* __restrict__ off,
                                                        double* __restrict__ out, const double* __restrict__ r, int nr,
                                                        double* __restrict__ r_out, const int* __restrict__ comp, int ncomp,
                                                        int* __restrict__ comp_out) {
  const int k = blockIdx.x, c = threadIdx.x;
  if (k >= K) {
    const int nrb = r_out ? (nr + 1023) / 1024 : 0;
    const int b = k - K;
    if (b < nrb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b * 1024 + u * 128 + c;
        if (i < nr) r_out[i] = r[i];
      }
    } else if (comp_out) {
      for (int i = (b - nrb) * 1024 + c; i < min(ncomp, (b - nrb + 1) * 1024); i += 128) comp_out[i] = comp[i];
    }
    return;
  }
  // The keypoint's packed span: its structurally non-zero column blocks [scale 1][rootAA 3][rootT 3][joint AA 3 each ...][beta: the
  // rest] (njb: joint blocks = joints - 1) in block order, each block as its two rows back to back — Ceres' [2][size] row-major
  // block, so the host serves a block with one copy.  Thread = OUTPUT element (consecutive threads store consecutive words: whole
  // lines cross PCIe); which (block, row, column) that is follows from the mask, the same for the whole workgroup.
  const unsigned m = mask[k];
  const unsigned o0 = off[k];
  const int total = (int)(off[k + 1] - o0);
  const int nblk = 4 + njb, beta_sz = ncols - (7 + 3 * njb);
  for (int i = c; i < total; i += 128) {
    int at = 0, blk = 0, sz = 1, first = 0;
    for (int b = 0; b < nblk; ++b) {
      if (!((m >> b) & 1u)) continue;
      const int s2 = (b == 0) ? 1 : (b < 3 + njb ? 3 : beta_sz);
      if (i < at + 2 * s2) { blk = b; sz = s2; break; }
      at += 2 * s2;
    }
    first = (blk == 0) ? 0 : (blk < 3 + njb ? 1 + 3 * (blk - 1) : 7 + 3 * njb);
    const int e = i - at, row = e >= sz ? 1 : 0, col = first + (e - row * sz);
    out[o0 + i] = J[(size_t)(2 * k + row) * ncols + col];
  }
}
}  // namespace
void launch_pack_jacobian(int K, int ncols, int n_joint_blocks, const double* d_J, const unsigned* d_mask, const unsigned* d_off,
                          double* d_out, const double* d_r, int nr, double* r_out, const int* d_comp, int ncomp, int* comp_out,
                          hipStream_t s) {
  const int extra = (r_out ? (nr + 1023) / 1024 : 0) + (comp_out ? (ncomp + 1023) / 1024 : 0);
  if (K + extra > 0)
    BODYFIT_LAUNCH(k_pack_jacobian, dim3(K + extra), dim3(128), 0, s, K, ncols, n_joint_blocks, d_J, d_mask, d_off, d_out, d_r, nr,
                   r_out, d_comp, ncomp, comp_out);
}

void launch_regress(int nJ, int V, int ncol, const double* d_reg, const double* d_x, double* d_out, hipStream_t s) {
  BODYFIT_LAUNCH(k_regress, dim3(nJ, ncol), dim3(256), 0, s, V, ncol, d_reg, d_x, d_out);
}


// ---- post-solve write-back (include/MultiFrameBA.h:154-174, include/Sim3BA.h:481-505) -----------------------------
namespace {
// R0' = R(rootAA) R0 (left-multiplied, so it compounds over repeated solves: quirk Q8); update parameters
// [1, 0 0 0, rootT, jointAA]: the Sim3 scale is dropped by the write-back (quirk Q5).
__global__ __launch_bounds__(64) void k_writeback_prepare(int F, int npose, const double* __restrict__ params,
                                                          const double* __restrict__ R0, double* __restrict__ params_upd,
                                                          double* __restrict__ R0_new) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= F) return;
  const double* x = params + (size_t)f * npose;
  double* y = params_upd + (size_t)f * npose;
  const double ax = x[1], ay = x[2], az = x[3];
  const double th = sqrt(ax * ax + ay * ay + az * az);
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > 1e-12) {   // Eigen::AngleAxisd(theta, aa / theta).toRotationMatrix()
    const double kx = ax / th, ky = ay / th, kz = az / th, c = cos(th), s = sin(th), v = 1.0 - c;
    R[0] = c + kx * kx * v;      R[1] = kx * ky * v - kz * s; R[2] = kx * kz * v + ky * s;
    R[3] = ky * kx * v + kz * s; R[4] = c + ky * ky * v;      R[5] = ky * kz * v - kx * s;
    R[6] = kz * kx * v - ky * s; R[7] = kz * ky * v + kx * s; R[8] = c + kz * kz * v;
  }
  const double* A = R0 + (size_t)f * 9;
  double* B = R0_new + (size_t)f * 9;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[i * 3 + j] = R[i * 3] * A[j] + R[i * 3 + 1] * A[3 + j] + R[i * 3 + 2] * A[6 + j];
  y[0] = 1.0; y[1] = 0.0; y[2] = 0.0; y[3] = 0.0;
  for (int i = 4; i < npose; ++i) y[i] = x[i];
}
// mean_pixel_error (include/Utils.h:102-115) of every frame against its own posed joints; keypoints that are not FK
// joints (vertex landmarks, which the reference does not have) are left out of the mean; no keypoints -> 0.
__global__ __launch_bounds__(64) void k_mean_pixel_error(int F, int nJ, const int* __restrict__ kp_offset,
                                                         const int* __restrict__ kp_id, const double* __restrict__ kp_uv,
                                                         const double* __restrict__ joints, double fx, double fy, double cx,
                                                         double cy, double* __restrict__ out) {
  const int f = blockIdx.x, lane = threadIdx.x;
  double sum = 0.0, cnt = 0.0;
  for (int k = kp_offset[f] + lane; k < kp_offset[f + 1]; k += 64) {
    const int id = kp_id[k];
    if (id >= nJ) continue;
    const double* X = joints + ((size_t)f * nJ + id) * 3;
    const double u = fx * X[0] / X[2] + cx, v = fy * X[1] / X[2] + cy;
    const double du = u - kp_uv[2 * (size_t)k], dv = v - kp_uv[2 * (size_t)k + 1];
    sum += sqrt(du * du + dv * dv);
    cnt += 1.0;
  }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); cnt += __shfl_xor(cnt, off, 64); }
  if (lane == 0) out[f] = cnt > 0.0 ? sum / cnt : 0.0;
}
}  // namespace

void launch_writeback_prepare(int F, int npose, const double* d_params, const double* d_R0, double* d_params_upd,
                              double* d_R0_new, hipStream_t s) {
  if (F > 0) BODYFIT_LAUNCH(k_writeback_prepare, dim3((F + 63) / 64), dim3(64), 0, s, F, npose, d_params, d_R0, d_params_upd, d_R0_new);
}
void launch_mean_pixel_error(int F, int nJ, const int* d_kp_offset, const int* d_kp_id, const double* d_kp_uv,
                             const double* d_joints, double fx, double fy, double cx, double cy, double* d_out, hipStream_t s) {
  if (F > 0) BODYFIT_LAUNCH(k_mean_pixel_error, dim3(F), dim3(64), 0, s, F, nJ, d_kp_offset, d_kp_id, d_kp_uv, d_joints, fx, fy, cx, cy, d_out);
}

}  // namespace bodyfit
