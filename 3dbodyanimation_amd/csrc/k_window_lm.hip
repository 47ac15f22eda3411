// k_window_lm.hip — device-resident Levenberg-Marquardt for ONE problem over all frames of a window with a shared shape
// block: what OptimizeMultiFrame hands to ceres::Solve with DENSE_QR (include/MultiFrameBA.h:144-151).
//
// The normal equations of such a window are block tridiagonal in the frames (76 x 76 blocks, coupled by the temporal
// term: diagonal off-diagonal blocks) with a 10-wide border (beta).  host_solver.cpp factors the chain frame after frame
// on the host; here it is solved on the device, parallel in the frames, by BLOCK CYCLIC REDUCTION with the border carried
// as 11 right-hand sides [B | rhs]:
//   level l: every second remaining frame j (neighbours a < j < b) is eliminated (the larger independent set when the count
//   is odd: bodyfit_api.hip build_cr_schedule):
//       D_j = L L^T,  P = L^-1 U_a^T,  Q = L^-1 U_j,  Y = L^-1 R_j                         (k_cr_factor, two workgroups per j)
//       D_a -= P^T P,  D_b -= Q^T Q,  U_a := -P^T Q,  R_a -= P^T Y,  R_b -= Q^T Y          (k_cr_update, f64 MFMA)
//   after ceil(log2 F) levels one frame is left: x = D^-1 R; then down again: x_j = L^-T (Y - P x_a - Q x_b)  (k_cr_back)
//   beta:  S = C - B^T X_B,  d_beta = S^-1 (rhs_b - B^T x),  d_f = x_f - X_B,f d_beta     (k_win_schur*, k_win_beta_solve)
// (U_j: coupling block (j, next remaining frame); at level 0 it is the diagonal temporal block, afterwards dense.)
// Around it, all of Ceres' trust-region logic as restated in host_solver.cpp (Jacobi scaling fixed at the first iterate,
// LM damping, projected scale bounds, step quality, radius update, the three termination tests) runs in small kernels on
// the device: per LM iteration the host launches a fixed sequence and reads back one 16-double status record (every fourth
// iteration on one GPU).  Single-GPU windows of up to 256 frames end an iteration with ONE launch for step, model change and
// decision (k_win_tail: the last workgroup, found by a ticket, decides).
// Every block that moves between global memory and LDS does so with all its loads issued before the first is used
// (BlockRegs): written as load-store loops these kernels spent a third of their time in dependent L2 round trips.
// Blocks are padded to 80 x 80 (identity on the padding), right-hand sides to 16 rows, all stored row-major; "t" buffers
// hold transposes (Pt[i][k] = P[k][i]) so that every product is  C[i][i'] = sum_k X[i][k] Y[i'][k]  with k contiguous.
#include "bodyfit_device.h"
#include "dense_inl.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(4))) double d4;
constexpr int NP = kFrameParams;     // 76
constexpr int NBETA = kMaxShape;     // 10
constexpr int WB = kWinBlock;        // 80
constexpr int WR = kWinRhs;          // 16
constexpr int LD = WB + 1;           // LDS leading dimension
constexpr int kHLd = kNormalLd, kHRows = kNormalRows;   // k_frame_normal panels: 87 x 88, n = 86

__device__ inline double huber_rho_w(double delta, double s) {
  const double b = delta * delta;
  if (delta > 0.0 && s > b) return 2.0 * delta * sqrt(s) - b;
  return s;
}
__device__ inline double block_sum_n(double v, double* red, int tid, int nwaves) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nwaves; ++w) s += red[w];
  return s;
}
__device__ inline double block_max_n(double v, double* red, int tid, int nwaves) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nwaves; ++w) s = fmax(s, red[w]);
  return s;
}
// temporal row i (0..74) constrains parameter src(i): rootT, rootAA, then the joints (include/MultiFrameBA.h:121-142)
__device__ inline int temporal_src(int i) { return (i < 3) ? (4 + i) : (i < 6 ? (1 + (i - 3)) : (7 + (i - 6))); }
__device__ inline int temporal_row_of(int s) { return (s >= 7) ? (s - 7 + 6) : (s >= 4 ? (s - 4) : (s - 1 + 3)); }   // s >= 1

// dst[i] = src[i], i in [0, n): eight loads in flight per thread and pass (a load-store loop is one dependent round trip per trip)
__device__ __forceinline__ void copy_batched(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nthreads) {
  int i = tid;
  for (; i + 7 * nthreads < n; i += 8 * nthreads) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * nthreads];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * nthreads] = v[u];
  }
  for (; i < n; i += nthreads) dst[i] = src[i];
}

// ---- cost of a residual vector: 1/2 sum rho(|r_kp|^2) over the keypoints + 1/2 |other rows|^2 -----------------------
__device__ double window_cost(const WinProblem& P, const double* __restrict__ r, double* red, int tid, int nthreads) {
  // independent partial sums per thread, sixteen loads in flight per pass (a 1024-frame window has 170 rows per thread: one
  // dependent load per pass made this the second longest kernel of an iteration, four in flight still left it at 39 us)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = tid;
  for (; k + 7 * nthreads < P.K; k += 8 * nthreads) {
    double xx[8], yy[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { xx[u] = r[2 * (size_t)(k + u * nthreads)]; yy[u] = r[2 * (size_t)(k + u * nthreads) + 1]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u += 4) {
      a0 += 0.5 * huber_rho_w(P.huber, xx[u] * xx[u] + yy[u] * yy[u]);
      a1 += 0.5 * huber_rho_w(P.huber, xx[u + 1] * xx[u + 1] + yy[u + 1] * yy[u + 1]);
      a2 += 0.5 * huber_rho_w(P.huber, xx[u + 2] * xx[u + 2] + yy[u + 2] * yy[u + 2]);
      a3 += 0.5 * huber_rho_w(P.huber, xx[u + 3] * xx[u + 3] + yy[u + 3] * yy[u + 3]);
    }
  }
  for (; k < P.K; k += nthreads) {
    const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
    a0 += 0.5 * huber_rho_w(P.huber, r0 * r0 + r1 * r1);
  }
  int i = 2 * P.K + tid;
  for (; i + 15 * nthreads < P.total_rows; i += 16 * nthreads) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = r[i + u * nthreads];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u += 4) { a0 += 0.5 * v[u] * v[u]; a1 += 0.5 * v[u + 1] * v[u + 1]; a2 += 0.5 * v[u + 2] * v[u + 2]; a3 += 0.5 * v[u + 3] * v[u + 3]; }
  }
  for (; i < P.total_rows; i += nthreads) a0 += 0.5 * r[i] * r[i];
  const double acc = (a0 + a1) + (a2 + a3);
  return block_sum_n(acc, red, tid, nthreads / 64);
}

// The same cost from the partials the Jacobian sweep at that point left (P.cost_partials): F + tiles values instead of every
// residual row — at 1,024 frames the walk over the residual vector was 45 us of a 0.9 ms iteration, one workgroup reading 1.6 MB.
__device__ double window_cost_any(const WinProblem& P, const double* __restrict__ r, double* red, int tid, int nthreads) {
  if (!P.cost_partials) return window_cost(P, r, red, tid, nthreads);
  double a = 0.0;
  const int n = P.F + P.cost_tiles;
  for (int i = tid; i < n; i += nthreads) a += P.cost_partials[(size_t)i * kReducePartial + (i < P.F ? fold_slot_cost(0) : fold_slot_cost(1))];
  return block_sum_n(a, red, tid, nthreads / 64);
}

// mode 0: all; 1: this shard's cost -> W.fin[0] only; 2: initialise the status from W.fin[0] (summed over the shards)
__global__ __launch_bounds__(1024) void k_win_init(WinProblem P, WinBuf W, const double* __restrict__ r, int mode) {
  __shared__ double red[16];
  double c = 0.0;
  if (mode != 2) c = window_cost_any(P, r, red, threadIdx.x, 1024);
  if (mode == 1) { if (threadIdx.x == 0) W.fin[0] = c; return; }
  if (mode == 2) c = W.fin[0];
  if (threadIdx.x == 0) {
    double* st = W.status;
    st[kWsCost] = c; st[kWsInitialCost] = c; st[kWsRadius] = 1e4; st[kWsDec] = 2.0; st[kWsModel] = 0.0;
    st[kWsHasCand] = 0.0; st[kWsIters] = 0.0; st[kWsOk] = 0.0; st[kWsBad] = 0.0;
    const bool finite = (c == c) && c < 1e300;
    st[kWsActive] = finite ? 1.0 : 0.0;
    st[kWsTermination] = finite ? 1.0 : 2.0;
    st[kWsAccepted] = 1.0; st[kWsGmax] = 0.0; st[kWsNewCost] = c; st[kWsJsel] = 0.0;
    st[kWsPoison] = 0.0;   // (the record lives in the problem's pool: a poisoned solve must not poison the next one)
    *W.fail = 0;
  }
}

// ---- beta block: C = sum_f C_f (+ shape prior), g_beta; scaling; damped scaled copy ------------------------------------
// mode 0: all; 1: this shard's sums -> W.Craw, W.gbraw only; 2: scaling and the damped copy from W.Craw, W.gbraw (summed
// over the shards by the caller)
__global__ __launch_bounds__(1024) void k_win_beta(WinProblem P, WinBuf W, const double* __restrict__ Hpan,
                                                   const double* __restrict__ r, int first, int mode) {
  __shared__ double sC[NBETA * NBETA], sg[NBETA], ssc[NBETA];
  __shared__ double part[8][128];
  const int tid = threadIdx.x, F = P.F;
  if (tid == 0 && mode != 2) *W.fail = 0;
  if (P.nb == 0) return;
  if (mode == 2) {
    if (tid < NBETA * NBETA) sC[tid] = W.Craw[tid];
    if (tid < NBETA) sg[tid] = W.gbraw[tid];
  } else {
    // C = sum_f H_f[beta, beta], g_beta = sum_f H_f[rhs row, beta]: 110 words, eight frame lanes of 128 threads, four
    // independent loads per pass
    const int w = tid & 127, g = tid >> 7;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (w < NBETA * NBETA + NBETA) {
      size_t off;
      if (w < NBETA * NBETA) {
        const int a = w / NBETA, b = w % NBETA, lo = a > b ? a : b, hi = a > b ? b : a;
        off = (size_t)(NP + lo) * kHLd + NP + hi;
      } else {
        off = (size_t)(NP + NBETA) * kHLd + NP + (w - NBETA * NBETA);
      }
      const size_t fs = (size_t)kHRows * kHLd;
      int f = g;
      for (; f + 120 < F; f += 128) {        // sixteen loads in flight
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = Hpan[(size_t)(f + 8 * u) * fs + off];
        __builtin_amdgcn_sched_barrier(0);     // (without it the scheduler pairs every load with its add again)
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
      }
      for (; f < F; f += 8) a0 += Hpan[(size_t)f * fs + off];
    }
    part[g][w] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < NBETA * NBETA + NBETA) {
      const double v = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) +
                       ((part[4][tid] + part[5][tid]) + (part[6][tid] + part[7][tid]));
      if (tid < NBETA * NBETA) sC[tid] = v; else sg[tid - NBETA * NBETA] = v;
    }
  }
  __syncthreads();
  if (mode != 2 && tid < NBETA && P.shape_rows > 0) {   // ShapePriorL2Analytic: r = beta_s w, J = beta_s I (include/Sim3BA.h:336-340)
    sC[tid * NBETA + tid] += P.beta_shape * P.beta_shape;
    sg[tid] += P.beta_shape * r[P.row_shape + tid];
  }
  __syncthreads();
  if (mode == 1) {
    if (tid < NBETA * NBETA) W.Craw[tid] = sC[tid];
    if (tid < NBETA) W.gbraw[tid] = sg[tid];
    return;
  }
  if (tid < NBETA) {
    const double s = first ? 1.0 / (1.0 + sqrt(sC[tid * NBETA + tid])) : W.scale[(size_t)F * NP + tid];
    if (first) W.scale[(size_t)F * NP + tid] = s;
    ssc[tid] = s;
    W.gbraw[tid] = sg[tid];
    W.rhsb[tid] = -sg[tid] * s;
    W.gmaxp[F] = 0.0;
  }
  __syncthreads();
  if (tid < NBETA * NBETA) {
    const int a = tid / NBETA, b = tid % NBETA;
    W.Craw[tid] = sC[tid];
    double v = sC[tid] * ssc[a] * ssc[b];
    if (a == b) v += fmin(fmax(v, 1e-6), 1e32) / W.status[kWsRadius];
    W.Cs[tid] = v;
  }
  if (tid == 0) {
    double gm = 0.0;
    for (int b = 0; b < NBETA; ++b) gm = fmax(gm, fabs(sg[b]));
    W.gmaxp[F] = gm;
  }
}

// ---- per frame: complete the normal-equation block (priors, temporal), scale, damp, write the CR operands ---------------
// Shards of a window (bodyfit_solve_sharded): x_left = parameters of the frame in front of the shard's first one (null: none),
// P.halo: a temporal pair leaves the shard behind its last frame (its residual rows are this shard's), scale_halo = Jacobi
// scaling of that next frame (null on the first pass of the first iteration, which only produces this shard's scaling).
constexpr int kAsmThreads = 1024;
__global__ __launch_bounds__(kAsmThreads) void k_win_assemble(WinProblem P, WinBuf W, const double* __restrict__ Hpan,
                                                      const double* __restrict__ r, const double* __restrict__ x,
                                                      const unsigned char* __restrict__ constant, int first,
                                                      const double* __restrict__ x_left, const double* __restrict__ scale_halo) {
  __shared__ double sA[NP * (NP + 1)];
  __shared__ double sB[NBETA * NP];           // H[beta rows][pose columns]: the frame's B block
  __shared__ double sg[NP], ss[NP], ssn[NP], scf[WB], sbs[NBETA], red[kAsmThreads / 64];
  const int f = blockIdx.x, tid = threadIdx.x, F = P.F;
  const double* H = Hpan + (size_t)f * kHRows * kHLd;
  const double lam2 = P.lambda_t * P.lambda_t, bp2 = P.beta_pose * P.beta_pose;
  const bool pair_right = f + 1 < F || P.halo, pair_left = f > 0 || x_left != nullptr;
  const int npairs = (P.lambda_t > 0.0) ? ((int)pair_right + (int)pair_left) : 0;
  // Every global operand of the frame is requested before the first is used (a load-store loop compiles to one dependent
  // L2 round trip per trip: 23 of them for the 76 x 76 block alone)
  constexpr int kAPasses = (NP * NP + kAsmThreads - 1) / kAsmThreads, kBPasses = (NBETA * NP + kAsmThreads - 1) / kAsmThreads;
  double av[kAPasses], bv[kBPasses];
#pragma unroll
  for (int u = 0; u < kAPasses; ++u) {
    const int e = min(tid + u * kAsmThreads, NP * NP - 1);
    const int i = e / NP, j = e % NP, lo = i > j ? i : j, hi = i > j ? j : i;
    av[u] = H[(size_t)lo * kHLd + hi];
  }
#pragma unroll
  for (int u = 0; u < kBPasses; ++u) {
    const int e = min(tid + u * kAsmThreads, NBETA * NP - 1);
    bv[u] = H[(size_t)(NP + e / NP) * kHLd + e % NP];
  }
  const int tc = min(tid, NP - 1);
  double g_in = H[(size_t)(NP + NBETA) * kHLd + tc];
  const double rp_in = (P.prior_rows > 0) ? r[P.row_prior + (size_t)f * P.prior_rows + max(tc - 7, 0)] : 0.0;
  const int ti_c = temporal_row_of(max(tc, 1));
  const double rt_r = (P.lambda_t > 0.0 && pair_right) ? r[P.row_temporal + (size_t)f * 75 + ti_c] : 0.0;
  const double rt_l = (P.lambda_t > 0.0 && f > 0) ? r[P.row_temporal + (size_t)(f - 1) * 75 + ti_c] : 0.0;
  const double xl_in = (f == 0 && x_left) ? x_left[tc] - x[tc] : 0.0;
  const double cf_in = (constant && constant[tc]) ? 1.0 : 0.0;
  const double sc_in = first ? 0.0 : W.scale[(size_t)f * NP + tc];
  const double scn_in = (!first && f + 1 < F) ? W.scale[(size_t)(f + 1) * NP + tc] : 0.0;
  double dn_in = (first && f + 1 < F) ? Hpan[(size_t)(f + 1) * kHRows * kHLd + (size_t)tc * kHLd + tc] : 0.0;
  const double sh_in = (f + 1 == F && P.halo && scale_halo) ? scale_halo[tc] : 0.0;
  const double sbeta_in = W.scale[(size_t)F * NP + min(tid, NBETA - 1)];
  const double inv_radius = 1.0 / W.status[kWsRadius];
#pragma unroll
  for (int u = 0; u < kAPasses; ++u) {
    const int e = tid + u * kAsmThreads;
    if (e < NP * NP) {
      const int i = e / NP, j = e % NP;
      double v = av[u];
      if (i == j) {
        if (i >= 7 && P.prior_rows > 0) v += bp2;       // PosePriorAAAnalytic, L2 branch (include/Sim3BA.h:304-310)
        if (i >= 1) v += lam2 * npairs;                 // Vec3DiffCost on rootT, rootAA, joints (include/MultiFrameBA.h:121-142)
      }
      sA[i * (NP + 1) + j] = v;
    }
  }
#pragma unroll
  for (int u = 0; u < kBPasses; ++u)
    if (tid + u * kAsmThreads < NBETA * NP) sB[tid + u * kAsmThreads] = bv[u];
  if (tid < WB) scf[tid] = (tid < NP) ? cf_in : 0.0;
  if (tid < NBETA) sbs[tid] = sbeta_in;
  if (tid < NP) {
    double g = g_in;
    if (tid >= 7 && P.prior_rows > 0) g += P.beta_pose * rp_in;
    if (tid >= 1 && P.lambda_t > 0.0) {
      if (pair_right) g += P.lambda_t * rt_r;
      if (f > 0) g -= P.lambda_t * rt_l;
      else if (x_left) g -= P.lambda_t * (P.lambda_t * xl_in);   // the previous shard's last pair
    }
    sg[tid] = g;
  }
  __syncthreads();
  if (tid < NP) {
    double s, sn = 0.0;
    if (first) {
      s = 1.0 / (1.0 + sqrt(sA[tid * (NP + 1) + tid]));
      W.scale[(size_t)f * NP + tid] = s;
      if (f + 1 < F) {   // the next frame's scale, from its diagonal entry alone (its workgroup may not have run yet)
        double dn = dn_in;
        if (tid >= 7 && P.prior_rows > 0) dn += bp2;
        if (tid >= 1 && P.lambda_t > 0.0) dn += lam2 * (1 + (int)(f + 2 < F || P.halo));
        sn = 1.0 / (1.0 + sqrt(dn));
      }
    } else {
      s = sc_in;
      if (f + 1 < F) sn = scn_in;
    }
    if (f + 1 == F && P.halo && scale_halo) sn = sh_in;
    ss[tid] = s; ssn[tid] = sn;
    W.graw[(size_t)f * NP + tid] = sg[tid];
    W.Eraw[(size_t)f * NP + tid] = (tid >= 1 && pair_right && P.lambda_t > 0.0) ? -lam2 : 0.0;
  }
  __syncthreads();
  double* D = W.D + (size_t)f * WB * WB;
  double* U = W.U + (size_t)f * WB * WB;
  for (int e = tid; e < WB * WB; e += kAsmThreads) {
    const int i = e / WB, j = e % WB;
    double v = (i == j) ? 1.0 : 0.0, u = 0.0;
    if (i < NP && j < NP) {
      if (scf[i] == 0.0 && scf[j] == 0.0) {
        v = sA[i * (NP + 1) + j] * ss[i] * ss[j];
        if (i == j) {
          v += fmin(fmax(v, 1e-6), 1e32) * inv_radius;
          if (i >= 1 && pair_right && P.lambda_t > 0.0) u = -lam2 * ss[i] * ssn[i];
        }
      }
    }
    D[e] = v;
    U[e] = u;
  }
  for (int e = tid; e < NP * NP; e += kAsmThreads) W.Araw[(size_t)f * NP * NP + e] = sA[(e / NP) * (NP + 1) + e % NP];
  double* Rt = W.Rt + (size_t)f * WR * WB;
  double* Rt0 = W.Rt0 + (size_t)f * WR * WB;
  for (int e = tid; e < WR * WB; e += kAsmThreads) {
    const int c = e / WB, i = e % WB;
    double v = 0.0;
    if (i < NP && scf[i] == 0.0) {
      if (c < P.nb) v = sB[c * NP + i] * ss[i] * sbs[c];
      else if (c == NBETA) v = -sg[i] * ss[i];
    }
    Rt[e] = v;
    Rt0[e] = v;
  }
  for (int e = tid; e < NP * NBETA; e += kAsmThreads) {
    const int i = e / NBETA, c = e % NBETA;
    W.Braw[(size_t)f * NP * NBETA + e] = (c < P.nb) ? sB[c * NP + i] : 0.0;
  }
  // gradient tolerance test: max |g_i| over the free parameters, the bounded scale projected (Ceres gradient_tolerance)
  double gm = 0.0;
  if (tid < NP && scf[tid] == 0.0) {
    double gi = sg[tid];
    if (tid == 0) {
      const double s0 = x[(size_t)f * NP];
      gi = s0 - fmin(fmax(s0 - gi, P.scale_lo), P.scale_hi);
    }
    gm = fabs(gi);
  }
  gm = block_max_n(gm, red, tid, kAsmThreads / 64);
  if (tid == 0) W.gmaxp[f] = gm;
}

// Staging of 80 x 80 (or 16 x 80) blocks between HBM/L2 and LDS.  A plain `for (idx = tid; idx < n; idx += threads) lds[..] =
// g[idx]` compiles to load -> s_waitcnt vmcnt(0) -> ds_write per trip: thirteen DEPENDENT L2 round trips per block (k_cr_factor
// spent 9 of its 27 us in them).  Here every load of a block is issued before the first is used (fixed trip count, clamped
// index, predicated use).
constexpr int kCrThreads = 512, kCrWaves = 8;
template <int ROWS>
struct BlockRegs { static constexpr int kPasses = (ROWS * WB + kCrThreads - 1) / kCrThreads; double v[kPasses]; };
template <int ROWS>
__device__ __forceinline__ void block_load(BlockRegs<ROWS>& r, const double* __restrict__ src, int tid) {
#pragma unroll
  for (int u = 0; u < BlockRegs<ROWS>::kPasses; ++u) r.v[u] = src[min(tid + u * kCrThreads, ROWS * WB - 1)];
}
// dst[row][col] (leading dimension LD); transposed: the element (i, k) of the source lands at row k, column i
template <int ROWS, bool kTransposed = false, bool kLowerOnly = false>
__device__ __forceinline__ void block_to_lds(const BlockRegs<ROWS>& r, double* dst, int tid) {
#pragma unroll
  for (int u = 0; u < BlockRegs<ROWS>::kPasses; ++u) {
    const int idx = tid + u * kCrThreads;
    if (idx < ROWS * WB) {
      const int i = idx / WB, k = idx % WB;
      const double v = (kLowerOnly && k > i) ? 0.0 : r.v[u];
      dst[kTransposed ? k * LD + i : i * LD + k] = v;
    }
  }
}
// LDS [ROWS][LD] -> global [ROWS][WB]: the LDS reads of the block first, then its stores
template <int ROWS, bool kLowerOnly = false>
__device__ __forceinline__ void block_from_lds(double* __restrict__ dst, const double* src, int tid) {
  BlockRegs<ROWS> r;
#pragma unroll
  for (int u = 0; u < BlockRegs<ROWS>::kPasses; ++u) {
    const int idx = min(tid + u * kCrThreads, ROWS * WB - 1);
    r.v[u] = src[(idx / WB) * LD + idx % WB];
  }
#pragma unroll
  for (int u = 0; u < BlockRegs<ROWS>::kPasses; ++u) {
    const int idx = tid + u * kCrThreads;
    if (idx < ROWS * WB) dst[idx] = (kLowerOnly && idx % WB > idx / WB) ? 0.0 : r.v[u];
  }
}

// ---- cyclic reduction: factor one eliminated block, solve its appended rows -------------------------------------------
// Two workgroups per eliminated frame j (side = blockIdx.x & 1): both factor D_j = L L^T (right-looking, 16-column
// panels, the k_lm_step scheme: diagonal block in the registers of wave 0, panel solve one row per thread, trailing update
// on the f64 matrix cores) with rows appended below that receive L^-T from the right:
//   side 0:  rows of U_a (-> Pt_j)  and the 16 rows of Rt_j (-> Yt_j);  writes L_j
//   side 1:  rows of U_j^T (-> Qt_j)
constexpr int kCrRowsMax = WB + WB + WR;   // 176
#ifdef BODYFIT_CR_STAMPS   // diagnostic build only (tools/ubench/cr_factor_phases.hip): s_memtime of wave 0 per phase
__device__ unsigned long long g_cr_stamps[64];
#define CR_STAMP(i)                                                                   \
  do {                                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                        \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
      g_cr_stamps[i] = t_;                                                            \
    }                                                                                 \
  } while (0)
#else
#define CR_STAMP(i)
#endif
// (the pointers of the kernel's first loads come as leading scalar arguments: with -mllvm -amdgpu-kernarg-preload-count they are
//  in SGPRs when the wave starts, one scalar round trip earlier than fields of the by-value struct)
__global__ __launch_bounds__(kCrThreads) void k_cr_factor(const int* __restrict__ elim, const double* __restrict__ Dp,
                                                          const double* __restrict__ Up, const double* __restrict__ Rtp,
                                                          int n_elim, WinBuf W) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* M = sm;                          // [kCrRowsMax][LD]
  double* invd = sm + kCrRowsMax * LD;     // [WB]
  double* stat = invd + WB;                // [1]
  const int e = blockIdx.x >> 1, side = blockIdx.x & 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = elim[3 * e], a = elim[3 * e + 1], b = elim[3 * e + 2];
  if (side == 1 && b < 0) return;
  CR_STAMP(0);
  const int nU = (side == 0) ? (a >= 0 ? WB : 0) : WB;       // appended coupling rows
  const int nApp = nU + (side == 0 ? WR : 0);
  const int nRows = WB + nApp;
  {
    // the node's blocks: all three requested before any lands in LDS (one round trip)
    BlockRegs<WB> rD, rU;
    BlockRegs<WR> rR;
    block_load<WB>(rD, Dp + (size_t)j * WB * WB, tid);
    if (nU) block_load<WB>(rU, Up + (size_t)(side == 0 ? a : j) * WB * WB, tid);
    if (side == 0) block_load<WR>(rR, Rtp + (size_t)j * WR * WB, tid);
    block_to_lds<WB, false, true>(rD, M, tid);
    if (nU) {
      if (side == 0) block_to_lds<WB>(rU, M + WB * LD, tid);              // row i of U_a
      else block_to_lds<WB, true>(rU, M + WB * LD, tid);                  // row i of U_j^T = column i of U_j
    }
    if (side == 0) block_to_lds<WR>(rR, M + (WB + nU) * LD, tid);
  }
  if (tid == 0) stat[0] = 1.0;
  __syncthreads();
  CR_STAMP(1);
  constexpr int NPAN = WB / 16;   // 5
  double* Linv = stat + 8;        // [16][17]: L_pp^-T of the current panel
  // (a) diagonal block p + identity below it, in the registers of wave 0, spread over all 64 lanes in the layout of an f64
  //     16 x 16 accumulator (dense_inl.h diag_factor16_acc): lane (m, kk), register q <-> row kk + 4 q, column m
  auto diag_block = [&](int p) {
    const int c0 = 16 * p;
    const int m = lane & 15, kk = lane >> 4;
    double av[4], bv[4], invc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // (the FULL symmetric block: only its lower triangle is kept up to date in LDS, mirror it)
      const int r = kk + 4 * q, lo = max(r, m), hi = min(r, m);
      av[q] = M[(c0 + lo) * LD + c0 + hi];
      bv[q] = (r == m) ? 1.0 : 0.0;
    }
    const bool okp = diag_factor16_acc(av, bv, lane, invc, min(16, NP - c0));
    double* Lg = W.Li + ((size_t)j * (WB / 16) + p) * 256;   // kept for the way down (k_cr_back)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = kk + 4 * q;
      if (r >= m) M[(c0 + r) * LD + c0 + m] = av[q];           // L (lower triangle)
      Linv[r * 17 + m] = bv[q];                                // row r of L_pp^-T (zero left of the diagonal)
      if (side == 0) Lg[r * 16 + m] = bv[q];
    }
    if (lane == 0 && !okp) stat[0] = 0.0;
  };
  if (wave == 0) diag_block(0);
  __syncthreads();
  CR_STAMP(2);
  const int nRowTiles = nRows / 16;
  for (int p = 0; p < NPAN; ++p) {
    const int c0 = 16 * p;
    // (b) panel solve on the matrix cores: every 16-row tile below the diagonal block  X = A L_pp^-T
    {
      const int m = lane & 15, kk = lane >> 4;
      for (int I = p + 1 + wave; I < nRowTiles; I += kCrWaves) {
        double a4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a4[s4] = M[(16 * I + m) * LD + c0 + 4 * s4 + kk];
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], Linv[(4 * s4 + kk) * 17 + m], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) M[(16 * I + kk + 4 * q) * LD + c0 + m] = acc[q];
      }
    }
    __syncthreads();
    CR_STAMP(3 + 4 * p);
    // (c) trailing update on the matrix cores: rows of tile I, columns of panel Kc > p:  M[I][Kc] -= X_I X_Kc^T.
    //     Look-ahead: wave 0 updates the next diagonal tile first and factors it at once (the serial part of a panel)
    //     while the other seven waves update the rest.
    if (p + 1 < NPAN) {
      const int m = lane & 15, kk = lane >> 4;
      auto tile_update = [&](int I, int Kc) {
        d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = M[(16 * I + kk + 4 * q) * LD + 16 * Kc + m];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double av = -M[(16 * I + m) * LD + c0 + 4 * s4 + kk];
          const double bv = M[(16 * Kc + m) * LD + c0 + 4 * s4 + kk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) M[(16 * I + kk + 4 * q) * LD + 16 * Kc + m] = acc[q];
      };
      // a whole row of tiles per wave: the row tile's own panel entries (A operand) are read once, every LDS read of the row
      // is issued before the first product, and the <= 4 column tiles are four independent accumulator chains (tile by tile,
      // each product waited for its two LDS reads and the previous product: ~790 cycles per tile against 256 of matrix work)
      auto row_update = [&](int I) {
        const int kc1 = (I < NPAN) ? I : NPAN - 1;                 // last column tile of this row
        double a4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a4[s4] = -M[(16 * I + m) * LD + c0 + 4 * s4 + kk];
        d4 acc[4];
        double b4[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int Kc = min(p + 1 + j, NPAN - 1);                 // (clamped: loads of unused tiles stay inside the block)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[j][q] = M[(16 * I + kk + 4 * q) * LD + 16 * Kc + m];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) b4[j][s4] = M[(16 * Kc + m) * LD + c0 + 4 * s4 + kk];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (p + 1 + j <= kc1) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], b4[j][s4], acc[j], 0, 0, 0);   // (uniform)
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p + 1 + j <= kc1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) M[(16 * I + kk + 4 * q) * LD + 16 * (p + 1 + j) + m] = acc[j][q];
          }
        }
      };
      if (wave == 0) {
        tile_update(p + 1, p + 1);
        CR_STAMP(4 + 4 * p);
        diag_block(p + 1);
        CR_STAMP(5 + 4 * p);
      } else {
        for (int I = p + 2 + (wave - 1); I < nRowTiles; I += kCrWaves - 1) row_update(I);
      }
    }
    __syncthreads();
    CR_STAMP(6 + 4 * p);
  }
  if (tid == 0 && stat[0] == 0.0) *W.fail = 1;
  if (side == 0) {
    block_from_lds<WB, true>(W.L + (size_t)j * WB * WB, M, tid);
    if (nU) block_from_lds<WB>(W.Pt + (size_t)j * WB * WB, M + WB * LD, tid);
    block_from_lds<WR>(W.Yt + (size_t)j * WR * WB, M + (WB + nU) * LD, tid);
  } else {
    block_from_lds<WB>(W.Qt + (size_t)j * WB * WB, M + WB * LD, tid);
  }
  CR_STAMP(24);
}

// C[ti][tj] (16 x 16 tile, accumulator layout: row = (lane >> 4) + 4 q, column = lane & 15) += sign * sum_k X[i][k] Y[i'][k]
// with X, Y staged in LDS (leading dimension LD), K = WB
__device__ __forceinline__ d4 tile_xyT(const double* X, const double* Y, int ti, int tj, int lane, d4 acc, double sign) {
  const int m = lane & 15, kk = lane >> 4;
#pragma unroll 4
  for (int s = 0; s < WB / 4; ++s) {
    const double av = sign * X[(16 * ti + m) * LD + 4 * s + kk];
    const double bv = Y[(16 * tj + m) * LD + 4 * s + kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}

// ---- cyclic reduction: Schur updates of one remaining frame a (left eliminated neighbour jl, right one jr, next
//      remaining frame b).  Four workgroups per frame: diagonal block (two halves of its tiles), coupling block, rhs. -----
__global__ __launch_bounds__(kCrThreads) void k_cr_update(const int* __restrict__ surv, int n_surv, int split,
                                                          double* __restrict__ Dp, const double* __restrict__ Qtp,
                                                          const double* __restrict__ Ptp, WinBuf W) {   // (leading scalars: k_cr_factor)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* X0 = sm;                 // [WB][LD]
  double* X1 = sm + WB * LD;       // [WB][LD]
  double* Ys = X1 + WB * LD;       // [2][WR][LD]
  // four workgroups per remaining frame: the diagonal block's 15 lower tiles in two halves (its 600 f64 matrix
  // instructions were the longest part by 2x: one tile per wave now), the coupling block, the right-hand sides
  // (split = 1, levels that do not fill the chip; on the throughput-bound levels of a long window one workgroup takes both
  //  halves: three workgroups per frame)
  const int sidx = split ? (int)(blockIdx.x >> 2) : (int)(blockIdx.x / 3), part4 = split ? (int)(blockIdx.x & 3) : -1;
  const int part = split ? (part4 < 2 ? 0 : part4 - 1) : (int)(blockIdx.x % 3);
  const int h0 = split ? (part4 & 1) : 0, h1 = split ? h0 + 1 : 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int a = surv[4 * sidx], jl = surv[4 * sidx + 1], jr = surv[4 * sidx + 2], b = surv[4 * sidx + 3];
  const int m = lane & 15, kk = lane >> 4;
  // Every operand of the part is requested before the first is used: the source blocks (13 loads per thread each) and the
  // accumulator tiles of the wave (read-modify-write of global memory), one round trip instead of one per trip / per tile.
  if (part == 0) {
    // D_a -= Qt_jl Qt_jl^T + Pt_jr Pt_jr^T  (15 lower tiles over 8 waves: tiles wave and wave + 8)
    double* D = Dp + (size_t)a * WB * WB;
    BlockRegs<WB> r0, r1;
    if (jl >= 0) block_load<WB>(r0, Qtp + (size_t)jl * WB * WB, tid);
    if (jr >= 0) block_load<WB>(r1, Ptp + (size_t)jr * WB * WB, tid);
    int tis[2], tjs[2];
    d4 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                         // this wave's tile of each half it carries
      const int t = min(wave + 8 * u, 14);
      int ti = 0, tj = t;
      while (tj > ti) { tj -= ti + 1; ++ti; }             // t -> (ti, tj) of the lower triangle, row-major
      tis[u] = ti; tjs[u] = tj;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[u][q] = (u >= h0 && u < h1) ? D[(size_t)(16 * ti + kk + 4 * q) * WB + 16 * tj + m] : 0.0;
    }
    if (jl >= 0) block_to_lds<WB>(r0, X0, tid);
    if (jr >= 0) block_to_lds<WB>(r1, X1, tid);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u < h0 || u >= h1 || wave + 8 * u > 14) continue;
      // the two sources as two independent accumulator chains
      d4 acc2 = {0.0, 0.0, 0.0, 0.0};
      if (jl >= 0) acc[u] = tile_xyT(X0, X0, tis[u], tjs[u], lane, acc[u], -1.0);
      if (jr >= 0) acc2 = tile_xyT(X1, X1, tis[u], tjs[u], lane, acc2, -1.0);
#pragma unroll
      for (int q = 0; q < 4; ++q) D[(size_t)(16 * tis[u] + kk + 4 * q) * WB + 16 * tjs[u] + m] = acc[u][q] + acc2[q];
    }
  } else if (part == 1) {
    // U_a := -Pt_jr Qt_jr^T  (coupling of a with the next remaining frame b)
    if (jr < 0 || b < 0) return;
    BlockRegs<WB> r0, r1;
    block_load<WB>(r0, W.Pt + (size_t)jr * WB * WB, tid);
    block_load<WB>(r1, W.Qt + (size_t)jr * WB * WB, tid);
    block_to_lds<WB>(r0, X0, tid);
    block_to_lds<WB>(r1, X1, tid);
    __syncthreads();
    double* U = W.U + (size_t)a * WB * WB;
    for (int t = wave; t < 25; t += kCrWaves) {
      const int ti = t / 5, tj = t % 5;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      acc = tile_xyT(X0, X1, ti, tj, lane, acc, -1.0);
#pragma unroll
      for (int q = 0; q < 4; ++q) U[(size_t)(16 * ti + kk + 4 * q) * WB + 16 * tj + m] = acc[q];
    }
  } else {
    // Rt_a -= Yt_jl Qt_jl^T + Yt_jr Pt_jr^T   ([16 x 80]: one column tile per wave, waves 0-4)
    double* Rt = W.Rt + (size_t)a * WR * WB;
    BlockRegs<WB> r0, r1;
    BlockRegs<WR> y0, y1;
    if (jl >= 0) { block_load<WB>(r0, W.Qt + (size_t)jl * WB * WB, tid); block_load<WR>(y0, W.Yt + (size_t)jl * WR * WB, tid); }
    if (jr >= 0) { block_load<WB>(r1, W.Pt + (size_t)jr * WB * WB, tid); block_load<WR>(y1, W.Yt + (size_t)jr * WR * WB, tid); }
    const int tj = min(wave, 4);
    d4 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Rt[(size_t)(kk + 4 * q) * WB + 16 * tj + m];
    if (jl >= 0) { block_to_lds<WB>(r0, X0, tid); block_to_lds<WR>(y0, Ys, tid); }
    if (jr >= 0) { block_to_lds<WB>(r1, X1, tid); block_to_lds<WR>(y1, Ys + WR * LD, tid); }
    __syncthreads();
    if (wave < 5) {
      if (jl >= 0) acc = tile_xyT(Ys, X0, 0, tj, lane, acc, -1.0);
      if (jr >= 0) acc = tile_xyT(Ys + WR * LD, X1, 0, tj, lane, acc, -1.0);
#pragma unroll
      for (int q = 0; q < 4; ++q) Rt[(size_t)(kk + 4 * q) * WB + 16 * tj + m] = acc[q];
    }
  }
}

// ---- cyclic reduction, way down: x_j = L_j^-T (Y_j - P_j x_a - Q_j x_b), 11 right-hand sides ---------------------------
// In the transposed storage:  Zt = Yt - Xt_a Pt - Xt_b Qt  ([16 x 80] = [16 x 80][80 x 80], f64 MFMA, B operands straight from
// L2: 16 consecutive doubles per lane group), then  Xt L = Zt  solved panel by panel from the last one: the products with
// the already known panels on the matrix cores, the 16 x 16 diagonal blocks through their explicit inverses (computed
// here, one block per wave, while the other waves form Zt).
__global__ __launch_bounds__(kCrThreads) void k_cr_back(const int* __restrict__ elim, const double* __restrict__ Lp,
                                                        const double* __restrict__ Xtp, int n_elim, WinBuf W) {   // (leading scalars: k_cr_factor)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Ls = sm;                     // [WB][LD]
  double* Zt = sm + WB * LD;           // [WR][LD]   right-hand sides, overwritten by the solution panel by panel
  double* Xa = Zt + WR * LD;           // [WR][LD]
  double* Xb = Xa + WR * LD;           // [WR][LD]
  double* Li = Xb + WR * LD;           // [5][16][17]  inverses of the diagonal blocks of L
  const int e = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = lane & 15, kk = lane >> 4;
  const int j = elim[3 * e], a = elim[3 * e + 1], b = elim[3 * e + 2];
  {
    BlockRegs<WB> rL;
    BlockRegs<WR> rA, rB;
    block_load<WB>(rL, Lp + (size_t)j * WB * WB, tid);
    if (a >= 0) block_load<WR>(rA, Xtp + (size_t)a * WR * WB, tid);
    if (b >= 0) block_load<WR>(rB, Xtp + (size_t)b * WR * WB, tid);
    block_to_lds<WB>(rL, Ls, tid);
    if (a >= 0) block_to_lds<WR>(rA, Xa, tid);
    if (b >= 0) block_to_lds<WR>(rB, Xb, tid);
  }
  __syncthreads();
  if (wave < 5) {
    // Zt tile (all 16 rows, columns 16 wave ..): accumulate -X P and -X Q on top of Yt
    const int tj = wave;
    const double* Yt = W.Yt + (size_t)j * WR * WB;
    d4 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Yt[(size_t)(kk + 4 * q) * WB + 16 * tj + m];
    for (int src = 0; src < 2; ++src) {
      const int nb_ = src == 0 ? a : b;
      if (nb_ < 0) continue;
      const double* G = (src == 0 ? W.Pt : W.Qt) + (size_t)j * WB * WB;
      const double* Xs = src == 0 ? Xa : Xb;
      double bv[WB / 4];
#pragma unroll
      for (int s4 = 0; s4 < WB / 4; ++s4) bv[s4] = G[(size_t)(4 * s4 + kk) * WB + 16 * tj + m];
#pragma unroll
      for (int s4 = 0; s4 < WB / 4; ++s4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xs[m * LD + 4 * s4 + kk], bv[s4], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Zt[(kk + 4 * q) * LD + 16 * tj + m] = acc[q];
  } else if (wave == 5 || wave == 6) {
    // inverses of the five 16 x 16 diagonal blocks of L, as k_cr_factor left them: (L_pp^-T)[c][r] = (L_pp^-1)[r][c]
    const double* Lg = W.Li + (size_t)j * (WB / 16) * 256;
    for (int it = tid - 320; it < 5 * 256; it += 128) {
      const int blk = it >> 8, c = (it >> 4) & 15, r = it & 15;
      Li[(blk * 16 + r) * 17 + c] = Lg[it];
    }
  }
  __syncthreads();
  // Xt[:, p] = (Zt[:, p] - sum_{q > p} Xt[:, q] L[q, p]) Linv_pp, panels from the last to the first; wave 0 only (each step
  // depends on the previous one; 4 + 4 (5 - p - 1) MFMAs per step)
  if (wave == 0) {
    for (int p = 4; p >= 0; --p) {
      d4 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = Zt[(kk + 4 * q) * LD + 16 * p + m];
      for (int qp = p + 1; qp < 5; ++qp) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Zt[m * LD + 16 * qp + 4 * s4 + kk], Ls[(16 * qp + 4 * s4 + kk) * LD + 16 * p + m],
                                                     acc, 0, 0, 0);
      }
      // through LDS: the accumulator tile becomes the A operand of the product with the inverse
#pragma unroll
      for (int q = 0; q < 4; ++q) Zt[(kk + 4 * q) * LD + 16 * p + m] = acc[q];
      d4 xo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        xo = __builtin_amdgcn_mfma_f64_16x16x4f64(Zt[m * LD + 16 * p + 4 * s4 + kk], Li[(p * 16 + 4 * s4 + kk) * 17 + m], xo, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) Zt[(kk + 4 * q) * LD + 16 * p + m] = xo[q];
    }
  }
  __syncthreads();
  double* Xt = W.Xt + (size_t)j * WR * WB;
  for (int idx = tid; idx < WR * WB; idx += kCrThreads) {
    const int c = idx / WB, k = idx % WB;
    Xt[idx] = (c <= NBETA) ? Zt[c * LD + k] : 0.0;
  }
}

// ---- beta Schur complement: per-frame partials, then the 10 x 10 solve ---------------------------------------------------
__global__ __launch_bounds__(128) void k_win_schur_part(WinProblem P, WinBuf W) {
  __shared__ double sBt[(NBETA + 1) * WB], sXt[(NBETA + 1) * WB];
  const int f = blockIdx.x, tid = threadIdx.x;
  const double* B = W.Rt0 + (size_t)f * WR * WB;     // scaled [B | rhs]^T as assembled
  const double* X = W.Xt + (size_t)f * WR * WB;
  {
    // both 11 x 80 operands into LDS, every load issued before the first is used
    constexpr int kN = (NBETA + 1) * WB, kPasses = (kN + 127) / 128;
    double bv[kPasses], xv[kPasses];
#pragma unroll
    for (int u = 0; u < kPasses; ++u) { const int e = min(tid + u * 128, kN - 1); bv[u] = B[e]; xv[u] = X[e]; }
#pragma unroll
    for (int u = 0; u < kPasses; ++u) { const int e = tid + u * 128; if (e < kN) { sBt[e] = bv[u]; sXt[e] = xv[u]; } }
  }
  __syncthreads();
  if (tid < NBETA * NBETA + NBETA) {
    const int a = (tid < NBETA * NBETA) ? tid / NBETA : tid - NBETA * NBETA;
    const int c = (tid < NBETA * NBETA) ? tid % NBETA : NBETA;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
    for (int i = 0; i + 1 < NP; i += 2) { s0 += sBt[a * WB + i] * sXt[c * WB + i]; s1 += sBt[a * WB + i + 1] * sXt[c * WB + i + 1]; }
    W.part[(size_t)f * kWinPart + tid] = s0 + s1;
  }
}

// mode 0: all; 1: this shard's sums of the Schur partials -> W.sred[110] only; 2: solve from W.sred (summed over the shards)
__global__ __launch_bounds__(1024) void k_win_beta_solve(WinProblem P, WinBuf W, const double* __restrict__ beta,
                                                         double* __restrict__ beta_new, int mode) {
  __shared__ double red[8][128];
  __shared__ double S[NBETA * NBETA], rb[NBETA];
  const int tid = threadIdx.x, F = P.F;
  if (P.nb == 0) return;
  // sums over the frames: 110 words, eight frame lanes of 128 threads, four independent loads per pass
  const int w = tid & 127, g = tid >> 7;
  if (mode != 2) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (w < NBETA * NBETA + NBETA) {
      int f = g;
      for (; f + 120 < F; f += 128) {        // sixteen loads in flight
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = W.part[(size_t)(f + 8 * u) * kWinPart + w];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
      }
      for (; f < F; f += 8) s0 += W.part[(size_t)f * kWinPart + w];
    }
    red[g][w] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < NBETA * NBETA + NBETA) {
    double s;
    if (mode == 2) s = W.sred[tid];
    else s = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
    if (mode == 1) W.sred[tid] = s;
    if (tid < NBETA * NBETA) S[tid] = W.Cs[tid] - s; else rb[tid - NBETA * NBETA] = W.rhsb[tid - NBETA * NBETA] - s;
  }
  if (mode == 1) return;
  __syncthreads();
  if (tid < 64) {
    // 10 x 10 Cholesky + both substitutions in the registers of one wave (dense_inl.h): lanes 0-15 the rows of S padded with
    // the identity, lane 16 the right-hand side as one more row (comes out as y^T = rb^T L^-T), lanes 32-47 the identity
    // (comes out as L^-T);  x = L^-T y
    const int rr = tid & 15, grp = tid >> 4;
    double av[16], iv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double v = 0.0;
      if (grp == 0) v = (rr < NBETA && k < NBETA) ? ((k <= rr) ? S[rr * NBETA + k] : 0.0) : (rr == k ? 1.0 : 0.0);
      else if (grp == 1) v = (rr == 0 && k < NBETA) ? rb[k] : 0.0;
      else if (grp == 2) v = (rr == k) ? 1.0 : 0.0;
      av[k] = v;
    }
    const bool ok = diag_factor16(av, rr, grp == 0, iv, NBETA);
    double x = 0.0;
#pragma unroll
    for (int k = 0; k < NBETA; ++k) x += av[k] * readlane_f64w(av[k], 16);     // lanes 32 + r: sum_k (L^-T)[r][k] y[k]
    if (grp == 2 && rr < NBETA) {
      const double dsb = ok ? x : 0.0;
      W.dsb[rr] = dsb;
      const double di = dsb * W.scale[(size_t)F * NP + rr];
      W.d[(size_t)F * NP + rr] = di;
      beta_new[rr] = beta[rr] + di;
    }
    if (tid == 0 && !ok) *W.fail = 1;
  }
}

// ---- step of one frame: d_f = S_f (x_f - X_B,f d_beta), candidate projected on the scale bounds --------------------------
__global__ __launch_bounds__(128) void k_win_step(WinProblem P, WinBuf W, const double* __restrict__ x,
                                                  double* __restrict__ x_new) {
  const int f = blockIdx.x, tid = threadIdx.x;
  if (tid >= NP) return;
  const double* X = W.Xt + (size_t)f * WR * WB;
  double xc[NBETA], dc[NBETA];
#pragma unroll
  for (int c = 0; c < NBETA; ++c) { xc[c] = X[c * WB + tid]; dc[c] = W.dsb[c]; }   // (dsb is zero past nb)
  double ds = X[NBETA * WB + tid];
  const double sc = W.scale[(size_t)f * NP + tid];
  const double xi = x[(size_t)f * NP + tid];
#pragma unroll
  for (int c = 0; c < NBETA; ++c) ds -= (c < P.nb) ? xc[c] * dc[c] : 0.0;
  double di = ds * sc;
  if (tid == 0) {
    const double s_new = fmin(fmax(xi + di, P.scale_lo), P.scale_hi);
    di = s_new - xi;
  }
  W.d[(size_t)f * NP + tid] = di;
  x_new[(size_t)f * NP + tid] = xi + di;
}
__global__ __launch_bounds__(128) void k_win_model(WinProblem P, WinBuf W, const double* __restrict__ x,
                                                   const double* __restrict__ d_halo) {
  __shared__ double sd[NP], sdn[NP], sdb[NBETA], red[2];
  const int f = blockIdx.x, tid = threadIdx.x, F = P.F;
  if (tid < NP) {
    sd[tid] = W.d[(size_t)f * NP + tid];
    sdn[tid] = (f + 1 < F) ? W.d[(size_t)(f + 1) * NP + tid] : ((P.halo && d_halo) ? d_halo[tid] : 0.0);
  }
  if (tid < NBETA) sdb[tid] = (tid < P.nb) ? W.d[(size_t)F * NP + tid] : 0.0;
  __syncthreads();
  double pm = 0.0, dn = 0.0, xn = 0.0;
  if (tid < NP) {
    // A is symmetric: thread i walks COLUMN i (consecutive threads read consecutive words), 19 loads in flight per batch
    const double* Ac = W.Araw + (size_t)f * NP * NP + tid;
    double hd = 0.0, h1 = 0.0;
#pragma unroll 1
    for (int jb = 0; jb < NP; jb += 19) {
      double aw[19];
#pragma unroll
      for (int u = 0; u < 19; ++u) aw[u] = Ac[(size_t)(jb + u) * NP];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u + 1 < 19; u += 2) { hd += aw[u] * sd[jb + u]; h1 += aw[u + 1] * sd[jb + u + 1]; }
      hd += aw[18] * sd[jb + 18];
    }
    hd += h1;
    double bw[NBETA];
#pragma unroll
    for (int c = 0; c < NBETA; ++c) bw[c] = W.Braw[((size_t)f * NP + tid) * NBETA + c];
#pragma unroll
    for (int c = 0; c < NBETA; ++c) hd += 2.0 * bw[c] * sdb[c];
    hd += 2.0 * W.Eraw[(size_t)f * NP + tid] * sdn[tid];
    pm = -sd[tid] * W.graw[(size_t)f * NP + tid] - 0.5 * sd[tid] * hd;
    dn = sd[tid] * sd[tid];
    const double xv = x[(size_t)f * NP + tid];
    xn = xv * xv;
  }
  pm = block_sum_n(pm, red, tid, 2);
  dn = block_sum_n(dn, red, tid, 2);
  xn = block_sum_n(xn, red, tid, 2);
  if (tid == 0) {
    double* o = W.part + (size_t)f * kWinPart + 112;
    o[0] = pm; o[1] = dn; o[2] = xn;
  }
}

// ---- decide: gradient tolerance, failed factorisation, parameter tolerance, or a candidate --------------------------------
// mode 0: all; 1: this shard's sums -> W.fin[0..3] = {model, |d|^2, |x|^2, max |g_frames|}, W.fin[4] = fail flag, only;
// 2: decide from W.fin (the first three summed, the last two maximised over the shards)
// the decision of an iteration from the sums over the frames (256 threads): gradient tolerance, failed factorisation,
// parameter tolerance, or a candidate
__device__ __forceinline__ void finish_core(const WinProblem& P, const WinBuf& W, const double* __restrict__ x,
                                            const double* __restrict__ beta, double* __restrict__ x_new,
                                            double* __restrict__ beta_new, double pm, double dn, double xn, double gm, int tid) {
  const int F = P.F;
  if (tid == 0) gm = fmax(gm, W.gmaxp[F]);
  // the beta block's operands into LDS first (one round trip; thread 0 walking global memory made this an 11 us kernel)
  __shared__ double sCr[NBETA * NBETA], sdb2[NBETA], sgb[NBETA], sbt[NBETA];
  if (tid < NBETA * NBETA) sCr[tid] = W.Craw[tid];
  if (tid >= 128 && tid < 128 + NBETA) {
    const int a = tid - 128;
    sdb2[a] = (a < P.nb) ? W.d[(size_t)F * NP + a] : 0.0;
    sgb[a] = W.gbraw[a];
    sbt[a] = (a < P.nb) ? beta[a] : 0.0;
  }
  __syncthreads();
  __shared__ int no_cand;
  if (tid == 0) {
    double* st = W.status;
    for (int a = 0; a < P.nb; ++a) {
      const double da = sdb2[a];
      pm -= da * sgb[a];
      double h = 0.0;
      for (int c = 0; c < P.nb; ++c) h += sCr[a * NBETA + c] * sdb2[c];
      pm -= 0.5 * da * h;
      dn += da * da;
      xn += sbt[a] * sbt[a];
    }
    st[kWsGmax] = gm;
    st[kWsHasCand] = 0.0;
    no_cand = 1;
    if (st[kWsActive] != 0.0) {
      if (gm <= 1e-10) {                                   // Ceres gradient_tolerance
        st[kWsActive] = 0.0; st[kWsTermination] = 0.0;
      } else if (*W.fail) {                                // the damped system was not positive definite
        const double rad = st[kWsRadius] / st[kWsDec];
        st[kWsRadius] = rad; st[kWsDec] *= 2.0; st[kWsBad] += 1.0; st[kWsIters] += 1.0;
        st[kWsAccepted] = 0.0;
        if (rad < 1e-32) { st[kWsActive] = 0.0; st[kWsTermination] = 2.0; }
      } else if (sqrt(dn) <= 1e-8 * (sqrt(xn) + 1e-8)) {   // Ceres parameter_tolerance
        st[kWsActive] = 0.0; st[kWsTermination] = 0.0;
      } else {
        st[kWsModel] = pm; st[kWsHasCand] = 1.0;
        no_cand = 0;
      }
    }
  }
  __syncthreads();
  if (no_cand) {   // the residual sweep that follows still reads a well-defined point
    copy_batched(x_new, x, F * NP, tid, 256);
    if (tid < P.nb) beta_new[tid] = beta[tid];
  }
}

__global__ __launch_bounds__(256) void k_win_finish(WinProblem P, WinBuf W, const double* __restrict__ x,
                                                    const double* __restrict__ beta, double* __restrict__ x_new,
                                                    double* __restrict__ beta_new, int mode) {
  __shared__ double red[4];
  const int tid = threadIdx.x, F = P.F;
  double pm = 0.0, dn = 0.0, xn = 0.0, gm = 0.0;
  if (mode != 2) {
    for (int f = tid; f < F; f += 256) {
      const double* o = W.part + (size_t)f * kWinPart + 112;
      pm += o[0]; dn += o[1]; xn += o[2];
      gm = fmax(gm, W.gmaxp[f]);
    }
    pm = block_sum_n(pm, red, tid, 4);
    dn = block_sum_n(dn, red, tid, 4);
    xn = block_sum_n(xn, red, tid, 4);
    gm = block_max_n(gm, red, tid, 4);
    if (mode == 1) {
      if (tid == 0) { W.fin[0] = pm; W.fin[1] = dn; W.fin[2] = xn; W.fin[3] = gm; W.fin[4] = *W.fail ? 1.0 : 0.0; }
      return;
    }
  } else {
    pm = W.fin[0]; dn = W.fin[1]; xn = W.fin[2]; gm = W.fin[3];
    if (tid == 0 && W.fin[4] != 0.0) *W.fail = 1;
  }
  finish_core(P, W, x, beta, x_new, beta_new, pm, dn, xn, gm, tid);
}

// ---- single-GPU tail of an iteration in ONE launch: the step of every frame (k_win_step), its share of the model cost
//      change (k_win_model; the next frame's step, which the temporal term needs, is recomputed here instead of read), and —
//      by the LAST workgroup to finish, found by a ticket — the decision (k_win_finish).  Two launch floors (~5 us each) and
//      their boundaries less per iteration.  Sharded solves exchange boundary rows between these steps and keep the three
//      kernels. ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_tail(WinProblem P, WinBuf W, const double* __restrict__ x,
                                                  const double* __restrict__ beta, double* __restrict__ x_new,
                                                  double* __restrict__ beta_new) {
  __shared__ double sd[2][NP], sdb[NBETA], red[4];
  __shared__ int s_last;
  const int f = blockIdx.x, tid = threadIdx.x, F = P.F;
  {
    // d of frame f (threads 0..75, stored) and of frame f + 1 (threads 128..203, kept here)
    const int h = tid >> 7, i = tid & 127, ff = f + h;
    if (i < NP) {
      double di = 0.0;
      if (ff < F) {
        const double* X = W.Xt + (size_t)ff * WR * WB;
        double xc[NBETA], dc[NBETA];
#pragma unroll
        for (int c = 0; c < NBETA; ++c) { xc[c] = X[c * WB + i]; dc[c] = W.dsb[c]; }
        double ds = X[NBETA * WB + i];
        const double sc = W.scale[(size_t)ff * NP + i];
        const double xi = x[(size_t)ff * NP + i];
#pragma unroll
        for (int c = 0; c < NBETA; ++c) ds -= (c < P.nb) ? xc[c] * dc[c] : 0.0;
        di = ds * sc;
        if (i == 0) {
          const double s_new = fmin(fmax(xi + di, P.scale_lo), P.scale_hi);
          di = s_new - xi;
        }
        if (h == 0) {
          W.d[(size_t)f * NP + i] = di;
          x_new[(size_t)f * NP + i] = xi + di;
        }
      }
      sd[h][i] = di;
    }
    if (tid >= 224 && tid < 224 + NBETA) sdb[tid - 224] = (tid - 224 < P.nb) ? W.d[(size_t)F * NP + tid - 224] : 0.0;
  }
  __syncthreads();
  double pm = 0.0, dn = 0.0, xn = 0.0;
  if (tid < NP) {
    const double* Ac = W.Araw + (size_t)f * NP * NP + tid;
    double hd = 0.0, h1 = 0.0;
#pragma unroll 1
    for (int jb = 0; jb < NP; jb += 19) {
      double aw[19];
#pragma unroll
      for (int u = 0; u < 19; ++u) aw[u] = Ac[(size_t)(jb + u) * NP];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u + 1 < 19; u += 2) { hd += aw[u] * sd[0][jb + u]; h1 += aw[u + 1] * sd[0][jb + u + 1]; }
      hd += aw[18] * sd[0][jb + 18];
    }
    hd += h1;
    double bw[NBETA];
#pragma unroll
    for (int c = 0; c < NBETA; ++c) bw[c] = W.Braw[((size_t)f * NP + tid) * NBETA + c];
#pragma unroll
    for (int c = 0; c < NBETA; ++c) hd += 2.0 * bw[c] * sdb[c];
    hd += 2.0 * W.Eraw[(size_t)f * NP + tid] * sd[1][tid];
    pm = -sd[0][tid] * W.graw[(size_t)f * NP + tid] - 0.5 * sd[0][tid] * hd;
    dn = sd[0][tid] * sd[0][tid];
    const double xv = x[(size_t)f * NP + tid];
    xn = xv * xv;
  }
  pm = block_sum_n(pm, red, tid, 4);
  dn = block_sum_n(dn, red, tid, 4);
  xn = block_sum_n(xn, red, tid, 4);
  if (tid == 0) {
    // the partial is at the memory side before the ticket is taken: write-through stores and this wave's vmcnt(0) (cdna guide,
    // Guideline 16 R1), not an agent-scope release fence (buffer_wbl2 writes back every dirty line of the XCD's L2)
    double* o = W.part + (size_t)f * kWinPart + 112;
    store_f64_through(o, pm); store_f64_through(o + 1, dn); store_f64_through(o + 2, xn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // two-level ticket: groups of 32 frames, then the groups (atomics on ONE word serialise at ~40 ns each: 1024 of them
    // were two thirds of this kernel at 1024 frames)
    const int grp = f >> 5, ngrp = (F + 31) >> 5, gsize = min(32, F - 32 * grp);
    int last = 0;
    if (atomicAdd(W.ticket + 1 + grp, 1) == gsize - 1) {
      __hip_atomic_store(W.ticket + 1 + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next launch)
      last = (atomicAdd(W.ticket, 1) == ngrp - 1) ? 1 : 0;
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: every frame's partial is at the memory side (sc1 loads: past this XCD's L2), decide ----
  if (tid == 0) __hip_atomic_store(W.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  pm = 0.0; dn = 0.0; xn = 0.0;
  double gm = 0.0;
  for (int g = tid; g < F; g += 256) {
    const double* o = W.part + (size_t)g * kWinPart + 112;
    pm += __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dn += __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xn += __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gm = fmax(gm, W.gmaxp[g]);
  }
  pm = block_sum_n(pm, red, tid, 4);
  dn = block_sum_n(dn, red, tid, 4);
  xn = block_sum_n(xn, red, tid, 4);
  gm = block_max_n(gm, red, tid, 4);
  finish_core(P, W, x, beta, x_new, beta_new, pm, dn, xn, gm, tid);
}

// ---- accept / reject the candidate (Ceres' step quality and radius rules, host_solver.cpp) ------------------------------
// the decision itself, for a candidate that exists (status HasCand), given the candidate's cost; thread 0 decides, every
// thread then copies the accepted point.  Returns (to every thread) whether the candidate was accepted.
__device__ __forceinline__ bool accept_core(const WinProblem& P, const WinBuf& W, double* __restrict__ x, double* __restrict__ beta,
                                            const double* __restrict__ x_new, const double* __restrict__ beta_new, double new_cost,
                                            int tid, int nthreads) {
  __shared__ int acc_flag;
  double* st = W.status;
  if (tid == 0) {
    const double cost = st[kWsCost], model = st[kWsModel];
    const double change = cost - new_cost, rho = change / model;
    const bool accept = (new_cost == new_cost) && new_cost < 1e300 && model > 0.0 && rho > 1e-3;
    st[kWsIters] += 1.0;
    st[kWsNewCost] = new_cost;
    if (accept) {
      st[kWsCost] = new_cost;
      const double t = 2.0 * rho - 1.0;
      st[kWsRadius] = fmin(1e16, st[kWsRadius] / fmax(1.0 / 3.0, 1.0 - t * t * t));
      st[kWsDec] = 2.0;
      st[kWsOk] += 1.0;
      if (fabs(change) < 1e-6 * cost) { st[kWsActive] = 0.0; st[kWsTermination] = 0.0; }   // function_tolerance
    } else {
      const double rad = st[kWsRadius] / st[kWsDec];
      st[kWsRadius] = rad; st[kWsDec] *= 2.0; st[kWsBad] += 1.0;
      if (rad < 1e-32) { st[kWsActive] = 0.0; st[kWsTermination] = 2.0; }
    }
    st[kWsAccepted] = accept ? 1.0 : 0.0;
    st[kWsJsel] = accept ? 1.0 : 2.0;     // (single-GPU loop: the candidate sweep also left the candidate's Jacobian)
    st[kWsHasCand] = 0.0;
    acc_flag = accept ? 1 : 0;
  }
  __syncthreads();
  if (acc_flag) {
    copy_batched(x, x_new, P.F * NP, tid, nthreads);
    if (tid < P.nb) beta[tid] = beta_new[tid];
  }
  return acc_flag != 0;
}
// mode 0: all; 1: this shard's cost at the candidate -> W.fin[0] only; 2: decide with W.fin[0] (summed over the shards);
// 3: this shard's cost at the candidate -> W.fin[5] only (sharded solves: the decision is k_win_decide's)
__global__ __launch_bounds__(1024) void k_win_accept(WinProblem P, WinBuf W, const double* __restrict__ r_new,
                                                     double* __restrict__ x, double* __restrict__ beta,
                                                     const double* __restrict__ x_new, const double* __restrict__ beta_new,
                                                     int mode) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  double* st = W.status;
  if (mode == 1 || mode == 3) {
    const double c = window_cost_any(P, r_new, red, tid, 1024);
    if (tid == 0) W.fin[mode == 1 ? 0 : 5] = c;
    return;
  }
  if (st[kWsHasCand] == 0.0) {
    if (tid == 0) st[kWsJsel] = 2.0;     // no candidate was produced (inactive solve, failed factorisation): nothing moved
    return;
  }
  const double new_cost = (mode == 2) ? W.fin[0] : window_cost_any(P, r_new, red, tid, 1024);
  (void)accept_core(P, W, x, beta, x_new, beta_new, new_cost, tid, 1024);
}

// ---- sharded solves (bodyfit_solve_sharded*): what the exchanges need ----------------------------------------------------
// sum of the shards' partials in rank order: out[i] = sum_r gathered[r][i] (every rank computes bit-identical totals)
__global__ __launch_bounds__(256) void k_sum_ranks(const double* __restrict__ g, int N, int stride, int n, double* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double s = 0.0;
    for (int r = 0; r < N; ++r) s += g[(size_t)r * stride + i];
    out[i] = s;
  }
}
// shard proxy (bodyfit_set_shard_proxy, a measurement aid): the one-rank all-gather has filled slot 0; the other N - 1 slots
// get copies, as if N identical shards had contributed
__global__ __launch_bounds__(256) void k_replicate_ranks(double* __restrict__ g, int n, int N) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double v = g[i];
    for (int r = 1; r < N; ++r) g[(size_t)r * n + i] = v;
  }
}
// this shard's contribution to the interface system: [D_first, D_last, U_first, U_last | Rt_first, Rt_last | extra], one
// contiguous buffer for ONE all-gather
__global__ __launch_bounds__(256) void k_iface_pack(WinBuf W, int F, const double* __restrict__ extra, int n_extra,
                                                    double* __restrict__ send) {
  constexpr int blk = WB * WB, rhs = WR * WB;
  const int fl = F - 1, total = 4 * blk + 2 * rhs + n_extra;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    double v;
    if (i < blk) v = W.D[i];
    else if (i < 2 * blk) v = W.D[(size_t)fl * blk + (i - blk)];
    else if (i < 3 * blk) v = W.U[i - 2 * blk];
    else if (i < 4 * blk) v = W.U[(size_t)fl * blk + (i - 3 * blk)];
    else if (i < 4 * blk + rhs) v = W.Rt[i - 4 * blk];
    else if (i < 4 * blk + 2 * rhs) v = W.Rt[(size_t)fl * rhs + (i - 4 * blk - rhs)];
    else v = extra[i - 4 * blk - 2 * rhs];
    send[i] = v;
  }
}
// the gathered contributions -> the interface chain of 2 N frames (every rank builds the same), the extras summed in rank order
__global__ __launch_bounds__(256) void k_iface_unpack(WinBuf Wi, const double* __restrict__ g, int N, int n_extra,
                                                      double* __restrict__ extra_sum) {
  constexpr int blk = WB * WB, rhs = WR * WB;
  const int per = 4 * blk + 2 * rhs + n_extra;
  const int total = N * (4 * blk + 2 * rhs) + n_extra;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (i < N * (4 * blk + 2 * rhs)) {
      const int r = i / (4 * blk + 2 * rhs), j = i - r * (4 * blk + 2 * rhs);
      const double v = g[(size_t)r * per + j];
      if (j < 2 * blk) Wi.D[(size_t)(2 * r) * blk + j] = v;
      else if (j < 4 * blk) Wi.U[(size_t)(2 * r) * blk + (j - 2 * blk)] = v;
      else Wi.Rt[(size_t)(2 * r) * rhs + (j - 4 * blk)] = v;
    } else {
      const int e = i - N * (4 * blk + 2 * rhs);
      double s = 0.0;
      for (int r = 0; r < N; ++r) s += g[(size_t)r * per + 4 * blk + 2 * rhs + e];
      extra_sum[e] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *Wi.fail = 0;
}
// The steps of the neighbouring shards' boundary frames, computed HERE from the interface solution every rank holds (node
// 2 r = first frame of shard r, 2 r + 1 = its last): the same arithmetic as k_win_step on the same numbers, so the row this
// rank keeps of its neighbour's frame is bit-identical to the neighbour's own.  block 0: the next shard's first frame (the
// halo row of the temporal pair this shard owns), block 1: the previous shard's last frame.
__global__ __launch_bounds__(128) void k_win_halo_step(WinProblem P, const double* __restrict__ Xi, const double* __restrict__ dsb,
                                                       int node_right, const double* __restrict__ scale_right,
                                                       const double* __restrict__ x_right, double* __restrict__ d_right,
                                                       double* __restrict__ xn_right, int node_left,
                                                       const double* __restrict__ scale_left, const double* __restrict__ x_left,
                                                       double* __restrict__ xn_left) {
  const int tid = threadIdx.x;
  if (tid >= NP) return;
  const bool right = blockIdx.x == 0;
  const int node = right ? node_right : node_left;
  if (node < 0) return;
  const double* X = Xi + (size_t)node * WR * WB;
  double ds = X[NBETA * WB + tid];
#pragma unroll
  for (int c = 0; c < NBETA; ++c) ds -= (c < P.nb) ? X[c * WB + tid] * dsb[c] : 0.0;
  const double sc = (right ? scale_right : scale_left)[tid];
  const double xi = (right ? x_right : x_left)[tid];
  double di = ds * sc;
  if (tid == 0) {
    const double s_new = fmin(fmax(xi + di, P.scale_lo), P.scale_hi);
    di = s_new - xi;
  }
  if (right) { d_right[tid] = di; xn_right[tid] = xi + di; }
  else xn_left[tid] = xi + di;
}
// a failed interface factorisation (every rank factors the same chain) is this shard's failure too
__global__ void k_win_fold_fail(WinBuf W, WinBuf Wi) {
  if (threadIdx.x == 0 && *Wi.fail) *W.fail = 1;
}
// The whole decision of a sharded iteration in one launch, from every shard's partials [model, |d|^2, |x|^2, max |g|, fail,
// cost at the candidate] (gathered, [N][8]): k_win_finish's tests, then — if there is a candidate — k_win_accept's.  Every
// rank runs it on the same numbers.  On acceptance the rows this rank keeps of its neighbours' boundary frames move too.
__global__ __launch_bounds__(256) void k_win_decide(WinProblem P, WinBuf W, double* __restrict__ x, double* __restrict__ beta,
                                                    double* __restrict__ x_new, double* __restrict__ beta_new,
                                                    const double* __restrict__ g, int N, double* __restrict__ x_halo,
                                                    const double* __restrict__ xn_halo, double* __restrict__ x_left,
                                                    const double* __restrict__ xn_left) {
  const int tid = threadIdx.x;
  double pm = 0.0, dn = 0.0, xn = 0.0, gm = 0.0, fl = 0.0, cost = 0.0, poison = 0.0;
  for (int r = 0; r < N; ++r) {
    const double* o = g + (size_t)r * 8;
    pm += o[0]; dn += o[1]; xn += o[2];
    gm = fmax(gm, o[3]); fl = fmax(fl, o[4]);
    cost += o[5];
    poison = fmax(poison, o[6]);
  }
  if (poison != 0.0) {
    // A rank could not produce its part of this iteration (a failed launch / HIP call: bodyfit_api.hip puts a 1 in slot 6 of its
    // scalars and keeps taking part in the exchanges).  Every rank reads the same gathered scalars, so every rank ends the solve
    // HERE, in the same iteration: nothing moves, the host loops find the solve inactive at their next status read and return.
    if (tid == 0) {
      W.status[kWsActive] = 0.0; W.status[kWsTermination] = 2.0; W.status[kWsHasCand] = 0.0; W.status[kWsJsel] = 2.0;
      W.status[kWsPoison] = poison;
    }
    return;
  }
  if (tid == 0 && fl != 0.0) *W.fail = 1;
  __syncthreads();
  finish_core(P, W, x, beta, x_new, beta_new, pm, dn, xn, gm, tid);
  __syncthreads();
  if (W.status[kWsHasCand] == 0.0) {
    if (tid == 0) W.status[kWsJsel] = 2.0;   // nothing moved
    return;
  }
  const bool accepted = accept_core(P, W, x, beta, x_new, beta_new, cost, tid, 256);
  if (accepted && tid < NP) {
    if (x_halo) x_halo[tid] = xn_halo[tid];
    if (x_left) x_left[tid] = xn_left[tid];
  }
}

}  // namespace

size_t win_factor_lds_bytes() { return (size_t)(kCrRowsMax * LD + WB + 8 + 16 * 17) * sizeof(double); }
size_t win_update_lds_bytes() { return (size_t)(2 * WB * LD + 2 * WR * LD) * sizeof(double); }
size_t win_back_lds_bytes() { return (size_t)(WB * LD + 3 * WR * LD + 5 * 16 * 17) * sizeof(double); }

void launch_win_init(const WinProblem& P, const WinBuf& W, const double* d_r, int mode, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_init, dim3(1), dim3(1024), 0, s, P, W, d_r, mode);
}
void launch_win_beta(const WinProblem& P, const WinBuf& W, const double* d_Hpan, const double* d_r, int first, int mode,
                     hipStream_t s) {
  BODYFIT_LAUNCH(k_win_beta, dim3(1), dim3(1024), 0, s, P, W, d_Hpan, d_r, first, mode);
}
void launch_win_assemble(const WinProblem& P, const WinBuf& W, const double* d_Hpan, const double* d_r, const double* d_x,
                         const unsigned char* d_constant, int first, const double* d_x_left, const double* d_scale_halo,
                         hipStream_t s) {
  BODYFIT_LAUNCH(k_win_assemble, dim3(P.F), dim3(kAsmThreads), 0, s, P, W, d_Hpan, d_r, d_x, d_constant, first, d_x_left,
                     d_scale_halo);
}
void launch_cr_factor(const WinBuf& W, const int* d_elim, int n_elim, hipStream_t s) {
  static DeviceOnce attr;
  attr.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cr_factor), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)win_factor_lds_bytes());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cr_update), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)win_update_lds_bytes());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cr_back), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)win_back_lds_bytes());
  });
  if (n_elim > 0)
    BODYFIT_LAUNCH(k_cr_factor, dim3(2 * n_elim), dim3(kCrThreads), win_factor_lds_bytes(), s, d_elim, W.D, W.U, W.Rt, n_elim, W);
}
void launch_cr_update(const WinBuf& W, const int* d_surv, int n_surv, hipStream_t s) {
  if (n_surv > 0) {
    const int split = 4 * n_surv <= 256 ? 1 : 0;        // (the diagonal block in two workgroups where CUs are idle anyway)
    BODYFIT_LAUNCH(k_cr_update, dim3((split ? 4 : 3) * n_surv), dim3(kCrThreads), win_update_lds_bytes(), s, d_surv, n_surv, split,
                   W.D, W.Qt, W.Pt, W);
  }
}
void launch_cr_back(const WinBuf& W, const int* d_elim, int n_elim, hipStream_t s) {
  if (n_elim > 0) BODYFIT_LAUNCH(k_cr_back, dim3(n_elim), dim3(kCrThreads), win_back_lds_bytes(), s, d_elim, W.L, W.Xt, n_elim, W);
}
void launch_win_schur_part(const WinProblem& P, const WinBuf& W, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_schur_part, dim3(P.F), dim3(128), 0, s, P, W);
}
void launch_win_beta_solve(const WinProblem& P, const WinBuf& W, const double* d_beta, double* d_beta_new, int mode, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_beta_solve, dim3(1), dim3(1024), 0, s, P, W, d_beta, d_beta_new, mode);
}
void launch_win_tail(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_beta, double* d_x_new, double* d_beta_new,
                     hipStream_t s) {
  BODYFIT_LAUNCH(k_win_tail, dim3(P.F), dim3(256), 0, s, P, W, d_x, d_beta, d_x_new, d_beta_new);
}
void launch_win_step(const WinProblem& P, const WinBuf& W, const double* d_x, double* d_x_new, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_step, dim3(P.F), dim3(128), 0, s, P, W, d_x, d_x_new);
}
void launch_win_model(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_halo_step, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_model, dim3(P.F), dim3(128), 0, s, P, W, d_x, d_halo_step);
}
void launch_win_finish(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_beta, double* d_x_new,
                       double* d_beta_new, int mode, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_finish, dim3(1), dim3(256), 0, s, P, W, d_x, d_beta, d_x_new, d_beta_new, mode);
}
void launch_win_accept(const WinProblem& P, const WinBuf& W, const double* d_r_new, double* d_x, double* d_beta,
                       const double* d_x_new, const double* d_beta_new, int mode, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_accept, dim3(1), dim3(1024), 0, s, P, W, d_r_new, d_x, d_beta, d_x_new, d_beta_new, mode);
}

}  // namespace bodyfit

namespace bodyfit {
void launch_sum_ranks(const double* d_g, int N, int stride, int n, double* d_out, hipStream_t s) {
  BODYFIT_LAUNCH(k_sum_ranks, dim3((n + 255) / 256), dim3(256), 0, s, d_g, N, stride, n, d_out);
}
void launch_replicate_ranks(double* d_g, int n, int N, hipStream_t s) {
  BODYFIT_LAUNCH(k_replicate_ranks, dim3(std::min(64, (n + 255) / 256)), dim3(256), 0, s, d_g, n, N);
}
int iface_doubles(int n_extra) { return 4 * WB * WB + 2 * WR * WB + n_extra; }
void launch_iface_pack(const WinBuf& W, int F, const double* d_extra, int n_extra, double* d_send, hipStream_t s) {
  BODYFIT_LAUNCH(k_iface_pack, dim3(64), dim3(256), 0, s, W, F, d_extra, n_extra, d_send);
}
void launch_iface_unpack(const WinBuf& Wi, const double* d_g, int N, int n_extra, double* d_extra_sum, hipStream_t s) {
  BODYFIT_LAUNCH(k_iface_unpack, dim3(64), dim3(256), 0, s, Wi, d_g, N, n_extra, d_extra_sum);
}
void launch_win_halo_step(const WinProblem& P, const double* d_Xi, const double* d_dsb, int node_right, const double* d_scale_right,
                          const double* d_x_right, double* d_d_right, double* d_xn_right, int node_left,
                          const double* d_scale_left, const double* d_x_left, double* d_xn_left, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_halo_step, dim3(2), dim3(128), 0, s, P, d_Xi, d_dsb, node_right, d_scale_right, d_x_right, d_d_right,
                     d_xn_right, node_left, d_scale_left, d_x_left, d_xn_left);
}
void launch_win_fold_fail(const WinBuf& W, const WinBuf& Wi, hipStream_t s) { BODYFIT_LAUNCH(k_win_fold_fail, dim3(1), dim3(64), 0, s, W, Wi); }
void launch_win_decide(const WinProblem& P, const WinBuf& W, double* d_x, double* d_beta, double* d_x_new, double* d_beta_new,
                       const double* d_g, int N, double* d_x_halo, const double* d_xn_halo, double* d_x_left,
                       const double* d_xn_left, hipStream_t s) {
  BODYFIT_LAUNCH(k_win_decide, dim3(1), dim3(256), 0, s, P, W, d_x, d_beta, d_x_new, d_beta_new, d_g, N, d_x_halo, d_xn_halo,
                     d_x_left, d_xn_left);
}
}  // namespace bodyfit
