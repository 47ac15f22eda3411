// frame_part_inl.h — per-frame keypoint residuals + analytic Jacobian, f64, one 8-wave workgroup per frame: the body
// shared by k_frame_resjac (its own launch) and the frame role of k_sweep_roles (one launch for the whole sweep: frame, mesh
// and prior workgroups side by side; kFused = true: the mesh operands and the reduction partial are handed over inside the
// launch, write-through), both in k_sweep.hip.
//
// Replaces, for every reprojection block of a frame at once, what the reference evaluates through
// ceres::DynamicAutoDiffCostFunction<ReprojCost[Shape]> (include/Sim3BA.h:34-88,126-227,420,581;
// include/MultiFrameBA.h:85-102): 22 dual-number passes per 2-residual block become one closed-form
// Jacobian.  The same workgroup also prepares the operands of the mesh kernel (blend-coefficient fragments in
// bf16 hi/lo, 24 skinning transforms) so the two kernels share one Rodrigues pass, and, for shared-beta problems,
// its frame's share of the [cost, g_beta, H_bb] reduction.
//
// Work layout (512 threads = 8 wavefronts per frame, two per SIMD; <= 128 VGPRs so that two workgroups share a CU at
// large frame counts; phases separated by workgroup barriers, all intermediate state in LDS, about 73 KB).  At one
// frame per CU the kernel is a latency chain, so every phase is organised around having its loads in flight together
// (fixed-trip predicated batches; a runtime-bounded loop of loads pays one L2 round trip per iteration) and around
// keeping all eight waves busy:
//   A  model tables, landmark weights, the frame's parameters -> LDS
//   B  wave 0: Rodrigues R_j, dR_j/da (both branches of Ceres' AngleAxisRotatePoint) | waves 1-3: chain offsets
//      o_j(beta), centred rest joints | wave 4: landmark rest vertices
//   C  waves 6-7: A_j columns and P_j as independent 3-vector walks up the kinematic chain (one branch-free pass over packed
//      ancestor lists), wave 6 also the camera matrices | wave 0 first: root entries, the mesh's blend-coefficient fragments |
//      waves 0-5: landmark items (landmark l, joint k) on lane k - 1 of half-wave l: the landmark's 27 posedirs values are
//      read ONCE (joint-minor table, coalesced) and give the blend row (half-wave shuffle reduction) and the Jacobian inner
//      products pd . vec(dR_{k,c}) | wave 7, behind an LDS counter (not a barrier): skinning transforms and posed joints as
//      (joint, row) items, and — one-launch sweep — the hand-off: drain, one agent-scope add to its 32-frame unit's counter
//   D  W_{k,c} = A_par(k) dR_{k,c} R_k^T A_par(k)^T (d x / d a_{k,c} = W (x - P_k)) | landmark LBS | waves 3-6: B_j columns
//      (d P_j / d beta) and T_j = B_j - A_j Sc_j
//   E  complete landmark terms d q_l / d theta_{k,c} per (landmark, joint) and d q_l / d beta = Ablend sd_l + sum_i w_i T_{j_i}
//   F  per chunk of 32 keypoints: keypoint stage (projection, residuals, d pi), then the Jacobian sweep with
//      thread = column (W_{k,c} and P_k in registers), consecutive threads on consecutive columns of the dense
//      row-major [2K][ncols] panel, written through L2; then (shared beta only) the frame's Gram partial on wave 0
#pragma once
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "priors_inl.h"

namespace bodyfit {
namespace {

// Jacobian panel store, written through L2 (sc0 sc1): the panel is this kernel's largest output (34 KB per frame) and
// nothing on the device re-reads it from this XCD's L2 before the next launch; left dirty it is flushed by the
// end-of-kernel release, which serialises ~9 MB of write-back behind the last wave.
__device__ __forceinline__ void store_through(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// One-launch sweep (kFused): the same.  (An earlier form of that kernel did better with plain stores: its Jacobian phases ran
// while other frame workgroups were still handing their operands over to consumers that polled the producers' lines, and
// 1.1 M eight-byte write-through stores delayed those signals by up to 7 us.  With two signals per frame on lines the consumers do
// not poll, the hand-off is over before the first panel row is written; same box, 256 frames: 22.0 against 22.6 us per step.
// BODYFIT_J_THROUGH=0 compiles the plain form for A/B runs.)
#ifndef BODYFIT_J_THROUGH
#define BODYFIT_J_THROUGH 1
#endif
// scope: 0 system (sc0 sc1, rounds 1-3), 1 agent (sc1: the form the mesh role's cloud stores use), 2 plain
template <bool kFused>
__device__ __forceinline__ void store_J(double* p, double v, int scope = 0) {
  if constexpr (kFused && !BODYFIT_J_THROUGH) *p = v;
  else if (scope == 1) store_f64_through(p, v);
  else if (scope == 2) *p = v;
  else store_through(p, v);
}


__device__ inline void mul33(const double* A, const double* B, double* C) {  // C = A B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ inline void mul33_bt(const double* A, const double* B, double* C) {  // C = A B^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C[r * 3 + c] = A[r * 3] * B[c * 3] + A[r * 3 + 1] * B[c * 3 + 1] + A[r * 3 + 2] * B[c * 3 + 2];
}
__device__ inline void mv3(const double* A, double x0, double x1, double x2, double* y) {
#pragma unroll
  for (int r = 0; r < 3; ++r) y[r] = A[r * 3] * x0 + A[r * 3 + 1] * x1 + A[r * 3 + 2] * x2;
}

// R(a) and dR/da_k for ONE k (0..2).  theta^2 <= DBL_EPSILON: R = I + [a]x, dR_k = [e_k]x (the first-order branch).
// One sincos of the half angle gives sin, cos and 1 - cos of theta (st = 2 s c, ct = 1 - 2 s^2, 1 - ct = 2 s^2 without
// cancellation).  Three lanes per joint each take one k, so the trigonometry (the long part) is the only serial piece.
__device__ void rodrigues_grad_k(double a0, double a1, double a2, int k, double* R, double* dRk) {
  const double th2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2), ith = 1.0 / th;
    double sh, ch;
    sincos(0.5 * th, &sh, &ch);
    const double st = 2.0 * sh * ch, omc = 2.0 * sh * sh, ct = 1.0 - omc;
    const double w[3] = {a0 * ith, a1 * ith, a2 * ith};
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = (r == c ? ct : 0.0) + st * K[r * 3 + c] + omc * w[r] * w[c];
    const double wk = (k == 0) ? w[0] : (k == 1 ? w[1] : w[2]);
    double dw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dw[i] = ((i == k ? 1.0 : 0.0) - w[i] * wk) * ith;
    const double dK[9] = {0, -dw[2], dw[1], dw[2], 0, -dw[0], -dw[1], dw[0], 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        dRk[r * 3 + c] = (r == c ? -st * wk : 0.0) + ct * wk * K[r * 3 + c] + st * dK[r * 3 + c] +
                         st * wk * w[r] * w[c] + omc * (dw[r] * w[c] + w[r] * dw[c]);
  } else {
    R[0] = 1; R[1] = -a2; R[2] = a1;
    R[3] = a2; R[4] = 1; R[5] = -a0;
    R[6] = -a1; R[7] = a0; R[8] = 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) dRk[i] = 0.0;
    if (k == 0) { dRk[5] = -1; dRk[7] = 1; }        // [e_x]x
    else if (k == 1) { dRk[2] = 1; dRk[6] = -1; }   // [e_y]x
    else { dRk[1] = -1; dRk[3] = 1; }               // [e_z]x
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

#ifdef BODYFIT_STAMPS
#define STAMP(i)                                                                              \
  do {                                                                                        \
    if (Pb.dbg && (threadIdx.x & 63) == 0) {                                                  \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      Pb.dbg[((size_t)f * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_;                  \
    }                                                                                         \
  } while (0)
#define STAMP_REAL(i)                                                                         \
  do {                                                                                        \
    if (Pb.dbg && (threadIdx.x & 63) == 0) {                                                  \
      unsigned long long t_;                                                                  \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      Pb.dbg[((size_t)f * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_;                  \
    }                                                                                         \
  } while (0)
#else
#define STAMP(i)
#define STAMP_REAL(i)
#endif

constexpr int KC = 32;  // keypoints staged per chunk
constexpr int kThreads = 512;   // 8 waves per frame = 2 per SIMD: the phases are latency-bound, the partner wave covers

// LDS carve (doubles)
constexpr int OFF_X = 0;                       // 88
constexpr int OFF_R = OFF_X + 88;              // 24*9
constexpr int OFF_DR = OFF_R + 216;            // 24*27
constexpr int OFF_A = OFF_DR + 648;            // 24*9
constexpr int OFF_P = OFF_A + 216;             // 24*3
constexpr int OFF_O = OFF_P + 72;              // 24*3
constexpr int OFF_JC = OFF_O + 72;             // 24*3
constexpr int OFF_W = OFF_JC + 72;             // 69*9 -> 624
constexpr int OFF_B = OFF_W + 624;             // 24*30
constexpr int OFF_FEAT = OFF_B + 720;          // 208
constexpr int OFF_CAM = OFF_FEAT + 208;        // Rr0[9], dRr0[27], pad -> 40
constexpr int OFF_KP = OFF_CAM + 40;           // KC*18
constexpr int OFF_TAB = OFF_KP + KC * 18;      // int tables (as 4-byte words): 128 ints = 64 doubles
constexpr int OFF_KPUV = OFF_TAB + 64;         // KC*2 observed pixels of the first keypoint chunk
constexpr int OFF_DS = OFF_KPUV + 2 * KC;      // 24*3*10  S_j - S_par(j)
constexpr int OFF_SC = OFF_DS + 720;           // 24*3*10  S_j - S_0
constexpr int OFF_PART = OFF_SC + 720;         // landmark rest vertices [<= 96], root keypoint offset [100..102]
constexpr int OFF_T = OFF_PART + 128;          // 24*3*10  T_j = B_j - A_j Sc_j (landmark shape columns)
constexpr int OFF_CHAIN = OFF_T + 720;         // 24 ancestor walk lists (64-bit words)
constexpr int OFF_LM = OFF_CHAIN + 24;         // landmarks: nL * LM_STRIDE, then their shapedirs rows nL * 3 * 10
constexpr int LM_VP = 0;                       // 3
constexpr int LM_Q = 3;                        // 3
constexpr int LM_A = 6;                        // 9  blended rotation
constexpr int LM_X = 15;                       // kMaxLmNnz*3 = 24
constexpr int LM_W = 39;                       // kMaxLmNnz weights
constexpr int LM_J = 47;                       // kMaxLmNnz joint ids (stored as doubles' worth of ints: 8 ints = 4 doubles)
constexpr int LM_NW = 51;                      // weight count (as double)
constexpr int LM_BETA = 52;                    // 10*3 = 30   d q / d beta
constexpr int LM_PD = 82;                      // 69*3 = 207  pose-blend Jacobian term
constexpr int LM_STRIDE = 290;
// int table layout inside OFF_TAB
constexpr int TAB_PARENT = 0;                  // 24
constexpr int TAB_ANC = 24;                    // 24
constexpr int TAB_KPID = 96;                   // KC

// What the one-launch sweep (k_sweep_roles) adds to the frame part: the frame's mesh operands are handed to the mesh
// workgroups inside the launch.
// What phase A of the frame role reads, as k_sweep_roles' leading scalar kernel arguments (preloaded into SGPRs with the wave:
// -amdgpu-kernarg-preload-count): the two table blocks (bodyfit_device.h kTab*, ptab_*_off), the per-launch pointers and the
// dimensions, so that the phase's loads do not wait for a scalar load of the by-value argument struct first.
struct FrameHead {
  const unsigned char* mtab;      // DevModel::tabA
  const unsigned char* ptab;      // DevProblem::ptab
  const double* R0;               // DevProblem::R0 (or the launch's override)
  int F, K;
  int dims;                       // nJ | nS << 6 | nL << 10 | ncols << 16 | use_shape << 24 | beta_stride << 25 | has mesh operands << 29
};
static_assert(kMaxJoints < 64 && kMaxShape < 16 && kMaxLandmarks < 64 && 7 + 3 * (kMaxJoints - 1) + kMaxShape < 256,
              "FrameHead::dims: 6 + 4 + 6 + 8 bits for nJ, nS, nL, ncols; beta_stride (0 or nS) in 4");
__host__ __device__ inline int frame_head_dims(int nJ, int nS, int nL, int ncols, int use_shape, int beta_stride, int has_coef) {
  return nJ | (nS << 6) | (nL << 10) | (ncols << 16) | ((use_shape ? 1 : 0) << 24) | (beta_stride << 25) | ((has_coef ? 1 : 0) << 29);
}

struct FusedFrame {
  unsigned* flag;                 // counter of 32-frame unit u at flag[u * kUnitCounterStride]: += 1 once a frame's mesh operands are published
  unsigned epoch;                 // this launch's number
  int j_scope;                    // Jacobian panel stores: 0 system-scope write-through, 1 agent-scope write-through, 2 plain (BODYFIT_J_SCOPE)
};

// Barrier between two phases.  The phases only exchange LDS data, so the fused kernel waits for the LDS counter alone:
// its tile operands are in flight (LDS-DMA, counted on vmcnt) and the Jacobian stores are outstanding, and a
// __syncthreads() would drain both at every phase (hipcc emits s_waitcnt vmcnt(0) in front of the barrier whenever an
// LDS-DMA may be pending).
template <bool kFused>
__device__ __forceinline__ void frame_sync() {
  if constexpr (kFused) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// 16-byte store of a mesh operand.  Fused: write-through (sc1), the payload form of the in-launch hand-off.
template <bool kFused>
__device__ __forceinline__ void store_operand16(void* p, uint4 v) {
  if constexpr (kFused) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
    const u32x4_ x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}

template <bool kFused>
__device__ __forceinline__ void frame_part(const DevModel& M, const DevProblem& Pb, const double* __restrict__ params,
                                           const double* __restrict__ beta, double* __restrict__ r_out,
                                           double* __restrict__ J_out, double* __restrict__ joints_out, const MeshCoef& mc,
                                           int want_jac, double* sm, int f, const FusedFrame& fu, const FrameHead& hd) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // dimensions and the pointers of phase A: in the one-launch sweep from the preloaded head (nothing of M / Pb / mc is touched
  // before the phase's loads are out: their scalar loads are then still in flight), otherwise from the structs
  const int nJ = kFused ? (hd.dims & 63) : M.nJ, nS = kFused ? ((hd.dims >> 6) & 15) : M.nS, nL = kFused ? ((hd.dims >> 10) & 63) : M.nL;
  const int ncols = kFused ? ((hd.dims >> 16) & 255) : Pb.ncols;
  const int npose = 7 + 3 * (nJ - 1);
  const bool use_shape = kFused ? (((hd.dims >> 24) & 1) != 0) : (Pb.use_shape != 0);
  const int beta_stride = kFused ? ((hd.dims >> 25) & 15) : Pb.beta_stride;
  const unsigned char* const mtab = kFused ? hd.mtab : M.tabA;
  const unsigned char* const ptab = kFused ? hd.ptab : Pb.ptab;
  const int* const t_parent = reinterpret_cast<const int*>(mtab + kTabParent);
  const unsigned* const t_anc = reinterpret_cast<const unsigned*>(mtab + kTabAnc);
  const unsigned long long* const t_chain = reinterpret_cast<const unsigned long long*>(mtab + kTabChain);
  const int* const t_lm_woff = reinterpret_cast<const int*>(mtab + kTabLmWoff);
  const int* const t_lm_wj = reinterpret_cast<const int*>(mtab + kTabLmWj);
  const double* const t_lm_ww = reinterpret_cast<const double*>(mtab + kTabLmWw);
  const double* const t_lm_vt = reinterpret_cast<const double*>(mtab + kTabLmVt);
  const double* const t_lm_sd = reinterpret_cast<const double*>(mtab + kTabLmSd);
  const double* const t_dS = reinterpret_cast<const double*>(mtab + kTabDS);
  const double* const t_Sc = reinterpret_cast<const double*>(mtab + kTabSc);
  const double* const t_offset = reinterpret_cast<const double*>(mtab + kTabOffset);
  const double* const t_Jc0 = reinterpret_cast<const double*>(mtab + kTabJc0);
  const int t_F = kFused ? hd.F : Pb.F, t_K = kFused ? hd.K : Pb.K;
  const int* const t_kp_offset = reinterpret_cast<const int*>(ptab);
  const int* const t_kp_id = reinterpret_cast<const int*>(ptab + ptab_id_off(t_F));
  const double* const t_kp_uv = reinterpret_cast<const double*>(ptab + ptab_uv_off(t_F, t_K));
  const double* const t_R0 = kFused ? hd.R0 : Pb.R0;
  const int P = M.P;
  double* sx = sm + OFF_X;
  double* sR = sm + OFF_R;
  double* sdR = sm + OFF_DR;
  double* sA = sm + OFF_A;
  double* sP = sm + OFF_P;
  double* sO = sm + OFF_O;
  double* sJc = sm + OFF_JC;
  double* sW = sm + OFF_W;
  double* sB = sm + OFF_B;
  double* sFeat = sm + OFF_FEAT;
  double* sCam = sm + OFF_CAM;
  double* sKp = sm + OFF_KP;
  int* sTab = reinterpret_cast<int*>(sm + OFF_TAB);
  double* sKpUv = sm + OFF_KPUV;
  double* sDS = sm + OFF_DS;
  double* sSc = sm + OFF_SC;
  double* sPart = sm + OFF_PART;
  double* sLm = sm + OFF_LM;
  double* sT = sm + OFF_T;
  double* sLmSd = sLm + nL * LM_STRIDE;          // [nL][3][nS] shapedirs - S_root of the landmark vertices
  int* sParent = sTab + TAB_PARENT;
  unsigned* sAnc = reinterpret_cast<unsigned*>(sTab + TAB_ANC);
  int* sKpId = sTab + TAB_KPID;
  unsigned* sKpAnc = reinterpret_cast<unsigned*>(sTab + 48);   // [KC] ancestor mask of each staged keypoint's joint
  unsigned long long* sChain = reinterpret_cast<unsigned long long*>(sm + OFF_CHAIN);   // [24] ancestor walk lists

  STAMP_REAL(10);
  STAMP(0);
#ifdef BODYFIT_STAMPS
  if (Pb.dbg && tid == 0) {   // which XCD this workgroup runs on (HW_REG_XCC_ID = 20, low 4 bits)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Pb.dbg[((size_t)f * 8) * 16 + 9] = xcc & 0xf;
  }
#endif
  // frame already converged (device LM).  Not in the one-launch sweep, which never carries frame flags: the test would put a
  // scalar round trip of its own (the flag pointer, ~0.2 us) in front of the role's operand loads
  if constexpr (!kFused) {
    if (Pb.frame_flags && !(Pb.frame_flags[f] & Pb.frame_mask)) return;
  }
  constexpr int kLoaders = kThreads;
  constexpr bool loader = true;
  // ---- A. small model tables, landmark weights and the frame's parameters into LDS: every load of the phase is issued
  //      before the first of them is used (unconditional loads from clamped indices; the stores, before the barrier, are
  //      predicated).  Written as load-store loops the phase compiled to one dependent L2 round trip per loop trip. --------
  constexpr int kSdPasses = (kMaxLandmarks * 3 * kMaxShape + kThreads - 1) / kThreads;   // 2
  constexpr int kPasses = (720 + kLoaders - 1) / kLoaders;                                 // 24 x 3 x 10 doubles: 2
  const bool coef_wave = kFused && wave == 5 && ((hd.dims >> 29) & 1) != 0;
  const int tj_c = min(tid, nJ - 1);
  const int par_in = t_parent[tj_c];
  const unsigned anc_in = t_anc[tj_c];
  const unsigned long long chain_in = t_chain[tj_c];
  const int nlw = nL * kMaxLmNnz, lw_i = min(tid, max(nlw, 1) - 1);          // <= 256 items: one pass
  const double lww_in = t_lm_ww[lw_i];
  const int lwj_in = t_lm_wj[lw_i];
  const int lwo_in = t_lm_woff[lw_i / kMaxLmNnz];
  const int nsd = nL * 3 * nS, nds = nJ * 3 * nS;
  double sd_in[kSdPasses], t0[kPasses], t1[kPasses];
#pragma unroll
  for (int u = 0; u < kSdPasses; ++u) sd_in[u] = t_lm_sd[min(tid + u * kThreads, max(nsd, 1) - 1)];
#pragma unroll
  for (int u = 0; u < kPasses; ++u) {
    const int i = min(tid + u * kLoaders, max(nds, 1) - 1);
    t0[u] = t_dS[i];
    t1[u] = t_Sc[i];
  }
  // the first keypoint chunk depends on kp_offset[f] (a second round trip): its loads are issued here but land in LDS
  // only after the barrier, so phase A waits for one round trip, not two (the staged keypoints are first read in F)
  const int k_begin0 = t_kp_offset[f], nk0 = min(KC, t_kp_offset[f + 1] - k_begin0);
  // model constants phase B needs (chain offsets, centred rest joints, landmark template rows): requested here, with
  // the tables, and consumed from registers after the barrier, so phase B has no round trip of its own
  const int o_i = tid - 128, v_row = tid - 256;
  const bool o_lane = tid >= 128 && o_i < nJ * 3, v_lane = tid >= 256 && v_row < nL * 3;
  double o_pre = 0.0, jc_pre = 0.0, vt_pre = 0.0;
  if (o_lane) { o_pre = t_offset[o_i]; jc_pre = t_Jc0[o_i]; }
  if (v_lane) vt_pre = t_lm_vt[v_row];
  // R0 of this frame for the camera matrices Rr0 = R_root R0, dRr0_c = dR_root,c R0 (36 entries, wave 6 in phase C):
  // lane's column c of R0, requested here
  const int cam_lane = tid - 384;
  const bool cam_on = cam_lane >= 0 && cam_lane < 36;
  double r0c0 = 0.0, r0c1 = 0.0, r0c2 = 0.0;
  if (cam_on) {
    const double* R0 = t_R0 + (size_t)f * 9;
    const int c = cam_lane % 3;
    r0c0 = R0[c]; r0c1 = R0[3 + c]; r0c2 = R0[6 + c];
  }
  double x_in = 0.0;
  const bool x_lane = tid >= 128 && tid - 128 < npose, b_lane = tid >= 224 && tid - 224 < nS;
  if (x_lane) x_in = params[(size_t)f * npose + tid - 128];
  if (b_lane) x_in = (use_shape && beta) ? beta[(size_t)f * beta_stride + tid - 224] : 0.0;
  // issued last and by every lane (the arrays are padded by one chunk): loads return in order, so everything above is
  // waited for with these still in flight
  int kp_id0 = 0;
  double kp_u0 = 0.0, kp_v0 = 0.0;
  if (loader) {
    kp_id0 = t_kp_id[k_begin0 + (tid & (KC - 1))];
    kp_u0 = t_kp_uv[2 * (size_t)(k_begin0 + (tid & (KC - 1)))];
    kp_v0 = t_kp_uv[2 * (size_t)(k_begin0 + (tid & (KC - 1))) + 1];
  }
  if (tid < nJ) { sParent[tid] = par_in; sAnc[tid] = anc_in; sChain[tid] = chain_in; }
  volatile int* sWalkDone = reinterpret_cast<volatile int*>(sPart + 104);   // phase C: wave 6's chain quantities are in LDS
  if (tid == 0) *sWalkDone = 0;
  if (tid < nlw) {   // landmark skinning weights, fixed stride (padded with weight 0)
    const int l = tid / kMaxLmNnz, i = tid % kMaxLmNnz;
    double* L = sLm + l * LM_STRIDE;
    L[LM_W + i] = lww_in;
    reinterpret_cast<int*>(L + LM_J)[i] = lwj_in;
    if (i == 0) L[LM_NW] = (double)lwo_in;   // weight count
  }
#pragma unroll
  for (int u = 0; u < kSdPasses; ++u)
    if (tid + u * kThreads < nsd) sLmSd[tid + u * kThreads] = sd_in[u];
#pragma unroll
  for (int u = 0; u < kPasses; ++u) {
    const int i = tid + u * kLoaders;
    if (i < nds) { sDS[i] = t0[u]; sSc[i] = t1[u]; }
  }
  if (x_lane) sx[tid - 128] = x_in;
  if (b_lane) sx[npose + tid - 224] = x_in;
  frame_sync<kFused>();
  if (tid < nk0) {
    sTab[TAB_KPID + tid] = kp_id0;
    sKpUv[2 * tid] = kp_u0;
    sKpUv[2 * tid + 1] = kp_v0;
  }
  const double* sbeta = sx + npose;

  STAMP(1);
  // ---- B. wave 0: Rodrigues + gradient per joint (joint 0 = root angle-axis);
  //         waves 1-3: chain offsets o_j(beta) (include/Sim3BA.h:142-170,179-205), centred rest joints ------
  if (tid < 3 * nJ) {     // (joint, k): wave 0 and a few lanes of wave 1
    const int jj = tid / 3, k = tid - 3 * jj;
    const double* aa = (jj == 0) ? (sx + 1) : (sx + 7 + 3 * (jj - 1));
    double R[9], dRk[9];
    rodrigues_grad_k(aa[0], aa[1], aa[2], k, R, dRk);
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sR[jj * 9 + i] = R[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) sdR[jj * 27 + k * 9 + i] = dRk[i];
  }
  if (v_lane) {
    // landmark rest vertex before the pose blend: v_t + shapedirs . beta (10 independent loads)
    const int row = v_row;
    double sv[kMaxShape];
#pragma unroll
    for (int k = 0; k < kMaxShape; ++k) sv[k] = (use_shape && k < nS) ? sLmSd[row * nS + k] : 0.0;
    double acc = vt_pre;
#pragma unroll
    for (int k = 0; k < kMaxShape; ++k) acc += sv[k] * sbeta[min(k, nS > 0 ? nS - 1 : 0)];
    sPart[row] = acc;
  }
  if (coef_wave) {
    // One-launch sweep, wave 5 (nothing else to do in this phase): the rotations behind the mesh role's blend coefficients (MFMA A
    // fragments: [vec(R_j - I) | beta | 1 1 0 ..] of this frame in bf16 hi + lo), in f32, straight from the raw parameters —
    // not from wave 0's f64 rotations, which exist only at the end of this phase: round 3 packed those in phase C and
    // published them 5.2 us into the launch.  (The mesh is an f32 product split in bf16: the f64 rotation rounded to f32 and
    // the f32 rotation differ by ~1e-7 relative on coefficients that multiply centimetre-scale directions; mesh tolerance
    // 5e-6 m.)  23 lanes one joint each, the lanes from 32 on beta, the template's two slots (coefficient 1.0) and K's padding
    // -> 224 floats of scratch (sW .. sB are free until phase C / D).  Packing, storing and signalling: top of phase C (all of
    // it here made this wave the phase's longest by 1.9 k cycles; the rotations in phase A, under the table loads, from a copy
    // of the parameters wave 5 requests first, published at 3.3 us but stretched phases A and B by 0.9 k cycles each: 23.8
    // against 21.5 us per step).
    float* cf = reinterpret_cast<float*>(sW);
    const int nfeat = 9 * (nJ - 1);
    if (lane < nJ - 1) {
      const float a0 = (float)sx[7 + 3 * lane], a1 = (float)sx[8 + 3 * lane], a2 = (float)sx[9 + 3 * lane];
      const float th2 = a0 * a0 + a1 * a1 + a2 * a2;
      float st = 1.0f, omc = 0.0f, w0 = a0, w1 = a1, w2 = a2;      // first-order branch: R - I = [a]x
      if (th2 > 1e-20f) {
        const float ith = rsqrtf(th2), th = th2 * ith;
        float sh, ch;
        sincosf(0.5f * th, &sh, &ch);
        st = 2.0f * sh * ch; omc = 2.0f * sh * sh;                  // 1 - cos without cancellation
        w0 = a0 * ith; w1 = a1 * ith; w2 = a2 * ith;
      }
      const float on = Pb.pose_blend ? 1.0f : 0.0f;
      float* o = cf + 9 * lane;                                     // R - I = sin [w]x + (1 - cos) (w w^T - I)
      o[0] = on * (omc * (w0 * w0 - 1.0f)); o[1] = on * (omc * w0 * w1 - st * w2); o[2] = on * (omc * w0 * w2 + st * w1);
      o[3] = on * (omc * w1 * w0 + st * w2); o[4] = on * (omc * (w1 * w1 - 1.0f)); o[5] = on * (omc * w1 * w2 - st * w0);
      o[6] = on * (omc * w2 * w0 - st * w1); o[7] = on * (omc * w2 * w1 + st * w0); o[8] = on * (omc * (w2 * w2 - 1.0f));
    } else if (lane >= 32 && lane - 32 < 16 * kBlendKSteps - kPoseFeat) {
      const int i = lane - 32;                                        // slot 207 + i: beta_i, the template's two slots, padding
      cf[kPoseFeat + i] = (i < nS) ? (float)sbeta[i] : ((i == kMaxShape || i == kMaxShape + 1) ? 1.0f : 0.0f);
    }
    if (lane < 32)
      for (int k = nfeat + lane; k < kPoseFeat; k += 32) cf[k] = 0.0f;   // (models with fewer joints: the unused pose slots)
  }
  if (o_lane) {   // 3 nJ <= 72 < 128: one item per lane
    const int i = o_i;
    double o = o_pre, jc = jc_pre;
    if (use_shape) {
      for (int k = 0; k < nS; ++k) {
        o += sDS[i * nS + k] * sbeta[k];
        jc += sSc[i * nS + k] * sbeta[k];
      }
    }
    sO[i] = (i < 3) ? 0.0 : o;
    if (i < 3) sPart[100 + i] = o;     // the root keypoint's own q = offset_0 + S_0 beta (include/Sim3BA.h:142-170)
    sJc[i] = jc;
  }
  // Everything phase A requested has been consumed by now (or, R0 on wave 6, landed microseconds ago), but each of those
  // loads and its use sit under the same lane predicate in DIFFERENT blocks, so hipcc's wait-count pass still carries their
  // destination registers as "maybe pending" and puts s_waitcnt vmcnt(0) in front of every later reuse of such a register
  // — in phase C that is right behind the 27 landmark loads (wave 0: in front of the root entries the hand-off waits for).
  // An explicit vmcnt(0) the pass can see (the builtin, not asm text) costs nothing here and clears that state.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  frame_sync<kFused>();

  STAMP(2);
  // ---- C. everything that only needs R_j and o_j, with no ordering between items:
  //   pose feature vec(R_j - I); chain quantities as 3-vector walks up the kinematic chain
  //     A_j[:,c] : v = R_j[:,c];   v <- R_k v            (k = ancestors of j below the root)
  //     P_j      : v = o_j;        v <- R_k v + o_k        (the reference's own walk, Sim3BA.h:173-207)
  //     B_j[:,b] : v = dS_j[:,b];  v <- R_k v + dS_k[:,b]  (d P_j / d beta_b)
  //   and the landmark blend rows (wave reductions over the 207 pose-blend columns) --------------------------
  // Landmark pose-blend terms, item = (landmark l, joint k) on lane k - 1 of half-wave l (two landmarks per wave):
  // the 27 posedirs values pd[l][a][9 (k - 1) + e] are loaded ONCE (coalesced, joint-minor table) and serve both the
  // blend row  v_p[l][a] = sum_k pd . vec(R_k - I)  (reduced over the half-wave's lanes, no LDS partials) and the
  // Jacobian term  h[c][a] = pd . vec(dR_{k,c})  (parked in the landmark's LM_PD slot until phase E applies the
  // blended rotation).  The loads are issued before the chain walks and land while those run.
  const int lm_k = lane & 31, lm_l = 2 * wave + (lane >> 5);
  const bool lm_blend = Pb.pose_blend && P > 0;
  auto lm_load = [&](int l, double (&pdv)[27]) {
    // UNCONDITIONAL loads from a clamped (always valid) row, masked by a factor: written as `on ? load : 0` hipcc branches
    // around every pair of loads and waits vmcnt(0) inside each branch, i.e. 14 dependent L2 round trips (seen in the .s)
    // The mask is NOT applied here: a product right behind the loads makes hipcc wait for all 27 of them on the spot
    // (s_waitcnt vmcnt(0) in the .s), i.e. the wave sits out an L2 round trip before the work that was meant to run under it
    // (on wave 0: the blend-coefficient stores and the root entries wave 7's hand-off waits for).  lm_terms masks the factors.
    const double* row = M.lm_pd + (size_t)min(l, max(nL, 1) - 1) * 27 * 32 + lm_k;
#pragma unroll
    for (int ae = 0; ae < 27; ++ae) pdv[ae] = row[ae * 32];
  };
  auto lm_terms = [&](int l, const double (&pdv)[27]) {
    const int k = min(lm_k + 1, nJ - 1);
    const bool on = l < nL && lm_k < nJ - 1;
    const double onf = (lm_blend && on) ? 1.0 : 0.0;      // (masked lanes loaded a clamped, valid row)
    double part[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const double fe = (sR[k * 9 + e] - ((e % 4 == 0) ? 1.0 : 0.0)) * onf;
#pragma unroll
      for (int a = 0; a < 3; ++a) part[a] += pdv[a * 9 + e] * fe;
    }
    if (want_jac) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double* d = sdR + k * 27 + c * 9;
        double h[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          const double de = d[e] * onf;
#pragma unroll
          for (int a = 0; a < 3; ++a) h[a] += pdv[a * 9 + e] * de;
        }
        if (on) {
          double* o = sLm + l * LM_STRIDE + LM_PD + (3 * (k - 1) + c) * 3;
          o[0] = h[0]; o[1] = h[1]; o[2] = h[2];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) part[a] += __shfl_xor(part[a], off, 32);
    }
    if (on && lm_k == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) sLm[l * LM_STRIDE + LM_VP + a] = sPart[l * 3 + a] + part[a];
    }
  };
  // (waves whose two landmark slots are both beyond nL skip the loads and the 108 products per lane altogether: with 11
  //  landmarks that is waves 6-7, which carry the chain walks, this phase's longest items)
  const bool lm_wave0 = 2 * wave < nL, lm_wave1 = 2 * wave + 16 < nL;
  // root entries A_0 = I, P_0 = 0, B_0 = 0 (wave 0), first thing: wave 7's mesh operands wait for them (counter below)
  if (tid < 9) sA[tid] = (tid % 4 == 0) ? 1.0 : 0.0;
  if (tid >= 16 && tid < 19) sP[tid - 16] = 0.0;
  if (tid >= 32 && tid - 32 < 3 * nS) { sB[tid - 32] = 0.0; sT[tid - 32] = 0.0; }
  if (wave == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(sWalkDone), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (coef_wave) {
    // the 56 fragments of the blend coefficients out of the scratch wave 5 filled in phase B, write-through; drain; ONE
    // agent-scope add to the coefficient counter of the frame's 32-frame unit (cdna guide, Guideline 16 R1) — all in front of
    // this wave's landmark loads (vmcnt would cover them too)
    const float* cf = reinterpret_cast<const float*>(sW);
    if (lane < kBlendKSteps * 4) {
      const int kstep = lane >> 2, h = (lane >> 1) & 1, hl = lane & 1;
      const float4 v0 = *reinterpret_cast<const float4*>(cf + 16 * kstep + 8 * h);
      const float4 v1 = *reinterpret_cast<const float4*>(cf + 16 * kstep + 8 * h + 4);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      uint32_t pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint16_t b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint16_t hi = f32_to_bf16(x[2 * q + t]);
          b[t] = hl == 0 ? hi : f32_to_bf16(x[2 * q + t] - bf16_to_f32(hi));
        }
        pk[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
      }
      // MFMA row of this frame inside its 32-frame tile: accumulator register i of half-wave h holds frame 2 i + h, so that one
      // register row covers two CONSECUTIVE frames (2,304 contiguous bytes of transforms)
      const int ftile = f / kFTile, phi = f % kFTile;
      const int row = 8 * (phi >> 3) + 4 * (phi & 1) + ((phi >> 1) & 3);
      uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(mc.featA) + ((size_t)ftile * kBlendKSteps + kstep) * 2048 +
                                            feat_frag_off(h * 32 + row, hl));
      store_operand16<true>(dst, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_REAL(14);
    if (lane == 0)
      (void)__hip_atomic_fetch_add(fu.flag + (size_t)(f / kFTile) * kUnitCounterStride + kUnitCoefOffset, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
  }
  double pdv0[27];
  if (lm_wave0) lm_load(lm_l, pdv0);
  for (int i = tid; i < 208; i += kThreads) {
    double v = 0.0;
    if (i < 9 * (nJ - 1)) v = sR[9 + i] - (((i % 9) % 4 == 0) ? 1.0 : 0.0);
    sFeat[i] = v;
  }
  // chain walk of one item up the kinematic chain (kind 0: column sel of A_j, 1: P_j, 2: column sel of B_j).
  // The ancestors come from the joint's packed walk list (ONE LDS read), not from parent[parent[...]] (a dependent LDS round
  // trip per level), and the next level's rotation and offset are requested before this level's products: a level costs
  // its nine dependent f64 products, not three LDS round trips (walks: 6.4 k -> ~2.5 k cycles on the hand-off's critical path).
  // `kind` may differ from lane to lane (phase C walks A columns and P vectors in the same pass): the loop body is the same
  // for every kind — v <- R_k v + m T[(3 k + a) stride + sel] with per-lane (table, stride, sel, m) — so lanes of different
  // kinds walk TOGETHER; written as one branch per kind, the kinds of a wave ran one after the other (twice the chain depth).
  auto walk = [&](int kind, int j, int sel, bool on) {
    j = on ? j : 1;
    const double* tab = (kind == 2) ? sDS : sO;            // the per-level addend: o_k (kind 1), dS_k[:, sel] (kind 2), none (0)
    const int stride = (kind == 2) ? nS : 1, off = (kind == 2) ? sel : 0;
    const double m = (kind == 0) ? 0.0 : 1.0;
    const double* ini = (kind == 0) ? (sR + j * 9 + sel) : (tab + (j * 3) * stride + off);
    const int istride = (kind == 0) ? 3 : stride;
    double v0 = ini[0], v1 = ini[istride], v2 = ini[2 * istride];
    unsigned long long ch = on ? sChain[j] : 0ull;
    int k = (int)(ch & 31ull);
    double Rc[9], oc[3];
    auto fetch = [&](int kk, double (&Rn)[9], double (&on_)[3]) {   // (kk = 0 past the end: the root's entries, not used)
#pragma unroll
      for (int e = 0; e < 9; ++e) Rn[e] = sR[kk * 9 + e];
#pragma unroll
      for (int a3 = 0; a3 < 3; ++a3) on_[a3] = tab[(kk * 3 + a3) * stride + off];
    };
    fetch(k, Rc, oc);
    while (k > 0) {
      ch >>= 5;
      const int kn = (int)(ch & 31ull);
      double Rn[9], on_[3];
      fetch(kn, Rn, on_);
      const double t0 = Rc[0] * v0 + Rc[1] * v1 + Rc[2] * v2 + m * oc[0];
      const double t1 = Rc[3] * v0 + Rc[4] * v1 + Rc[5] * v2 + m * oc[1];
      const double t2 = Rc[6] * v0 + Rc[7] * v1 + Rc[8] * v2 + m * oc[2];
      v0 = t0; v1 = t1; v2 = t2;
#pragma unroll
      for (int e = 0; e < 9; ++e) Rc[e] = Rn[e];
      oc[0] = on_[0]; oc[1] = on_[1]; oc[2] = on_[2];
      k = kn;
    }
    if (!on) return;
    if (kind == 0) { sA[j * 9 + sel] = v0; sA[j * 9 + 3 + sel] = v1; sA[j * 9 + 6 + sel] = v2; }
    else if (kind == 1) { sP[j * 3] = v0; sP[j * 3 + 1] = v1; sP[j * 3 + 2] = v2; }
    else {
      sB[(j * 3 + 0) * nS + sel] = v0; sB[(j * 3 + 1) * nS + sel] = v1; sB[(j * 3 + 2) * nS + sel] = v2;
      // T_j[:, sel] = B_j[:, sel] - A_j Sc_j[:, sel]: what a skinning weight on joint j adds to d q_l / d beta_sel
      double c[3];
      mv3(sA + j * 9, sSc[(j * 3 + 0) * nS + sel], sSc[(j * 3 + 1) * nS + sel], sSc[(j * 3 + 2) * nS + sel], c);
      sT[(j * 3 + 0) * nS + sel] = v0 - c[0]; sT[(j * 3 + 1) * nS + sel] = v1 - c[1]; sT[(j * 3 + 2) * nS + sel] = v2 - c[2];
    }
  };
  // blend-coefficient fragments of the mesh kernel: they need R_j (phase B) and beta only, so wave 0 stores them first
  // thing, in the shadow of its landmark posedirs loads, and they have long left when it says so (below).  (Measured and
  // rejected: on wave 6, the phase's lightest wave, in front of its chain walk — wave 0 is this phase's longest wave by 1.8 k
  // cycles — : wave 7 then waits for wave 6's later walk, the transforms go out 2 us later and the phase is no shorter.)
  if (!kFused && wave == 0 && mc.featA) {
    // MFMA row of this frame inside its 32-frame tile.  feat_perm (k_sweep_roles' mesh role): accumulator register i of the
    // half-wave h holds frame 2 i + h, so that one register row covers two CONSECUTIVE frames (2,304 contiguous bytes of
    // transforms); otherwise (k_mesh_blend_lbs) the natural order, register i <-> frames 8 (i >> 2) + (i & 3) + 4 h.
    const int ftile = f / kFTile, phi = f % kFTile;
    const int row = Pb.feat_perm ? (8 * (phi >> 3) + 4 * (phi & 1) + ((phi >> 1) & 3)) : phi;
    if (lane >= 8 && lane - 8 < kBlendKSteps * 4) {
      const int wl = lane - 8;
      const int kstep = wl >> 2, h = (wl >> 1) & 1, hl = wl & 1;
      // eight consecutive blend coefficients k0 .. k0 + 7: pose feature vec(R_j - I) (k < 207), then beta, then the
      // template's two slots.  Branch-free: all sixteen LDS reads are issued together from clamped indices and the value is
      // selected (written as if / else-if the eight values cost eight dependent LDS round trips)
      const int k0 = kstep * 16 + 8 * h;
      const int nfeat = 9 * (nJ - 1);
      double vr[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        vr[u] = sR[9 + min(k0 + u, nfeat - 1)];
        vb[u] = sbeta[min(max(k0 + u - kPoseFeat, 0), max(nS, 1) - 1)];
      }
      uint32_t pk[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        uint16_t b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = k0 + 2 * jj + u;
          const float xr = (Pb.pose_blend && k < nfeat) ? (float)(vr[2 * jj + u] - (((k % 9) % 4 == 0) ? 1.0 : 0.0)) : 0.0f;
          const float xb = (k - kPoseFeat < nS) ? (float)vb[2 * jj + u]
                                                : ((k == kPoseFeat + kMaxShape || k == kPoseFeat + kMaxShape + 1) ? 1.0f : 0.0f);
          const float x = (k < kPoseFeat) ? xr : xb;
          const uint16_t hi = f32_to_bf16(x);
          b[u] = hl == 0 ? hi : f32_to_bf16(x - bf16_to_f32(hi));
        }
        pk[jj] = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
      }
      uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(mc.featA) + ((size_t)ftile * kBlendKSteps + kstep) * 2048 +
                                            feat_frag_off(h * 32 + row, hl));
      store_operand16<kFused>(dst, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    }
  }
  if (cam_on) {
    // Rr0 = R_root R0, dRr0_c = dR_root,c R0 (include/Sim3BA.h:210-216 and its derivative): before the chain walks, so that
    // the mesh operands can be published right behind this phase's barrier
    const int mtx = cam_lane / 9, e = cam_lane % 9, r = e / 3;
    const double* Lm = (mtx == 0) ? sR : (sdR + (mtx - 1) * 9);
    sCam[cam_lane] = Lm[r * 3] * r0c0 + Lm[r * 3 + 1] * r0c1 + Lm[r * 3 + 2] * r0c2;
  }
  {
    // A_j columns and P_j (92 items) on waves 6-7, beside the landmark items of waves 0-5 (two landmarks per wave); each of
    // the two waves takes half of the A columns and half of the (dearer) P walks.  The 230 B_j columns nothing needs
    // before phase E are walked in phase D, where most threads are idle
    const int nA = 3 * (nJ - 1), nP = nJ - 1;
    const int nA6 = (nA + 1) / 2, nP6 = (nP + 1) / 2;
    if (wave == 6 || wave == 7) {   // ONE walk call per wave: its A lanes and its P lanes go up their chains together
      const int a0 = wave == 6 ? 0 : nA6, na = wave == 6 ? nA6 : nA - nA6;
      const int p0 = wave == 6 ? 0 : nP6, np = wave == 6 ? nP6 : nP - nP6;
      const bool isA = lane < na, isP = !isA && lane - na < np;
      const int item = isA ? a0 + lane : p0 + (lane - na);
      walk(isA ? 0 : 1, isA ? 1 + item / 3 : 1 + item, isA ? item % 3 : 0, isA || isP);
    }
  }
  // ---- mesh operands (blend-coefficient fragments, skinning transforms) and posed joints: wave 7, still in phase C ----
  // They need R_j (phase B), the chain quantities A_j, P_j (waves 6 and 7, just above), the root entries (waves 0, above)
  // and Rr0 (wave 6, above).  Wave 6 and wave 0 say so through one LDS word each wave-program-ordered behind its writes;
  // everything is stored by this ONE wave, so the in-launch hand-off needs no workgroup barrier (cdna guide, Guideline 16 R1:
  // one lane signals for all the stores of its workgroup after every STORING wave's vmcnt(0)); the drain and the signal
  // are wave 7's only work in phase D.
  // Who tells wave 7 what, through one LDS counter (every add program-ordered behind the adder's LDS writes; LDS serves a wave
  // in order, and the lgkmcnt wait keeps hipcc from sinking the writes):
  //   wave 0: the root entries are in LDS (first thing in this phase)       (+1)
  //   wave 6: its chain quantities and Rr0 are in LDS                      (+256)  -> count 257: wave 7 builds the transforms
  // The hand-off itself is TWO signals per frame (one-launch sweep), each by the wave that stored the payload, after its own
  // vmcnt(0): wave 0 for the blend coefficients (they need R_j only: out at ~4.8 us, the slowest of 256 frames at 5.8 us),
  // wave 7 for the transforms (behind the chain walks: 5.9 us median, 7.7-8.3 us the slowest).  A mesh wave starts its blend on
  // the first and needs the second only in front of its skinning rows, 3 us later.
  // (cdna guide, Guideline 16: "each wave adds to a counter in LDS after its wait and the wave whose add is last signals")
  if (wave == 6) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(sWalkDone), 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // (one-launch sweep: the blend coefficients left at the top of this phase, by wave 5)
  if (wave == 7) {
    STAMP(13);
    for (int spin = 0; spin < (1 << 20) && *sWalkDone < 257; ++spin) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    STAMP(14);
    const double s_ = sx[0];
    const double* Rr0_ = sCam;
    // item = (joint jj, row r) of the 3 x 4 transform  s Rr0 [A_jj | P_jj - A_jj Jc_jj] + [0 | t]: item it of frame f is the
    // 16-byte chunk 3 nJ f + it of skinT (and double 3 nJ f + it of joints_out), so a wave's stores are CONTIGUOUS (nine full
    // 128-byte lines per frame; one lane per joint wrote 16-byte pieces 48 bytes apart, three partial writes to every line:
    // the write-through acknowledgements the hand-off waits for had a tail of 1-2 us)
    for (int it = lane; it < 3 * nJ; it += 64) {
      const int jj = it / 3, r = it - 3 * jj;
      const double* Aj = sA + jj * 9;
      const double a0 = Rr0_[r * 3], a1 = Rr0_[r * 3 + 1], a2 = Rr0_[r * 3 + 2];
      double q[3];
      mv3(Aj, sJc[jj * 3], sJc[jj * 3 + 1], sJc[jj * 3 + 2], q);
      const double p0 = sP[jj * 3], p1 = sP[jj * 3 + 1], p2 = sP[jj * 3 + 2];
      if (mc.skinT) {
        const double ra0 = a0 * Aj[0] + a1 * Aj[3] + a2 * Aj[6];      // row r of Rr0 A_jj (mul33's order of products)
        const double ra1 = a0 * Aj[1] + a1 * Aj[4] + a2 * Aj[7];
        const double ra2 = a0 * Aj[2] + a1 * Aj[5] + a2 * Aj[8];
        const double tr = a0 * (p0 - q[0]) + a1 * (p1 - q[1]) + a2 * (p2 - q[2]);
        float4* T = reinterpret_cast<float4*>(mc.skinT + (size_t)f * nJ * 12) + it;
        store_operand16<kFused>(T, make_uint4(__float_as_uint((float)(s_ * ra0)), __float_as_uint((float)(s_ * ra1)),
                                              __float_as_uint((float)(s_ * ra2)), __float_as_uint((float)(s_ * tr + sx[4 + r]))));
      }
      if (joints_out) joints_out[(size_t)f * nJ * 3 + it] = s_ * (a0 * p0 + a1 * p1 + a2 * p2) + sx[4 + r];
    }
    STAMP(15);
    if constexpr (kFused) {
      // hand-off of the transforms, still inside phase C (wave 7 has ~1 k cycles of slack before wave 0 reaches the phase's
      // barrier): this wave's stores have left, then one agent-scope add to the counter of the frame's 32-frame unit
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP_REAL(9);
      if (lane == 0)
        (void)__hip_atomic_fetch_add(fu.flag + (size_t)(f / kFTile) * kUnitCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      STAMP_REAL(12);
    }
  }
  if (lm_wave0) lm_terms(lm_l, pdv0);
  if (lm_wave1) {   // landmarks 16..31: second pass
    double pdv1[27];
    lm_load(lm_l + 16, pdv1);
    lm_terms(lm_l + 16, pdv1);
  }
  STAMP(3);
  frame_sync<kFused>();

  STAMP(4);
  // ---- D. wave 0-1: W_{k,c} = A_p (dR_{k,c} R_k^T) A_p^T ; wave 2: landmark LBS ; waves 3-6: B_j columns (d P_j / d beta) ;
  //         (wave 7 stored and handed over the mesh operands in phase C) ----
  if (use_shape && want_jac) {
    for (int it = tid - 192; it >= 0 && tid < 448 && it < nS * (nJ - 1); it += 256) walk(2, 1 + it / nS, it % nS, true);
  }
  if (want_jac && tid < 3 * (nJ - 1)) {
    const int k = 1 + tid / 3, c = tid % 3, p = sParent[k];
    double T1[9], T2[9], Wm[9];
    mul33_bt(sdR + k * 27 + c * 9, sR + k * 9, T1);
    mul33(sA + p * 9, T1, T2);
    mul33_bt(T2, sA + p * 9, Wm);
#pragma unroll
    for (int e = 0; e < 9; ++e) sW[tid * 9 + e] = Wm[e];
  }
  if (wave == 2 && lane < nL) {
    double* L = sLm + lane * LM_STRIDE;
    const int* Lj = reinterpret_cast<const int*>(L + LM_J);
    double q[3] = {0, 0, 0}, Ab[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nw = (int)L[LM_NW];
    for (int i = 0; i < nw; ++i) {
      const int j = Lj[i];
      const double w = L[LM_W + i];
      double xj[3];
      mv3(sA + j * 9, L[LM_VP] - sJc[j * 3], L[LM_VP + 1] - sJc[j * 3 + 1], L[LM_VP + 2] - sJc[j * 3 + 2], xj);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        xj[a] += sP[j * 3 + a];
        q[a] += w * xj[a];
        L[LM_X + i * 3 + a] = xj[a];
      }
#pragma unroll
      for (int e = 0; e < 9; ++e) Ab[e] += w * sA[j * 9 + e];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) L[LM_Q + a] = q[a];
#pragma unroll
    for (int e = 0; e < 9; ++e) L[LM_A + e] = Ab[e];
  }
  if (wave == 2 && M.lm_gcount) {
    // keypoint regressor rows: the position of a row is the sum over its slots, left in the first one (LDS serves a wave in
    // order: the stores above are complete for every lane)
    const int cnt = (lane < nL) ? M.lm_gcount[lane] : 0;
    if (cnt > 1) {
      double* L = sLm + lane * LM_STRIDE;
      double q0 = L[LM_Q], q1 = L[LM_Q + 1], q2 = L[LM_Q + 2];
      for (int i = 1; i < cnt; ++i) {
        const double* Li = L + i * LM_STRIDE;
        q0 += Li[LM_Q]; q1 += Li[LM_Q + 1]; q2 += Li[LM_Q + 2];
      }
      L[LM_Q] = q0; L[LM_Q + 1] = q1; L[LM_Q + 2] = q2;
    }
  }
  frame_sync<kFused>();
  const double s = sx[0];
  const double* Rr0 = sCam;
  const double* dRr0 = sCam + 9;

  // keypoint stage of a chunk (phase F): camera point, residual, projection derivative per keypoint.  It depends on
  // nothing phase E produces, so the frame's first chunk runs on wave 6 (which has the least to do there) DURING E
  const int k_begin = k_begin0, k_end = k_begin0 + (Pb.kp_offset[f + 1] - k_begin0);
  const bool fold = Pb.beta_partials != nullptr && want_jac && ncols > npose && nS == kMaxShape;
  auto stage_kp = [&](int tid, int kc0) {   // `tid`: slot of the keypoint in its chunk
    {
      const int kg = kc0 + tid;
      const bool first = kc0 == k_begin;
      const int id = first ? sKpId[tid] : Pb.kp_id[kg];
      const double u_obs = first ? sKpUv[2 * tid] : Pb.kp_uv[2 * (size_t)kg];
      const double v_obs = first ? sKpUv[2 * tid + 1] : Pb.kp_uv[2 * (size_t)kg + 1];
      sKpId[tid] = id;
      sKpAnc[tid] = (id < nJ) ? sAnc[id] : 0u;    // the Jacobian sweep reads id and mask in one LDS round trip
      double q[3];
      if (id < nJ) {
        if (id == 0 || sParent[id] < 0) {
          // include/Sim3BA.h:142-170 without a chain: q = offset + S_id beta (no parent term), staged in phase B
#pragma unroll
          for (int a = 0; a < 3; ++a) q[a] = sPart[100 + a];
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a) q[a] = sP[id * 3 + a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = sLm[(id - nJ) * LM_STRIDE + LM_Q + a];
      }
      double z[3];
      mv3(Rr0, q[0], q[1], q[2], z);                     // include/Sim3BA.h:210-216
      const double X0 = s * z[0] + sx[4], X1 = s * z[1] + sx[5], X2 = s * z[2] + sx[6];  // :217-219
      const double iz = 1.0 / X2;                        // :222-223, Z unguarded as in the reference
      const double res0 = Pb.fx * X0 * iz + Pb.cx - u_obs, res1 = Pb.fy * X1 * iz + Pb.cy - v_obs;
      r_out[2 * (size_t)kg] = res0;
      r_out[2 * (size_t)kg + 1] = res1;
      if (fold) {   // kept for the folded beta reduction (the observation in sKpUv was consumed above; sFeat is dead)
        const double sq = res0 * res0 + res1 * res1, d2 = Pb.huber * Pb.huber;
        const bool outl = Pb.huber > 0.0 && sq > d2;
        const double rt = sqrt(sq);
        sKpUv[2 * tid] = res0;
        sKpUv[2 * tid + 1] = res1;
        sFeat[tid] = outl ? sqrt(Pb.huber / rt) : 1.0;                // sqrt(rho')
        sFeat[KC + tid] = 0.5 * (outl ? 2.0 * Pb.huber * rt - d2 : sq);   // 1/2 rho
      }
      double* kp = sKp + tid * 18;
      const double dpi[6] = {Pb.fx * iz, 0.0, -Pb.fx * X0 * iz * iz, 0.0, Pb.fy * iz, -Pb.fy * X1 * iz * iz};
#pragma unroll
      for (int a = 0; a < 3; ++a) { kp[a] = q[a]; kp[3 + a] = z[a]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) kp[6 + i] = dpi[i];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          kp[12 + rr * 3 + c] = s * (dpi[rr * 3] * Rr0[c] + dpi[rr * 3 + 1] * Rr0[3 + c] + dpi[rr * 3 + 2] * Rr0[6 + c]);
    }
  };
  STAMP(5);
  // ---- E. independent items: mesh operands + posed joints, landmark Jacobian terms, first keypoint chunk ----------
  if (wave == 6 && lane < min(KC, k_end - k_begin)) stage_kp(lane, k_begin);
  // (Measured and rejected, round 4: wave 7 — the least loaded here — writing the FK-joint keypoints' entries of the joint columns
  //  in this phase, ahead of phase F's sweep: they need none of this phase's landmark terms, and 42 % of the panel's stores would
  //  leave ~3 us earlier.  One wave's fourteen iterations of write-through stores took 16 k cycles — a wave's write-through
  //  stores are paced by their acknowledgements, ~600 cycles per store instruction here —: 23.0-23.3 against 21.3-21.8 us per
  //  step on the same box, profiles/r4_x10_stamps_early_fk_rows.txt.)
  if (nL > 0 && want_jac) {
    // d q_l / d theta_{k,c} for landmark l, complete, left in LM_PD[l][3 (k - 1) + c] for the Jacobian sweep:
    //   Ablend . (pd[:, 9(k-1):9k] . vec(dR_{k,c}))   the inner products were parked here by phase C (same lane mapping:
    //                                                 lane k - 1 of half-wave l); the blended rotation is applied now
    // + W_{k,c} . sum_{i : k is j_i or an ancestor of j_i} w_i (x_i - P_k)      the skinning term, which depends on the
    //                                                 landmark and the joint only, not on the keypoint that uses them
    for (int l = lm_l; l < nL; l += 16) {
      if (lm_k < nJ - 1) {
        const int k = lm_k + 1;
        const double* L = sLm + l * LM_STRIDE;
        const double* Ab = L + LM_A;
        const int* Lj = reinterpret_cast<const int*>(L + LM_J);
        const int nw = (int)L[LM_NW];
        const double pk0 = sP[k * 3], pk1 = sP[k * 3 + 1], pk2 = sP[k * 3 + 2];
        double a0 = 0, a1 = 0, a2 = 0;
        for (int i = 0; i < nw; ++i) {
          const int j = Lj[i];
          if (j == k || ((sAnc[j] >> k) & 1u)) {
            const double w = L[LM_W + i];
            a0 += w * (L[LM_X + i * 3] - pk0);
            a1 += w * (L[LM_X + i * 3 + 1] - pk1);
            a2 += w * (L[LM_X + i * 3 + 2] - pk2);
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double* o = sLm + l * LM_STRIDE + LM_PD + (3 * lm_k + c) * 3;
          const double* Wm = sW + (3 * lm_k + c) * 9;
          double t[3];
          mv3(Ab, o[0], o[1], o[2], t);
          o[0] = t[0] + Wm[0] * a0 + Wm[1] * a1 + Wm[2] * a2;
          o[1] = t[1] + Wm[3] * a0 + Wm[4] * a1 + Wm[5] * a2;
          o[2] = t[2] + Wm[6] * a0 + Wm[7] * a1 + Wm[8] * a2;
        }
      }
    }
    // shape columns  d q / d beta_k = sum_i w_i (A_j (sd_l - Sc_j) + B_j)[:, k] = Ablend sd_l[:, k] + sum_i w_i T_j[:, k]
    // item = (landmark, k); sd_l staged in LDS since phase A, T_j since phase D
    if (use_shape) {
      for (int it = kThreads - 1 - tid; it < nL * nS; it += kThreads) {
        const int l = it / nS, k = it % nS;
        const double* L = sLm + l * LM_STRIDE;
        const int* Lj = reinterpret_cast<const int*>(L + LM_J);
        double d[3];
        mv3(L + LM_A, sLmSd[(l * 3 + 0) * nS + k], sLmSd[(l * 3 + 1) * nS + k], sLmSd[(l * 3 + 2) * nS + k], d);
#pragma unroll
        for (int i = 0; i < kMaxLmNnz; ++i) {      // fixed stride, padded with weight 0 (joint 0)
          const int j = Lj[i];
          const double w = L[LM_W + i];
          d[0] += w * sT[(j * 3 + 0) * nS + k]; d[1] += w * sT[(j * 3 + 1) * nS + k]; d[2] += w * sT[(j * 3 + 2) * nS + k];
        }
        double* o = sLm + l * LM_STRIDE + LM_BETA + k * 3;
        o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
      }
    }
  }

  STAMP(6);
  if (M.lm_gcount && want_jac && nL > 0) {
    // keypoint regressor rows: d q / d theta (69 x 3) and d q / d beta (10 x 3) of a row are the sums over its slots, left in
    // the first one.  item = (slot, word of [LM_BETA, LM_PD + 207)); only models with regressor rows pay the two barriers.
    frame_sync<kFused>();
    constexpr int kWords = LM_PD + 207 - LM_BETA;     // 237 contiguous words: LM_BETA (30) then LM_PD (207)
    static_assert(LM_PD == LM_BETA + 30, "LM_BETA and LM_PD are adjacent");
    for (int it = tid; it < nL * kWords; it += kThreads) {
      const int l = it / kWords, w = it - l * kWords;
      const int cnt = M.lm_gcount[l];
      if (cnt > 1) {
        double* o = sLm + l * LM_STRIDE + LM_BETA + w;
        double v = o[0];
        for (int i = 1; i < cnt; ++i) v += o[i * LM_STRIDE];
        o[0] = v;
      }
    }
  }
  // ---- F. keypoints of this frame, KC at a time: stage per-keypoint data, then the flat (keypoint, column)
  //         sweep over all 512 threads (consecutive threads on consecutive columns of the row-major panel) ------
  double* sJb = sdR;            // [2 KC][10] d r / d beta of the chunk; dR is dead after phase D
  double fold_acc = 0.0;
  typedef __attribute__((ext_vector_type(4))) double fold_d4;
  fold_d4 fold_gram = {0.0, 0.0, 0.0, 0.0};
  for (int kc0 = k_begin; kc0 < k_end; kc0 += KC) {
    const int nk = min(KC, k_end - kc0);
    frame_sync<kFused>();   // phase E results visible / previous chunk's staging consumed
    if (kc0 != k_begin) {   // (the first chunk was staged by wave 6 during phase E: the barrier above published it)
      if (tid < nk) stage_kp(tid, kc0);
      frame_sync<kFused>();
    }
    STAMP(7);
    if (want_jac) {
      // (1) joint columns: thread = (column kc, keypoint group g).  W_{k,c} and P_k stay in registers for
      //     all keypoints of the group; consecutive threads write consecutive columns of a row; the
      //     FK / landmark branch is uniform across the threads of a group.
      const int njc = npose - 7;
      const int ngrp = kThreads / njc;                      // 7 groups of 69 columns
      if (tid < ngrp * njc) {
        const int g = tid / njc, kc = tid - g * njc, k = 1 + kc / 3;
        double Wm[9];
#pragma unroll
        for (int e9 = 0; e9 < 9; ++e9) Wm[e9] = sW[kc * 9 + e9];
        const double pk0 = sP[k * 3], pk1 = sP[k * 3 + 1], pk2 = sP[k * 3 + 2];
        for (int kk = g; kk < nk; kk += ngrp) {
          const int kg = kc0 + kk;
          const int id = sKpId[kk];
          const unsigned am = sKpAnc[kk];
          const double* kp = sKp + kk * 18;
          const double* G = kp + 12;
          double d0 = 0.0, d1 = 0.0, d2 = 0.0;
          if (id < nJ) {
            if ((am >> k) & 1u) {
              const double x0 = kp[0] - pk0, x1 = kp[1] - pk1, x2 = kp[2] - pk2;
              d0 = Wm[0] * x0 + Wm[1] * x1 + Wm[2] * x2;
              d1 = Wm[3] * x0 + Wm[4] * x1 + Wm[5] * x2;
              d2 = Wm[6] * x0 + Wm[7] * x1 + Wm[8] * x2;
            }
          } else {
            const double* o = sLm + (id - nJ) * LM_STRIDE + LM_PD + kc * 3;   // complete since phase E
            d0 = o[0]; d1 = o[1]; d2 = o[2];
          }
          store_J<kFused>(J_out + (size_t)(2 * kg) * ncols + 7 + kc, G[0] * d0 + G[1] * d1 + G[2] * d2, fu.j_scope);
          store_J<kFused>(J_out + (size_t)(2 * kg + 1) * ncols + 7 + kc, G[3] * d0 + G[4] * d1 + G[5] * d2, fu.j_scope);
        }
      }
      // (2) Sim3 columns (7) and shape columns (ncols - npose) per keypoint
      const int nsc = 7 + (ncols - npose);
      for (int e = tid; e < nk * nsc; e += kThreads) {
        const int kk = e / nsc, c = e - kk * nsc;
        const int kg = kc0 + kk;
        const int id = sKpId[kk];
        const double* kp = sKp + kk * 18;
        const double* G = kp + 12;
        double j0, j1;
        int col;
        if (c == 0) {
          col = 0;
          j0 = kp[6] * kp[3] + kp[7] * kp[4] + kp[8] * kp[5];
          j1 = kp[9] * kp[3] + kp[10] * kp[4] + kp[11] * kp[5];
        } else if (c < 4) {
          col = c;
          double t[3];
          mv3(dRr0 + (c - 1) * 9, kp[0], kp[1], kp[2], t);
          j0 = s * (kp[6] * t[0] + kp[7] * t[1] + kp[8] * t[2]);
          j1 = s * (kp[9] * t[0] + kp[10] * t[1] + kp[11] * t[2]);
        } else if (c < 7) {
          col = c;
          j0 = kp[6 + (c - 4)];
          j1 = kp[9 + (c - 4)];
        } else {
          const int k = c - 7;
          col = npose + k;
          double d0 = 0.0, d1 = 0.0, d2 = 0.0;
          if (use_shape) {
            if (id < nJ) {
              if (id == 0 || sParent[id] < 0) {
                d0 = sDS[(id * 3 + 0) * nS + k]; d1 = sDS[(id * 3 + 1) * nS + k]; d2 = sDS[(id * 3 + 2) * nS + k];
              } else {
                d0 = sB[(id * 3 + 0) * nS + k]; d1 = sB[(id * 3 + 1) * nS + k]; d2 = sB[(id * 3 + 2) * nS + k];
              }
            } else {
              const double* o = sLm + (id - nJ) * LM_STRIDE + LM_BETA + k * 3;
              d0 = o[0]; d1 = o[1]; d2 = o[2];
            }
          }
          j0 = G[0] * d0 + G[1] * d1 + G[2] * d2;
          j1 = G[3] * d0 + G[4] * d1 + G[5] * d2;
        }
        store_J<kFused>(J_out + (size_t)(2 * kg) * ncols + col, j0, fu.j_scope);
        store_J<kFused>(J_out + (size_t)(2 * kg + 1) * ncols + col, j1, fu.j_scope);
        if (fold && c >= 7) { sJb[(2 * kk) * kMaxShape + c - 7] = j0; sJb[(2 * kk + 1) * kMaxShape + c - 7] = j1; }
      }
    }
    if (fold) {
      // shared-beta reduction of this frame (k_reduce.hip's definition): the Gram matrix of the robustified rows
      // [sqrt(rho') J_beta | sqrt(rho') r] holds H_bb (upper 10 x 10) and g_beta (column 10); 4 rows per
      // v_mfma_f64_16x16x4_f64 with the same register as A and B operand, wave 0 only; the cost by one lane of wave 1
      frame_sync<kFused>();
      if (wave == 0) {
        const int col = lane & 15, kq = lane >> 4;
        for (int s0 = 0; s0 < (2 * nk + 3) / 4; s0 += 4) {   // four steps per pass: their LDS reads are in flight together
          double vv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int row = 4 * (s0 + u) + kq;
            const bool on = row < 2 * nk;
            const double sw = on ? sFeat[row >> 1] : 0.0;
            const double x = (on && col < nS) ? sJb[row * kMaxShape + col] : ((on && col == 10) ? sKpUv[row] : 0.0);
            vv[u] = sw * x;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) fold_gram = __builtin_amdgcn_mfma_f64_16x16x4f64(vv[u], vv[u], fold_gram, 0, 0, 0);
        }
      } else if (tid == 64) {
        for (int kk = 0; kk < nk; ++kk) fold_acc += sFeat[KC + kk];
      }
    }
  }
  if (fold) {
    double* out = Pb.beta_partials + (size_t)f * kReducePartial;
    // (one-launch sweep: write-through, the launch's last frame / prior workgroup may sum the partials, k_sweep.hip fold_tail)
    if (wave == 0) {     // D layout (f64 16x16): column = lane & 15, row = (lane >> 4) + 4 q; kept: (i, j), i < 10, i <= j <= 10
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gi = (lane >> 4) + 4 * q, gj = lane & 15;
        if (gi < 10 && gj >= gi && gj <= 10) {
          double* o = out + fold_slot_gram(gi, gj);
          if constexpr (kFused) store_f64_through(o, fold_gram[q]); else *o = fold_gram[q];
        }
      }
    } else if (tid == 64) {
      if constexpr (kFused) store_f64_through(out + fold_slot_cost(0), fold_acc); else out[fold_slot_cost(0)] = fold_acc;
    }
  }
  STAMP(8);
  STAMP_REAL(11);
}

constexpr size_t frame_lds_bytes(int nL) { return (size_t)(OFF_LM + nL * (LM_STRIDE + 3 * kMaxShape)) * sizeof(double); }

}  // namespace
}  // namespace bodyfit
