// k_sweep.hip — the kernels of one evaluation sweep (include/Sim3BA.h:126-227 evaluated for every block of every frame,
// plus the 6890-vertex forward of ark::Avatar::update(), call sites include/MultiFrameBA.h:53,173):
//
//   k_frame_resjac    frame_part_inl.h as its own launch: residuals + analytic Jacobian, one workgroup per frame
//   k_mesh_blend_lbs  mesh_part_inl.h as its own launch: one workgroup per 32-vertex tile, all frames
//   k_sweep_fused     the whole sweep as ONE launch (frames <= 256, i.e. one workgroup per CU): every workgroup runs the
//                     frame part of "its" frame, hands that frame's mesh operands (blend coefficients, skinning
//                     transforms: 2 KB) to all workgroups inside the launch, waits until every frame has been handed
//                     over, and runs the mesh part of "its" vertex tile.  Against the two launches this removes the
//                     dependent kernel boundary between them and hides the mesh part's start-up and its 84 KiB operand
//                     staging (HBM -> LDS) under the frame part.
//
// In-launch hand-off (cdna guide, Guideline 16, R1): producers store the operands write-through (sc1), every storing wave
// drains (s_waitcnt vmcnt(0)), workgroup barrier, one lane stores the frame's flag = the launch's epoch (sc1); consumers
// poll the F flags with sc1 loads from one wave (one 16-byte load per lane covers 256 flags), workgroup barrier, then read
// the operands with sc1 loads only.  Nothing is reset between launches (flags and claims carry the launch's epoch) and the
// normal path has no read-modify-write at all: 256 workgroups arriving on one counter serialise at ~12 ns each.
//
// Progress does not depend on dispatch order or on every workgroup being resident.  A workgroup marks frame blockIdx.x as
// started (claim[f] = epoch, a plain write-through store) and processes it; while it waits for the flags it ADOPTS, after
// a grace period, frames nobody has started (atomic exchange on claim[f] arbitrates between adopters), so the frames of
// workgroups that have not been dispatched yet (another process holding CUs, fewer CUs than workgroups) are processed by
// the resident ones, and the wait ends.  A frame processed twice (adopted, then run by its late owner) is written twice
// with identical bytes and its flag store is idempotent.  Every spin is bounded: after kFusedTimeoutTicks the workgroup sets
// the problem's error word and leaves.
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "frame_part_inl.h"
#include "mesh_part_inl.h"
#include "priors_inl.h"

namespace bodyfit {
namespace {

// LDS of the fused sweep: [0, 86,016) the tile operands (as in k_mesh_blend_lbs), [86,016, 163,840) the frame part's state,
// later the mesh part's transform slices.
constexpr int kFusedFrameLds = 77824;
constexpr int kFusedLdsBytes = kBBytes + kFusedFrameLds;     // 163,840 = the CU's 160 KiB
constexpr int kFusedCtrlOff = kFusedLdsBytes - 16;           // control words of the wait loop
[[maybe_unused]] constexpr unsigned long long kFusedStealTicks = 3000;        // 30 us of s_memrealtime (100 MHz) before adopting frames
[[maybe_unused]] constexpr unsigned long long kFusedTimeoutTicks = 5000000;   // 50 ms: give up, set the error word
static_assert(kWaves * kQuarterBytes <= kFusedCtrlOff - kBBytes, "transform slices fit behind the tile operands");

__global__ __launch_bounds__(kThreads, 4) void k_frame_resjac(DevModel M, DevProblem Pb, const double* __restrict__ params,
                                                      const double* __restrict__ beta, double* __restrict__ r_out,
                                                      double* __restrict__ J_out, double* __restrict__ joints_out,
                                                      MeshCoef mc, int want_jac, PriorArgs pa) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  if ((int)blockIdx.x < pa.n_tiles) {   // first workgroups of the launch: prior residuals of one 16-frame tile
#ifdef BODYFIT_STAMPS
    unsigned long long tp0 = 0;
    if (Pb.dbg && threadIdx.x == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tp0)::"memory");
#endif
    if (Pb.frame_flags) {
      // device LM: a tile none of whose 16 frames has a candidate has nothing to evaluate (every wave reads the same 16
      // flags, so the decision is uniform over the workgroup) — late iterations of a batch keep only a few tiles
      const int f = (int)blockIdx.x * kPriorTileF + (int)(threadIdx.x & 15);
      const int fl = (f < Pb.F) ? Pb.frame_flags[f] : 0;
      if (__ballot((fl & Pb.frame_mask) != 0) == 0ull) return;
    }
    prior_block(pa, (int)blockIdx.x, params, sm);   // (dispatched first so they never form the tail of the launch)
#ifdef BODYFIT_STAMPS
    if (Pb.dbg && threadIdx.x == 0) {
      unsigned long long tp1;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tp1)::"memory");
      Pb.dbg[((size_t)Pb.F * 8 + blockIdx.x) * 16 + 0] = tp0;
      Pb.dbg[((size_t)Pb.F * 8 + blockIdx.x) * 16 + 1] = tp1;
    }
#endif
    return;
  }
  const FusedFrame none{};
  // a frame workgroup that shares its CU with a prior workgroup (256 frames + 16 prior tiles on 256 CUs) is the launch's
  // critical path; the prior workgroup has slack: frame waves win the issue arbitration
  __builtin_amdgcn_s_setprio(2);
  frame_part<false>(M, Pb, params, beta, r_out, J_out, joints_out, mc, want_jac, sm, (int)blockIdx.x - pa.n_tiles, none);
}

__global__ __launch_bounds__(64 * kWaves) void k_mesh_blend_lbs(DevModel M, DevProblem Pb, MeshCoef mc,
                                                                   float* __restrict__ cloud_f, PriorArgs pa,
                                                                   const double* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if ((int)blockIdx.x >= M.nVTiles) {
    // The vertex tiles occupy 216 of the 256 CUs; the sweep's prior residuals (one 16-frame tile per workgroup,
    // priors_inl.h) ride on the idle ones instead of doubling up with k_frame_resjac's frame workgroups.
    prior_block(pa, (int)blockIdx.x - M.nVTiles, params, reinterpret_cast<double*>(lds));
    return;
  }
  mesh_part<false>(M, Pb, mc, cloud_f, (int)blockIdx.x, lds, lds + kBBytes);
}

// All arguments of the fused sweep, passed BY VALUE as one struct and read through the kernel-argument segment pointer where
// they are used.  (As separate by-value arguments hipcc loads all ~110 argument SGPRs at kernel entry and, the frame part and
// the mesh part each needing most of the register file, carries them across both as spills: 314 SGPR + 95 VGPR spills.)
struct FusedArgs {
  DevModel M;
  DevProblem Pb;
  const double* params;
  const double* beta;
  double* r_out;
  double* J_out;
  double* joints_out;
  MeshCoef mc;
  int want_jac;
  PriorArgs pa;
  float* cloud_f;
  FusedSync sy;
};
typedef const __attribute__((address_space(4))) FusedArgs* FusedArgP;
// the same pointer, opaque to the optimiser: loads through the result are neither merged with earlier ones nor hoisted
__device__ __forceinline__ FusedArgP reload_args(FusedArgP p) {
  asm volatile("" : "+s"(p));
  return p;
}

// The frame part of the fused sweep as a CALL, not inlined: as part of the kernel's body it is scheduled and register-
// allocated together with the mesh part (248 VGPRs, ~110 argument SGPRs live), and hipcc then serialises the LDS reads of
// its latency chains (the chain walk: four LDS round trips per level instead of one) and spills; as a function of its own
// it is compiled like k_frame_resjac.  Arguments come from the kernel-argument segment, not through the call.
__device__ __attribute__((noinline)) void fused_frame_call(FusedArgP A, double* smF, int f, unsigned char* ldsB,
                                                           const unsigned char* dirs, unsigned epoch) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass cannot copy structs out of address space 4)
  const DevModel M = A->M;
  const DevProblem Pb = A->Pb;
  const MeshCoef mc = A->mc;
  FusedFrame fu;
  fu.ldsB = ldsB;
  fu.dirsB = dirs;
  fu.flag = A->sy.flag;
  fu.epoch = epoch;
  frame_part<true>(M, Pb, A->params, A->beta, A->r_out, A->J_out, A->joints_out, mc, A->want_jac, smF, f, fu);
#endif
}

#ifdef BODYFIT_STAMPS
// diagnostic build: workgroup-level s_memrealtime stamps of the fused sweep (tools/stamp_fused.py)
#define FSTAMP(i)                                                                                      \
  do {                                                                                                 \
    if (A->Pb.dbg && tid == 0) {                                                                       \
      unsigned long long t_;                                                                           \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
      A->Pb.dbg[((size_t)1 << 20) + ((size_t)1 << 16) + (size_t)b * 8 + (i)] = t_;                     \
    }                                                                                                  \
  } while (0)
#else
#define FSTAMP(i)
#endif

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_fused(FusedArgs by_value) {
  (void)by_value;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the stub; it cannot copy structs out of address space 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const FusedArgP A = (FusedArgP)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = (int)blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nVTiles = A->M.nVTiles, F = A->Pb.F;
  const bool has_tile = b < nVTiles;
  unsigned char* ldsB = lds;
  double* smF = reinterpret_cast<double*>(lds + kBBytes);
  volatile unsigned* ctrl = reinterpret_cast<volatile unsigned*>(lds + kFusedCtrlOff);
  const unsigned char* dirs = reinterpret_cast<const unsigned char*>(A->M.dirsB) + (size_t)(has_tile ? b : 0) * kBBytes;
  const unsigned epoch = A->sy.epoch;
  const int test_skip = A->sy.test_skip;
  FSTAMP(0);

  // ---- frames: this workgroup's own first, then (only while the counter is short after a grace period) adopted ones ---
  // One call site of the frame part: `f` is the frame to process in this round, or -1.
  const bool own = b < F && !(test_skip > 0 && b % test_skip == 1);
  int f = own ? b : -1;
  if (own && tid == kThreads - 1) __hip_atomic_store(A->sy.claim + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool staged = false;
  if (has_tile && !own) {   // no frame of its own: stage the tile operands now (all waves)
#pragma unroll 1
    for (int pc = wave; pc < kPieces; pc += kWaves)
      __builtin_amdgcn_global_load_lds(dirs + (size_t)pc * 1024 + lane * 16,
                                       (__attribute__((address_space(3))) void*)(ldsB + (size_t)pc * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first barrier of the loop below publishes it to the other waves)
    staged = true;
  }
  unsigned long long t_enter = 0;
  uint32_t widx_pre = 0;
  float4 wv_pre = float4{0.f, 0.f, 0.f, 0.f};
  bool have_w = false;
  for (;;) {
    if (f >= 0) {
      // (a workgroup with a frame of its own stages its tile under it)
      fused_frame_call(A, smF, f, (has_tile && !staged) ? ldsB : nullptr, dirs, epoch);
      staged = true;   // (requested in the frame part's phase D; waited for in front of the mesh part)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // frame part's LDS is free
      FSTAMP(1);
    }
    if (!has_tile) break;   // no mesh part: nothing to wait for (the prior tiles below ride on these workgroups)
    if (!have_w) {          // the mesh part's per-lane constants: requested here, they arrive under the wait
      const int col = lane & 31;
      widx_pre = A->M.wIdx[(size_t)b * 32 + col];
      wv_pre = reinterpret_cast<const float4*>(A->M.wVal)[(size_t)b * 32 + col];
      have_w = true;
    }
    // wait until every frame's mesh operands have been handed over; adopt an unclaimed frame after a grace period
    if (wave == 0) {
      // every lane polls four flags with one 16-byte sc1 load (256 flags per wave-instruction); lane 0 runs the adoption scan
      const FusedArgP A2 = reload_args(A);
      unsigned* const claim = A2->sy.claim;
      const __amdgpu_buffer_rsrc_t flags = __builtin_amdgcn_make_buffer_rsrc(A2->sy.flag, 0, kFusedMaxFrames * 4, 0x00020000);
      unsigned action = 0, fs = 0;
      if (t_enter == 0) t_enter = __builtin_amdgcn_s_memrealtime();
      unsigned long long t_grace = __builtin_amdgcn_s_memrealtime();
      const unsigned scan = (unsigned)b * 37u;
      for (;;) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
        const u32x4_ fl = __builtin_amdgcn_raw_buffer_load_b128(flags, (unsigned)lane * 16u, 0, 16);
        const int f4 = lane * 4;
        const bool ok = (f4 >= F || fl.x == epoch) && (f4 + 1 >= F || fl.y == epoch) && (f4 + 2 >= F || fl.z == epoch) &&
                        (f4 + 3 >= F || fl.w == epoch);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (now - t_enter > kFusedTimeoutTicks) { action = 2; break; }
        if (now - t_grace > kFusedStealTicks) {
          // adopt one frame nobody has started in this launch (its workgroup is not resident yet)
          unsigned got = 0;
          if (lane == 0) {
            for (int i = 0; i < F && !got; ++i) {
              const unsigned fcand = (scan + (unsigned)i) % (unsigned)F;
              if (__hip_atomic_load(claim + fcand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch &&
                  __hip_atomic_exchange(claim + fcand, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                got = 1; fs = fcand;
              }
            }
          }
          got = __builtin_amdgcn_readfirstlane(got);
          if (got) { action = 1; break; }
          t_grace = now;   // every frame has been started: its workgroup is running, keep polling
        }
      }
      if (lane == 0) { ctrl[0] = action; ctrl[1] = fs; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned action = ctrl[0], fs = ctrl[1];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // control words read before they are rewritten
    if (action == 0) break;
    if (action == 2) {
      if (tid == 0) __hip_atomic_store(A->sy.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    f = (int)fs;
  }
  // ---- prior residuals ride on the workgroups that have no vertex tile -------------------------------------------
  if (!has_tile) {
    const int pt = b - nVTiles;
    const FusedArgP A3 = reload_args(A);
    const PriorArgs pa = A3->pa;
    if (pt < pa.n_tiles) prior_block(pa, pt, A3->params, smF);
    return;
  }
  // the tile operands have landed (requested at least a phase F earlier; this also retires the frame part's last Jacobian
  // stores, which the mesh part's first loads would wait for anyway: vmcnt retires in order)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  FSTAMP(2);
  // (the barriers above separate the polling lane's last flag read from EVERY load of the handed-off operands, and the
  //  frame part's last LDS use from the transform slices that alias it)
  {
    const FusedArgP A4 = reload_args(A);
    const DevModel M = A4->M;
    const DevProblem Pb = A4->Pb;
    const MeshCoef mc = A4->mc;
    mesh_part<true>(M, Pb, mc, A4->cloud_f, b, ldsB, lds + kBBytes, widx_pre, wv_pre);
  }
  FSTAMP(3);
#endif
}

}  // namespace

void launch_frame_resjac(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta,
                         double* d_r, double* d_J, double* d_joints, const MeshCoef& mc, int want_jac,
                         const PriorArgs& priors, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (P.F <= 0) return;
  const size_t lds = frame_lds_bytes(M.nL);
  static size_t lds_granted = 48 * 1024;
  if (lds > lds_granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_resjac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    lds_granted = lds;
  }
  hipExtLaunchKernelGGL(k_frame_resjac, dim3(P.F + priors.n_tiles), dim3(kThreads), lds, s, ev_start, ev_stop, 0, M, P,
                        d_params, d_beta, d_r, d_J, d_joints, mc, want_jac, priors);
}

void launch_mesh(const DevModel& M, const DevProblem& P, const MeshCoef& mc, float* d_cloud, const PriorArgs& pa,
                 const double* d_params, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (P.F <= 0) return;
  if ((size_t)P.nFTiles * kFTile * M.nVTiles * kVTile * 12 >= ((size_t)1 << 32)) return;   // refused at problem creation
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mesh_blend_lbs),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipExtLaunchKernelGGL(k_mesh_blend_lbs, dim3(M.nVTiles + pa.n_tiles), dim3(64 * kWaves), kLdsBytes, s, ev_start, ev_stop,
                        0, M, P, mc, d_cloud, pa, d_params);
}

// Whether one launch can carry the sweep: one workgroup per CU holds the tile operands AND the frame part's LDS, and the
// grid is one workgroup per frame / vertex tile / prior tile.
bool fused_sweep_fits(const DevModel& M, const DevProblem& P, int n_prior_tiles, int n_cus) {
  const int grid = P.F > M.nVTiles + n_prior_tiles ? P.F : M.nVTiles + n_prior_tiles;
  return P.F > 0 && P.F <= kFusedMaxFrames && grid <= n_cus && frame_lds_bytes(M.nL) <= (size_t)(kFusedCtrlOff - kBBytes) &&
         (size_t)P.nFTiles * kFTile * M.nVTiles * kVTile * 12 < ((size_t)1 << 32);
}

void launch_sweep_fused(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta, double* d_r,
                        double* d_J, double* d_joints, const MeshCoef& mc, int want_jac, const PriorArgs& pa, float* d_cloud,
                        const FusedSync& sy, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kFusedLdsBytes);
    attr_set = true;
  }
  const int grid = P.F > M.nVTiles + pa.n_tiles ? P.F : M.nVTiles + pa.n_tiles;
  FusedArgs A;
  A.M = M; A.Pb = P; A.params = d_params; A.beta = d_beta; A.r_out = d_r; A.J_out = d_J; A.joints_out = d_joints;
  A.mc = mc; A.want_jac = want_jac; A.pa = pa; A.cloud_f = d_cloud; A.sy = sy;
  hipExtLaunchKernelGGL(k_sweep_fused, dim3(grid), dim3(kThreads), kFusedLdsBytes, s, ev_start, ev_stop, 0, A);
}

}  // namespace bodyfit
