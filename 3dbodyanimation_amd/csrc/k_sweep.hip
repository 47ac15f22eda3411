// k_sweep.hip — the kernels of one evaluation sweep (include/Sim3BA.h:126-227 evaluated for every block of every frame,
// plus the 6890-vertex forward of ark::Avatar::update(), call sites include/MultiFrameBA.h:53,173):
//
//   k_frame_resjac    frame_part_inl.h as its own launch: residuals + analytic Jacobian, one workgroup per frame
//   k_mesh_blend_lbs  mesh_part_inl.h as its own launch: one workgroup per 32-vertex tile, all frames
//   k_sweep_roles     the whole sweep as ONE launch whose workgroups take one of three roles (by block index):
//                       frame role  frame_part_inl.h for one frame; publishes the frame's mesh operands (blend coefficients,
//                                   skinning transforms: 2 KB) inside the launch
//                       mesh role   mesh_role_inl.h for one 32-vertex tile x one group of 256 frames; each of its eight waves
//                                   waits for ITS 32 frames' blend coefficients, blends, checks their transforms, skins
//                       prior role  priors_inl.h for one 16-frame tile
//                     Every role fits 128 VGPRs and 80 KiB of LDS, so TWO workgroups share a CU: at 256 frames all 256 frame
//                     workgroups (a latency chain that issues on a few percent of its slots) and all 216 mesh workgroups
//                     (matrix pipe, LDS, stores) are resident together and run at the same time; with more frames the
//                     block order [frames g][frames g+1][mesh g][frames g+2][mesh g+1]... keeps both kinds on every CU.
//                     Against the two launches this removes the dependent kernel boundary and, above all, overlaps the two
//                     halves of the sweep in time instead of running them one after the other.
//
// In-launch hand-off (cdna guide, Guideline 16, R1): producers store the operands write-through (sc1); each of the two storing
// waves of a frame workgroup (wave 0: blend coefficients, wave 7: skinning transforms) drains (s_waitcnt vmcnt(0)) and then adds
// 1 to ITS counter of the frame's 32-frame unit (agent-scope atomics, executed at the memory side; two counters per unit, each
// on a line of its own).  A mesh wave polls the coefficient counter of its own unit with sc1 loads, starts its blend, looks at
// the transform counter a few k-steps before it needs the transforms, and reads all operands with sc1 loads only; the prior
// workgroups wait for the transform counters of their whole group.  Nothing is reset between launches: every launch adds
// exactly the unit's frame count, so a complete unit reads epoch x count.
//
// Progress: a mesh workgroup waits only for frame workgroups, which wait for nothing; frame workgroups precede the mesh
// workgroups that need them in block order, and the hardware dispatches blocks in order.  HIP does not promise that
// order, so every wait is bounded — by POLLS (FusedSync::timeout_ticks / 110: a poll is ~1.1 us of a running wave), not by wall
// time since role entry: a wave that is descheduled (several processes time-slicing one GPU, serialised counter passes) does
// not poll, so preemption cannot make a wait expire.  A mesh workgroup whose wait runs out sets the problem's error word and
// leaves (its tile of the cloud is not written); the synchronous entry points then re-issue the sweep as two launches, the
// asynchronous ones report it through bodyfit_sweep_status.  A prior workgroup whose wait runs out simply goes on: it waits
// only to keep out of the frame workgroups' way and needs nothing they produce, so r, J, the joints and the folded
// reduction are complete whatever happens to the waits.
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "frame_part_inl.h"
#include "mesh_part_inl.h"
#include "mesh_role_inl.h"
#include "priors_inl.h"

namespace bodyfit {
namespace {

[[maybe_unused]] constexpr int kPollWave = 7;

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
  const FrameHead no_head{};
  frame_part<false>(M, Pb, params, beta, r_out, J_out, joints_out, mc, want_jac, sm, (int)blockIdx.x - pa.n_tiles, none, no_head);
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
  mesh_part(M, Pb, mc, cloud_f, (int)blockIdx.x, lds, lds + kBBytes);
}

// All arguments of the one-launch sweep, passed BY VALUE as one struct and read through the kernel-argument segment pointer
// where they are used.  (As separate by-value arguments hipcc loads all ~110 argument SGPRs at kernel entry and carries them
// across the roles as spills.)
// Last frame / prior workgroup of a folding launch: [cost | g_beta | upper H_bb] from the per-workgroup partials, in
// k_reduce_stage2's order of additions (16 slices of partials summed as a tree of 16, then the slices in order), so the
// folded and the separately launched reduction agree to the last bit.  Called by every wave of a frame / prior workgroup once
// its own partial is stored (write-through).  lds: >= 16 * 67 * 8 + 16 bytes, free at this point.
__device__ __forceinline__ void fold_tail(unsigned* ticket, unsigned want, int n_partials, const double* partials,
                                          const double* beta, int shape_rows, double beta_shape, double* out66,
                                          unsigned char* lds) {
  constexpr int kEntries = kFoldEntries;             // 2 cost entries + the 65 entries (i, j), i < 10, i <= j <= 10 of the Gram tile
  const int tid = threadIdx.x;
  volatile unsigned* last = reinterpret_cast<volatile unsigned*>(lds);
  double* sred = reinterpret_cast<double*>(lds + 16);   // [16][kEntries]
  // every storing wave's partial has left (cdna guide, Guideline 16 R1), then ONE ticket per workgroup
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last[0] = (old + 1u == want) ? 1u : 0u;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (last[0] == 0u) return;
  auto entry_of = [](int t) { return t; };          // (the sweep's partial rows are compact: bodyfit_device.h kFoldEntries)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(partials), 0,
                                                                      n_partials * kReducePartial * 8, 0x00020000);
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
  constexpr int kItems = 16 * kEntries, kPasses = (kItems + kThreads - 1) / kThreads;
  // every load of the workgroup in ONE batch (the tail is a chain of round trips to the memory side: ticket, partials):
  // two items per thread and a third for the first kItems - 2 kThreads threads
  static_assert(kPasses == 3 && kItems - 2 * kThreads <= 64, "fold_tail: item split");
  u32x2_ a0[16], a1[16], a2[16];
  {
    const int s0 = tid / kEntries, e0 = entry_of(tid % kEntries);
    const int i1 = tid + kThreads, s1 = i1 / kEntries, e1 = entry_of(i1 % kEntries);
#pragma unroll
    for (int u = 0; u < 16; ++u) {   // sc1: written by workgroups of other XCDs in this launch; rows past n_partials read 0
      a0[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(((s0 + 16 * u) * kReducePartial + e0) * 8), 0, 16);
      a1[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(((s1 + 16 * u) * kReducePartial + e1) * 8), 0, 16);
    }
    if (tid < kItems - 2 * kThreads) {   // (wave 0 only)
      const int i2 = tid + 2 * kThreads, s2 = i2 / kEntries, e2 = entry_of(i2 % kEntries);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        a2[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(((s2 + 16 * u) * kReducePartial + e2) * 8), 0, 16);
    }
  }
  auto tree = [](const u32x2_ (&a)[16]) {
    auto d = [&](int u) { return __longlong_as_double((long long)(((unsigned long long)a[u].y << 32) | (unsigned long long)a[u].x)); };
    return (((d(0) + d(1)) + (d(2) + d(3))) + ((d(4) + d(5)) + (d(6) + d(7)))) +
           (((d(8) + d(9)) + (d(10) + d(11))) + ((d(12) + d(13)) + (d(14) + d(15))));
  };
  sred[tid] = tree(a0);
  sred[tid + kThreads] = tree(a1);
  if (tid < kItems - 2 * kThreads) sred[tid + 2 * kThreads] = tree(a2);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (tid < 66) {
    auto T = [&](int t) {
      double v = 0.0;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) v += sred[sl * kEntries + t];
      return v;
    };
    auto G = [&](int i, int j) { return T(fold_slot_gram(i, j)); };
    double v;
    if (tid == 0) {
      v = T(0) + T(1);
    } else if (tid < 11) {
      v = G(tid - 1, 10);
      if (tid - 1 < shape_rows) v += beta_shape * (beta_shape * beta[tid - 1]);   // shared shape prior: r = beta_s beta, J = beta_s I
    } else {
      int row = 0, rem = tid - 11;
      while (rem >= 10 - row) { rem -= 10 - row; ++row; }
      v = G(row, row + rem);
      if (rem == 0 && row < shape_rows) v += beta_shape * beta_shape;
    }
    out66[tid] = v;
  }
}

struct RoleArgs {
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
  FoldTail fold;
};
typedef const __attribute__((address_space(4))) RoleArgs* RoleArgP;

// F and nVT, which the role of a block is decoded from, are leading scalar arguments: with -amdgpu-kernarg-preload-count they are in
// SGPRs when the wave starts, and the decode no longer waits for a scalar load of its own in front of the role's operand loads.
// So is everything phase A of the frame role loads from (FrameHead, frame_part_inl.h): four integers and five pointers, the
// fourteen dwords the hardware preloads.
static_assert(alignof(RoleArgs) == 8, "kernel-argument segment: 4 ints, 5 pointers, then RoleArgs at offset 56");
[[maybe_unused]] constexpr int kRoleArgsOffset = 4 * 4 + 5 * 8;
__global__ __launch_bounds__(kThreads, 4) void k_sweep_roles(int F_arg, int nVT_arg, int K_arg, int dims_arg,
                                                             const unsigned char* __restrict__ mtab_arg,
                                                             const unsigned char* __restrict__ ptab_arg,
                                                             const double* __restrict__ R0_arg, const double* __restrict__ params_arg,
                                                             const double* __restrict__ beta_arg, RoleArgs by_value) {
  (void)by_value;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the stub; it cannot copy structs out of address space 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const RoleArgP A =
      (RoleArgP)((const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + kRoleArgsOffset);
  const int F = F_arg, nVT = nVT_arg;
  // ---- role of this block: [frames 0][frames 1][mesh 0][frames 2][mesh 1] ... [mesh nG-1][prior tiles] ----------------
  int role = 2, idx = 0, grp = 0;
  {
    const int nG = (F + kRoleGroup - 1) / kRoleGroup;
    int pos = (int)blockIdx.x;
    const int n0 = min(kRoleGroup, F);
    if (pos < n0) { role = 0; idx = pos; }
    else {
      pos -= n0;
      bool found = false;
      for (int g = 1; g < nG && !found; ++g) {
        const int ng = min(kRoleGroup, F - kRoleGroup * g);
        if (pos < ng) { role = 0; idx = kRoleGroup * g + pos; found = true; }
        else {
          pos -= ng;
          if (pos < nVT) { role = 1; idx = pos; grp = g - 1; found = true; }
          else pos -= nVT;
        }
      }
      if (!found) {
        if (pos < nVT) { role = 1; idx = pos; grp = nG - 1; }
        else { role = 2; idx = pos - nVT; }
      }
    }
  }
  if (role == 0) {
    const DevModel M = A->M;
    const DevProblem Pb = A->Pb;
    const MeshCoef mc = A->mc;
    FusedFrame fu;
    fu.flag = A->sy.flag;
    fu.epoch = A->sy.epoch;
    fu.j_scope = A->sy.j_scope;
    // a frame workgroup is the launch's critical path; the mesh workgroup it shares the CU with has slack
    __builtin_amdgcn_s_setprio(2);
    FrameHead hd;
    hd.mtab = mtab_arg; hd.ptab = ptab_arg; hd.R0 = R0_arg; hd.F = F_arg; hd.K = K_arg; hd.dims = dims_arg;
    frame_part<true>(M, Pb, params_arg, beta_arg, A->r_out, A->J_out, A->joints_out, mc, A->want_jac,
                     reinterpret_cast<double*>(lds), idx, fu, hd);
    if (A->fold.ticket)
      fold_tail(A->fold.ticket, A->fold.want, A->fold.n_partials, A->fold.partials, A->fold.beta, A->fold.shape_rows,
                A->fold.beta_shape, A->fold.out66, lds);
    return;
  }
  // ---- mesh and prior roles: both start their real work once the group's frames have been handed over ----------------------
  const unsigned epoch = A->sy.epoch;
  const unsigned long long kRoleTimeoutTicks = A->sy.timeout_ticks;
  unsigned* const flag_base = A->sy.flag;
  unsigned* const error_word = A->sy.error;
  if (role == 2) grp = (idx * kPriorTileF) / kRoleGroup;
  // How a waiting workgroup learns that frames have been handed over: counters, one per 32-frame unit, each on a line of its
  // own; every launch adds exactly the unit's frame count (one agent-scope add per frame), so after launch `epoch` a complete
  // unit reads epoch x count.  A look is an sc1 load: a round trip to the memory side, where agent-scope counters live
  // (~0.8 us).  Polling is kept sparse: a few hundred waiting workgroups that hammer a line delay the very adds and stores they
  // wait for (with 1 KB of per-frame flags polled every 128 cycles the hand-off took 3 us to arrive and the producers' store
  // drain 3 us instead of 1): one immediate look (later groups: their frames are long done), then nothing before 3 us after
  // entry (no frame workgroup is faster), then one look at a time with a short sleep, ~1.1 us per look.  Measured and
  // rejected: three looks in flight ~0.27 us apart (the step was no shorter, the hand-off itself came later: the producers'
  // adds queue behind the looks); other sleeps between looks (2, 20: no difference).  The polling wave must have no operand
  // stream of its own in flight: loads return in issue order, and a look issued behind LDS-DMA pieces came back 3 us late.
  const unsigned max_polls = (unsigned)min(kRoleTimeoutTicks / 110ull, 0x7fffffffull);
  auto wait_flags = [&]() {   // prior role: all eight units of the group, polled by wave 7 (lane u: counter u)
    const int tid = threadIdx.x, lane = tid & 63;
    if ((tid >> 6) == kPollWave) {
      const int fbeg = grp * kRoleGroup, nf = min(kRoleGroup, F - fbeg);
      const int nu = (nf + kFTile - 1) / kFTile;
      const unsigned want = epoch * (unsigned)min(kFTile, nf - min(lane, 7) * kFTile);
      const unsigned* ctr = flag_base + (size_t)(grp * (kRoleGroup / kFTile) + min(lane, 7)) * kUnitCounterStride;
      const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime();
      for (unsigned polls = 0;; ++polls) {
        const unsigned got = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(lane >= nu || got == want)) break;
        if (polls >= max_polls) break;     // (nothing here depends on the frames: go on)
        if (__builtin_amdgcn_s_memrealtime() - t_enter < 300) {
          for (int i = 0; i < 6 && __builtin_amdgcn_s_memrealtime() - t_enter < 300; ++i) __builtin_amdgcn_s_sleep(20);
        } else {
          __builtin_amdgcn_s_sleep(9);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  if (role == 2) {
    // The prior residuals depend on nothing the frame workgroups produce; they wait all the same: a prior workgroup is eight
    // waves of f64 matrix work, and the frame workgroup it shares a CU with reached its hand-off 5 us late beside it (15 us
    // against 9.6), which every mesh workgroup then waited for.  Behind the hand-off it only meets a Jacobian sweep.
    const PriorArgs pa = A->pa;
    if (idx >= pa.n_tiles) return;
    wait_flags();
    prior_block(pa, idx, A->params, reinterpret_cast<double*>(lds));
    if (A->fold.ticket)
      fold_tail(A->fold.ticket, A->fold.want, A->fold.n_partials, A->fold.partials, A->fold.beta, A->fold.shape_rows,
                A->fold.beta_shape, A->fold.out66, lds);
    return;
  }
  const DevModel M = A->M;
  const DevProblem Pb = A->Pb;
  const MeshCoef mc = A->mc;
  // mesh role: one wave, one unit — the wave's lanes all look at its unit's counter (one address: a broadcast load) and the wave
  // starts its blend as soon as ITS 32 frames are in (mesh_role_inl.h)
  const unsigned long long t_role = __builtin_amdgcn_s_memrealtime();
  const int my_unit = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned want = epoch * (unsigned)max(0, min(kFTile, F - (grp * kRoleGroup + my_unit * kFTile)));
  const unsigned* ctr = flag_base + (size_t)(grp * (kRoleGroup / kFTile) + my_unit) * kUnitCounterStride;
  auto wait_unit = [&](const unsigned* c) -> bool {
    for (unsigned polls = 0;; ++polls) {
      const unsigned got = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got == want) return true;
      if (polls >= max_polls) return false;
      if (__builtin_amdgcn_s_memrealtime() - t_role < 300) {
        for (int i = 0; i < 6 && __builtin_amdgcn_s_memrealtime() - t_role < 300; ++i) __builtin_amdgcn_s_sleep(20);
      } else {
        __builtin_amdgcn_s_sleep(9);   // (2 and 20 measured: no difference)
      }
    }
  };
  auto fail = [&]() {
    if (threadIdx.x == 0) __hip_atomic_store(error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  mesh_role(M, Pb, mc, A->cloud_f, idx, grp, lds, (int)blockIdx.x < A->sy.resident_blocks, ctr, want, wait_unit, fail,
            A->sy.mesh_prio_early, A->sy.trickle_start, A->sy.trickle_sleep);
#endif
}

}  // namespace

void launch_frame_resjac(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta,
                         double* d_r, double* d_J, double* d_joints, const MeshCoef& mc, int want_jac,
                         const PriorArgs& priors, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (P.F <= 0) return;
  const size_t lds = frame_lds_bytes(M.nL);
  static DeviceMax lds_granted;
  lds_granted.raise(current_device(), lds, 48 * 1024, [&](size_t want) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_resjac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)want);
  });
  BODYFIT_LAUNCH_EXT(k_frame_resjac, dim3(P.F + priors.n_tiles), dim3(kThreads), lds, s, ev_start, ev_stop, 0, M, P,
                        d_params, d_beta, d_r, d_J, d_joints, mc, want_jac, priors);
}

void launch_mesh(const DevModel& M, const DevProblem& P, const MeshCoef& mc, float* d_cloud, const PriorArgs& pa,
                 const double* d_params, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (P.F <= 0) return;
  if ((size_t)P.nFTiles * kFTile * M.nVTiles * kVTile * 12 >= ((size_t)1 << 32)) return;   // refused at problem creation
  static DeviceOnce attr;
  attr.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mesh_blend_lbs),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  });
  BODYFIT_LAUNCH_EXT(k_mesh_blend_lbs, dim3(M.nVTiles + pa.n_tiles), dim3(64 * kWaves), kLdsBytes, s, ev_start, ev_stop,
                        0, M, P, mc, d_cloud, pa, d_params);
}

// Whether one launch can carry the sweep: every role within 80 KiB of LDS (the frame role's carve grows with the
// model's landmark slots), the flag array long enough, the cloud addressable by a raw buffer.
bool role_sweep_fits(const DevModel& M, const DevProblem& P) {
  return P.F > 0 && P.F <= kRoleMaxFrames && M.nVTiles > 0 && frame_lds_bytes(M.nL) <= (size_t)kRoleLdsBytes &&
         (size_t)P.nFTiles * kFTile * M.nVTiles * kVTile * 12 < ((size_t)1 << 32);
}

void launch_sweep_roles(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta, double* d_r,
                        double* d_J, double* d_joints, const MeshCoef& mc, int want_jac, const PriorArgs& pa, float* d_cloud,
                        const FusedSync& sy, const FoldTail& fold, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  static DeviceOnce attr;
  attr.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_roles), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kRoleLdsBytes);
  });
  const int nG = (P.F + kRoleGroup - 1) / kRoleGroup;
  const int grid = P.F + nG * M.nVTiles + pa.n_tiles;
  RoleArgs A;
  A.M = M; A.Pb = P; A.Pb.feat_perm = 1; A.params = d_params; A.beta = d_beta; A.r_out = d_r; A.J_out = d_J;
  A.joints_out = d_joints; A.mc = mc; A.want_jac = want_jac; A.pa = pa; A.cloud_f = d_cloud; A.sy = sy; A.fold = fold;
  const int dims = frame_head_dims(M.nJ, M.nS, M.nL, P.ncols, P.use_shape, P.beta_stride, mc.featA != nullptr);
  BODYFIT_LAUNCH_EXT(k_sweep_roles, dim3(grid), dim3(kThreads), kRoleLdsBytes, s, ev_start, ev_stop, 0, P.F, M.nVTiles, P.K, dims,
                     M.tabA, P.ptab, P.R0, d_params, d_beta, A);
}

}  // namespace bodyfit
