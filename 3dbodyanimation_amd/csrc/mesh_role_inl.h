// mesh_role_inl.h — the mesh role of the one-launch sweep (k_sweep_roles, k_sweep.hip): batched SMPL forward of one
// 32-vertex tile x one group of 256 frames per workgroup, in <= 128 VGPRs and 78 KiB of LDS, so that a mesh workgroup and a
// frame workgroup (frame_part_inl.h: a latency chain that issues on a few percent of its slots) share a CU and run at
// the same time.  Replaces ark::Avatar::update()'s cloud (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173;
// src/main_single_frame.cpp:254), same arithmetic as mesh_part_inl.h.
//
// Per wave: one unit of 32 frames (MFMA M) x the tile's 32 vertices (MFMA N).
//   blend     D[frame][vertex] per coordinate = [pose feature | beta | 1] . [posedirs | shapedirs - S_root | template],
//             14 k-steps of v_mfma_f32_32x32x16_bf16, operands split hi + lo in bf16, three products per k-step.
//             The tile's B operands are NOT resident: one k-step (6 KiB) at a time streams HBM/L3 -> LDS by LDS-DMA through a
//             ring of four slabs shared by the eight waves (every wave issues one 1-KiB piece per slab), three slabs
//             ahead of the one being multiplied; ONE workgroup barrier per k-step publishes slab s + 1 and frees slab s.
//             The first four slabs are requested before the wait for the frame workgroups' hand-off, so the operand stream is
//             already flowing when the blend coefficients arrive.
//   skinning  per accumulator row (two consecutive frames x 32 vertices): gather the vertex's <= 4 joint transforms from
//             the wave's own ring of three rows in LDS (2,304 bytes per row, filled by LDS-DMA straight from the frame
//             workgroups' hand-off buffer: no staging registers), blend, apply, one 12-byte write-through store per lane.
// Every vector-memory operation of the two phases is issued in fixed per-step sets, so each "has my DMA landed" wait is
// a counted s_waitcnt vmcnt(N) with N known at compile time (vmcnt retires in order).
// Hand-off (cdna guide, Guideline 16 R1): the frame workgroups store blend coefficients and transforms write-through
// (sc1) and publish a per-frame flag; here wave 0 polls the group's flags, a workgroup barrier follows, and every
// load of the handed-off bytes is an sc1 load (global_load ... sc1 to registers, global_load_lds ... sc1 to LDS).
#pragma once
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "mesh_part_inl.h"

namespace bodyfit {
namespace {

constexpr int kRoleGroup = 256;                                   // frames per mesh workgroup (8 waves x 32)
constexpr int kRingSlabs = 4;
constexpr int kSlabBytes = 3 * 2 * 1024;                          // one k-step of B: [coord][hi/lo][64 lanes x 16 B]
constexpr int kBRingBytes = kRingSlabs * kSlabBytes;              // 24,576
constexpr int kTRowBytes = 2 * kRowBytes;                         // two frames' transforms: 2,304
constexpr int kTRing = 3;
constexpr int kTWaveBytes = kTRing * kTRowBytes;                  // 6,912
constexpr int kRoleCtrlOff = kBRingBytes + kWaves * kTWaveBytes;  // 79,872: control words of the flag wait
constexpr int kRoleLdsBytes = kRoleCtrlOff + 16;
constexpr int kLoadSc1 = 16;                                      // cache policy bit sc1 of loads / LDS-DMA
static_assert(kRoleLdsBytes <= 80 * 1024, "two workgroups per CU");

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef const __attribute__((address_space(3))) f32x4 lds_f32x4;

#ifdef BODYFIT_STAMPS
// diagnostic build: per-wave s_memrealtime stamps of the mesh role (tools/stamp_roles.py)
#define RSTAMP(i)                                                                                              \
  do {                                                                                                         \
    if (Pb.dbg && C.lane == 0) {                                                                               \
      unsigned long long t_;                                                                                   \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
      Pb.dbg[kStampBase + ((size_t)(group * M.nVTiles + vtile) * 8 + C.wave) * 16 + (i)] = t_;                 \
    }                                                                                                          \
  } while (0)
#else
#define RSTAMP(i)
#endif

#define ROLE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// vector-memory operations a wave issues in blend step t: its piece of slab t + 4, the A fragments (hi, lo) of k-step t + 3
__host__ __device__ constexpr int role_blend_ops(int t, bool active) {
  return (t < 0 || t >= kBlendKSteps) ? 0 : ((t + 4 < kBlendKSteps ? 1 : 0) + ((active && t + 3 < kBlendKSteps) ? 2 : 0));
}
// ... and in skinning row t: the row's store, the three DMA pieces of row t + 3
__host__ __device__ constexpr int role_skin_ops(int t) { return (t < 0 || t >= 16) ? 0 : (1 + (t + 3 < 16 ? 3 : 0)); }

struct RoleLane {                // per-lane constants of the skinning rows
  float w0, w1, w2, w3;          // skinning weights (scalars, not an array: hipcc kept an array member in scratch)
  unsigned tj0, tj1, tj2, tj3;   // LDS byte address of joint i's transform in ring slot 0, frame h
  unsigned out_off;              // byte offset of (frame h, vertex v) in the cloud
};

struct RoleCtx {
  lds_u8* ring;                  // B ring
  lds_u8* trow;                  // this wave's transform ring
  __amdgpu_buffer_rsrc_t dirs_rsrc, skin_rsrc;   // the model's operand blocks; the hand-off buffer of transforms
  unsigned dirs_soff, skin_soff;                 // byte offsets of this tile's block / of the unit's first frame
  int piece, slab_voff;                          // this wave's piece of every slab; piece * 1024 + lane * 16
  int lane, wave;
};

// LDS-DMA in its MUBUF form (buffer_load ... lds).  The FLAT-encoded global_load_lds makes hipcc treat every later
// dependency wait as "a flat operation is pending": it then emits s_waitcnt vmcnt(0) lgkmcnt(0) in front of the first use of
// ANY loaded register, which would drain the operand stream once per k-step (seen in the .s).
// this wave's 1 KiB piece of slab s.  A slab is six pieces and the workgroup has eight waves: waves 6 and 7 request pieces 0
// and 1 a second time (same bytes to the same place; an L2 hit), so that EVERY wave issues exactly one operation per slab
// and the counted waits are the same code for all of them.  (A 12-byte-per-lane DMA would split a slab evenly, but the
// hardware places lane l's 12 bytes at 16 l: tools/ubench/dma_layout.hip.)
__device__ __forceinline__ void role_dma_slab(const RoleCtx& C, int s) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.dirs_rsrc, (__attribute__((address_space(3))) void*)(C.ring + (s % kRingSlabs) * kSlabBytes + C.piece * 1024),
                                       16, C.slab_voff, C.dirs_soff + (unsigned)s * kSlabBytes, 0, 0);
}
// the transforms of accumulator row r (two consecutive frames, 2,304 bytes) into ring slot r % 3
__device__ __forceinline__ void role_dma_row(const RoleCtx& C, int r) {
  const unsigned so = C.skin_soff + (unsigned)r * kTRowBytes;
  lds_u8* l = C.trow + (r % kTRing) * kTRowBytes;
  // (the instruction's immediate offset would be added to the LDS address as well as to the memory address: keep it 0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)l, 16, C.lane * 16, so, 0, kLoadSc1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)(l + 1024), 16, C.lane * 16, so + 1024, 0, kLoadSc1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)(l + 2048), 4, C.lane * 4, so + 2048, 0, kLoadSc1);
}

// ---- blend: k-step S ------------------------------------------------------------------------------------------------
template <int S, bool kActive>
__device__ __forceinline__ void role_blend_step(const RoleCtx& C, const __amdgpu_buffer_rsrc_t& feat_rsrc, unsigned feat_off,
                                                f32x16 (&acc)[3], u32x4 (&a)[3][2], u32x4 (&bq)[3][2]) {
  // own piece of slab S + 1 landed (with it, in order, the A fragments of this k-step): everything but the sets of the last
  // two steps (for the first three steps: everything the prologue issued, which ended with vmcnt(0))
  constexpr int kYoung = (S >= 3) ? role_blend_ops(S - 2, kActive) + role_blend_ops(S - 1, kActive)
                                  : ((S >= 1 ? role_blend_ops(0, kActive) : 0) + (S >= 2 ? role_blend_ops(1, kActive) : 0));
  __builtin_amdgcn_sched_barrier(0);   // (MFMAs are not memory operations: without this they drift across the barrier)
  if constexpr (S >= 1) ROLE_WAIT_VM(kYoung);
  // every wave has read slab S into registers (the reads were issued in step S - 1) and waited for its piece of slab S + 1
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (S + 4 < kBlendKSteps) role_dma_slab(C, S + 4);   // into the slot slab S has just left
  if constexpr (kActive) {
    const bf16x8 a_hi = __builtin_bit_cast(bf16x8, a[S % 3][0]);
    const bf16x8 a_lo = __builtin_bit_cast(bf16x8, a[S % 3][1]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const bf16x8 bhi = __builtin_bit_cast(bf16x8, bq[c][0]);
      const bf16x8 blo = __builtin_bit_cast(bf16x8, bq[c][1]);
      if constexpr (S == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, z, 0, 0, 0);
      } else {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, acc[c], 0, 0, 0);
      }
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, blo, acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bhi, acc[c], 0, 0, 0);
      if constexpr (S + 1 < kBlendKSteps) {   // the next k-step's fragments of this coordinate, under the other coordinates' products
        const lds_u8* sl = C.ring + ((S + 1) % kRingSlabs) * kSlabBytes + (c * 2) * 1024 + C.lane * 16;
        bq[c][0] = *reinterpret_cast<lds_u32x4*>(sl);
        bq[c][1] = *reinterpret_cast<lds_u32x4*>(sl + 1024);
      }
    }
    if constexpr (S + 3 < kBlendKSteps) {     // A fragments three k-steps ahead (L2; handed over in this launch: sc1)
      const unsigned soff = feat_off + (unsigned)((S + 3) * 2 * 1024);
      a[S % 3][0] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (unsigned)(C.lane * 16), soff, kLoadSc1);
      a[S % 3][1] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (unsigned)(C.lane * 16 + 1024), soff, kLoadSc1);
    }
  }
  if constexpr (kActive) {
    // issue order of the step: each coordinate's three products, then the next k-step's two fragment reads of that
    // coordinate (a whole k-step of matrix work ahead of their use), the A fragments last
    if constexpr (S + 4 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (the slab piece)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                     // MFMA
      if constexpr (S + 1 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); // DS read
    }
    if constexpr (S + 3 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");   // the step's vector-memory set ends here (the counted waits rely on it)
}

// ---- skinning: accumulator row R (frames 2 R + h of the unit) ---------------------------------------------------------
template <int R>
__device__ __forceinline__ void role_skin_row(const RoleCtx& C, const RoleLane& L, const f32x16 (&acc)[3],
                                              __amdgpu_buffer_rsrc_t cloud, unsigned row_off) {
  // the row's transforms have landed: everything but the sets of the last two rows (rows 0-2 were requested before the blend)
  if constexpr (R >= 3) ROLE_WAIT_VM(role_skin_ops(R - 2) + role_skin_ops(R - 1));
  constexpr unsigned slot = (R % kTRing) * kTRowBytes;
  f32x4 t[12];
  const unsigned tj[4] = {L.tj0, L.tj1, L.tj2, L.tj3};
  const float wgt[4] = {L.w0, L.w1, L.w2, L.w3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_f32x4* T = (lds_f32x4*)(tj[i] + slot);
    t[3 * i + 0] = T[0]; t[3 * i + 1] = T[1]; t[3 * i + 2] = T[2];
  }
  // (x, y) and (z, w) of a transform row are natural register pairs: packed f32 without shuffles
  f32x2 b[6];
  const f32x2 w0 = {wgt[0], wgt[0]};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    b[2 * k] = w0 * f32x2{t[k].x, t[k].y};
    b[2 * k + 1] = w0 * f32x2{t[k].z, t[k].w};
  }
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const f32x2 wi = {wgt[i], wgt[i]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      b[2 * k] += wi * f32x2{t[3 * i + k].x, t[3 * i + k].y};
      b[2 * k + 1] += wi * f32x2{t[3 * i + k].z, t[3 * i + k].w};
    }
  }
  const f32x2 pxy = {acc[0][R], acc[1][R]};       // the rest vertex (the template rides in the contraction)
  const f32x2 pz1 = {acc[2][R], 1.0f};
  typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;
  u32x3 pk;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x2 m = b[2 * k] * pxy + b[2 * k + 1] * pz1;
    pk[k] = __float_as_uint(m.x + m.y);
  }
  __builtin_amdgcn_raw_buffer_store_b96(pk, cloud, L.out_off, row_off, kStoreAux);
  if constexpr (R + 3 < 16) role_dma_row(C, R + 3);   // into the slot this row has just been read from
  asm volatile("" ::: "memory");
}

// One workgroup: vertex tile `vtile`, frames [256 group, 256 group + 256).  flags_ready: the caller has waited for the
// group's hand-off flags (or the operands are from an earlier launch).  `lds`: kRoleLdsBytes.
template <typename WaitFlags>
__device__ __forceinline__ void mesh_role(const DevModel& M, const DevProblem& Pb, const MeshCoef& mc, float* __restrict__ cloud_f,
                                          int vtile, int group, unsigned char* lds_generic, WaitFlags wait_flags) {
  RoleCtx C;
  C.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  C.lane = threadIdx.x & 63;
  lds_u8* lds = (lds_u8*)lds_generic;
  C.ring = lds;
  C.trow = lds + kBRingBytes + C.wave * kTWaveBytes;
  C.dirs_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(M.dirsB), 0, M.nVTiles * kBBytes, 0x00020000);
  C.dirs_soff = (unsigned)vtile * kBBytes;
  const int nFT = Pb.nFTiles;
  const int ftile = group * (kRoleGroup / kFTile) + C.wave;          // this wave's unit
  const bool active = ftile < nFT;
  const int f0 = ftile * kFTile;
  C.skin_rsrc = __builtin_amdgcn_make_buffer_rsrc(mc.skinT, 0, nFT * kFTile * kRowBytes, 0x00020000);
  C.skin_soff = (unsigned)(active ? f0 : 0) * kRowBytes;
  C.piece = C.wave < 6 ? C.wave : C.wave - 6;
  C.slab_voff = C.piece * 1024 + C.lane * 16;
  const int col = C.lane & 31, h = C.lane >> 5;
  const int v = vtile * kVTile + col;

  RSTAMP(0);
  // ---- independent of the frame workgroups: the first four slabs of the operand stream, the lane's skinning weights -----
#pragma unroll
  for (int s = 0; s < kRingSlabs; ++s) role_dma_slab(C, s);
  const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
  const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
  RoleLane L;
  L.w0 = wv.x; L.w1 = wv.y; L.w2 = wv.z; L.w3 = wv.w;
  const unsigned trow_addr = (unsigned)(size_t)C.trow + (unsigned)h * kRowBytes;   // LDS byte address
  L.tj0 = trow_addr + (widx & 0xffu) * 48u;
  L.tj1 = trow_addr + ((widx >> 8) & 0xffu) * 48u;
  L.tj2 = trow_addr + ((widx >> 16) & 0xffu) * 48u;
  L.tj3 = trow_addr + (widx >> 24) * 48u;
  const unsigned stride = (unsigned)M.nVTiles * kVTile * 12;          // bytes per frame of the cloud
  L.out_off = (unsigned)h * stride + (unsigned)v * 12;
  const __amdgpu_buffer_rsrc_t cloud =
      __builtin_amdgcn_make_buffer_rsrc(cloud_f, 0, (int)((unsigned)nFT * kFTile * stride), 0x00020000);
  const __amdgpu_buffer_rsrc_t feat_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(mc.featA, 0, nFT * kBlendKSteps * 2 * 1024, 0x00020000);
  const unsigned feat_off = (unsigned)((active ? ftile : 0) * kBlendKSteps * 2 * 1024);

  // ---- the group's frames have been handed over -----------------------------------------------------------------------
  if (!wait_flags()) return;   // (workgroup-uniform; includes the barrier that orders the poll before every operand load)
  RSTAMP(1);

  f32x16 acc[3];
  u32x4 a[3][2], bq[3][2];
  if (active) {
#pragma unroll
    for (int r = 0; r < kTRing; ++r) role_dma_row(C, r);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const unsigned soff = feat_off + (unsigned)(ks * 2 * 1024);
      a[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (unsigned)(C.lane * 16), soff, kLoadSc1);
      a[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (unsigned)(C.lane * 16 + 1024), soff, kLoadSc1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slabs 0-3, rows 0-2, A fragments 0-2
  RSTAMP(2);
  if (active) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const lds_u8* sl = C.ring + (c * 2) * 1024 + C.lane * 16;
      bq[c][0] = *reinterpret_cast<lds_u32x4*>(sl);
      bq[c][1] = *reinterpret_cast<lds_u32x4*>(sl + 1024);
    }
#define RB(S) role_blend_step<S, true>(C, feat_rsrc, feat_off, acc, a, bq)
    RB(0); RB(1); RB(2); RB(3); RB(4); RB(5); RB(6); RB(7); RB(8); RB(9); RB(10); RB(11); RB(12); RB(13);
#undef RB
    // (the last steps issued nothing: every DMA of the blend has been waited for, rows 0-2 landed long ago)
    RSTAMP(3);
    const unsigned row0 = (unsigned)f0 * stride;
#define RS(R) role_skin_row<R>(C, L, acc, cloud, row0 + (unsigned)(2 * (R)) * stride)
    RS(0); RS(1); RS(2); RS(3); RS(4); RS(5); RS(6); RS(7); RS(8); RS(9); RS(10); RS(11); RS(12); RS(13); RS(14); RS(15);
#undef RS
    RSTAMP(4);
  } else {
    // a wave without frames (last group of a frame count that is not a multiple of 256) keeps the operand stream and the
    // barriers of the others going
#define RB(S) role_blend_step<S, false>(C, feat_rsrc, feat_off, acc, a, bq)
    RB(0); RB(1); RB(2); RB(3); RB(4); RB(5); RB(6); RB(7); RB(8); RB(9); RB(10); RB(11); RB(12); RB(13);
#undef RB
  }
}

}  // namespace
}  // namespace bodyfit
