// mesh_role_inl.h — the mesh role of the one-launch sweep (k_sweep_roles, k_sweep.hip): batched SMPL forward of one
// 32-vertex tile x one group of 256 frames per workgroup, in <= 128 VGPRs and 78 KiB of LDS, so that a mesh workgroup and a
// frame workgroup (frame_part_inl.h: a latency chain that issues on a few percent of its slots) share a CU and run at
// the same time.  Replaces ark::Avatar::update()'s cloud (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173;
// src/main_single_frame.cpp:254), same arithmetic as mesh_part_inl.h.
//
// Per wave: one unit of 32 frames (MFMA M) x the tile's 32 vertices (MFMA N).
//   blend     D[frame][vertex] per coordinate = [pose feature | beta | 1] . [posedirs | shapedirs - S_root | template],
//             14 k-steps of v_mfma_f32_32x32x16_bf16, operands split hi + lo in bf16, three products per k-step.
//             Thirteen of the tile's fourteen operand slabs (6 KiB per k-step) are RESIDENT in LDS: they are requested by
//             LDS-DMA at the very start, before the wait for the frame workgroups' hand-off (the operands do not depend on
//             the pose; as a trickle, so that the frame workgroups' own loads are not delayed), so the 78 KiB stream hides under that wait; the fourteenth slab goes L2 -> registers one
//             k-step ahead of its use.  The blend itself has no barrier and no operand traffic per k-step.
//   skinning  per accumulator row (two consecutive frames x 32 vertices): gather the vertex's <= 4 joint transforms from
//             the wave's own ring of four rows in LDS (2,304 bytes per row, filled by LDS-DMA straight from the frame
//             workgroups' hand-off buffer: no staging registers), blend, apply, one 12-byte write-through store per lane;
//   LDS       78 KiB, used twice: the slabs of k-steps 0-8 (region T) are dead after k-step 8, one workgroup barrier
//             later the same bytes are the eight waves' transform rings; the slabs of k-steps 9-12 (region A) give the
//             rings their fourth slot after the blend.  Two workgroup barriers in all.
// Every vector-memory operation of the skinning phase is issued in fixed per-row sets, so each "has my DMA landed" wait is
// a counted s_waitcnt vmcnt(N) with N known at compile time (vmcnt retires in order).
// Hand-off (cdna guide, Guideline 16 R1): the frame workgroups store blend coefficients and transforms write-through
// (sc1) and publish them per 32-frame unit with two counters (coefficients first, transforms ~1 us later).  Here each wave
// waits for the coefficient counter of ITS unit and starts its blend at once; the transform counter is looked at four
// k-steps ahead of the barrier behind which the first transform rows are requested.  Every load of the handed-off bytes is
// an sc1 load (buffer_load ... sc1 to registers, buffer_load ... lds sc1 to LDS).
#pragma once
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "mesh_part_inl.h"

namespace bodyfit {
namespace {

constexpr int kRoleGroup = 256;                                   // frames per mesh workgroup (8 waves x 32)
constexpr int kSlabBytes = 3 * 2 * 1024;                          // one k-step of B: [coord][hi/lo][64 lanes x 16 B]
constexpr int kResident = 13;                                     // k-steps of B resident in LDS (the 14th goes through registers)
constexpr int kRegionASlabs = 4;                                  // region A: slabs 9..12; after the blend: ring slot 3 of every wave
constexpr int kRegionABytes = kRegionASlabs * kSlabBytes;         // 24,576
constexpr int kRegionTSlabs = kResident - kRegionASlabs;          // region T: slabs 0..8; from k-step 9 on: the waves' transform rings
constexpr int kTRowBytes = 2 * kRowBytes;                         // two frames' transforms: 2,304
constexpr int kTRing = 4;                                         // rows per wave: slots 0-2 in region T, slot 3 in region A
constexpr int kTWaveBytes = 3 * kTRowBytes;                       // 6,912 of region T per wave
constexpr int kTSlot3Bytes = kRegionABytes / kWaves;              // 3,072 of region A per wave
constexpr int kRoleCtrlOff = kRegionABytes + kWaves * kTWaveBytes;   // 79,872: control words of the flag wait
constexpr int kRoleLdsBytes = kRoleCtrlOff + 16;
// How the mesh role reads the operands handed over inside the launch: every load of them (buffer loads, LDS-DMA) carries the
// cache policy bit sc1.  (Measured and rejected: one agent-scope acquire — buffer_inv sc1 — behind the poll and plain loads
// after it: no difference.)
constexpr int kLoadSc1 = 16;
static_assert(kRegionTSlabs * kSlabBytes == kWaves * kTWaveBytes, "the transform rings take over region T exactly");
static_assert(kTSlot3Bytes >= kTRowBytes, "ring slot 3");
static_assert(kRoleLdsBytes <= 80 * 1024, "two workgroups per CU");
// LDS byte offset of resident slab s
__host__ __device__ constexpr int role_slab_off(int s) {
  return s < kRegionTSlabs ? kRegionABytes + s * kSlabBytes : (s - kRegionTSlabs) * kSlabBytes;
}

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
// shader-cycle stamps of the blend's k-steps (slots 0-14) and the skinning rows (16-32), one 40-slot record per wave
#define RCYC(i)                                                                                                \
  do {                                                                                                         \
    if (Pb.dbg && C.lane == 0) {                                                                               \
      unsigned long long t_;                                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
      Pb.dbg[kStampBase + ((size_t)1 << 19) + ((size_t)(group * M.nVTiles + vtile) * 8 + C.wave) * 40 + (i)] = t_; \
    }                                                                                                          \
  } while (0)
#else
#define RSTAMP(i)
#define RCYC(i)
#endif

#define ROLE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// vector-memory operations a wave issues in skinning row t: the row's store, the three DMA pieces of row t + 4
__host__ __device__ constexpr int role_skin_ops(int t) { return (t < 0 || t >= 16) ? 0 : (1 + (t + kTRing < 16 ? 3 : 0)); }

struct RoleLane {                // per-lane constants of the skinning rows
  float w0, w1, w2, w3;          // skinning weights (scalars, not an array: hipcc kept an array member in scratch)
  unsigned tj0, tj1, tj2, tj3;   // LDS byte address of joint i's transform in ring slot 0, frame h
  unsigned slot3;                // what to add for ring slot 3 (region A; wave-uniform)
  unsigned out_off;              // byte offset of (frame h, vertex v) in the cloud
};

struct RoleCtx {
  lds_u8* ring;                  // LDS base (region A, then region T)
  unsigned ring_addr;            // ... as an integer + lane * 16: the B-fragment reads take their address from it (an LDS read
                                 // through a pointer hipcc can trace is made to wait for every LDS-DMA in flight, i.e. for region
                                 // A while k-steps 0-8 read region T)
  lds_u8* trow;                  // this wave's transform ring slots 0-2 (region T)
  lds_u8* trow3;                 // ... and slot 3 (region A)
  __amdgpu_buffer_rsrc_t dirs_rsrc, skin_rsrc;   // the model's operand blocks; the hand-off buffer of transforms
  unsigned dirs_soff, skin_soff;                 // byte offsets of this tile's block / of the unit's first frame
  int piece, slab_voff;                          // this wave's piece of every slab; piece * 1024 + lane * 16
  int lane, wave;
};

// LDS-DMA in its MUBUF form (buffer_load ... lds).  The FLAT-encoded global_load_lds makes hipcc treat every later
// dependency wait as "a flat operation is pending": it then emits s_waitcnt vmcnt(0) lgkmcnt(0) in front of the first use of
// ANY loaded register, which would drain the operand stream once per k-step (seen in the .s).
// this wave's 1 KiB piece of resident slab s.  A slab is six pieces and the workgroup has eight waves: waves 6 and 7 request
// pieces 0 and 1 a second time (same bytes to the same place; an L2 hit), so that every wave issues the same operations.
// (A 12-byte-per-lane DMA would split a slab evenly, but the hardware places lane l's 12 bytes at 16 l:
// tools/ubench/dma_layout.hip.)
__device__ __forceinline__ void role_dma_slab(const RoleCtx& C, int s) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.dirs_rsrc, (__attribute__((address_space(3))) void*)(C.ring + role_slab_off(s) + C.piece * 1024),
                                       16, C.slab_voff, C.dirs_soff + (unsigned)s * kSlabBytes, 0, 0);
}
// the transforms of accumulator row r (two consecutive frames, 2,304 bytes) into ring slot r % 4
__device__ __forceinline__ void role_dma_row(const RoleCtx& C, int r) {
  const unsigned so = C.skin_soff + (unsigned)r * kTRowBytes;
  lds_u8* l = (r % kTRing) == 3 ? C.trow3 : C.trow + (r % kTRing) * kTRowBytes;
  // (the instruction's immediate offset would be added to the LDS address as well as to the memory address: keep it 0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)l, 16, C.lane * 16, so, 0, kLoadSc1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)(l + 1024), 16, C.lane * 16, so + 1024, 0, kLoadSc1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.skin_rsrc, (__attribute__((address_space(3))) void*)(l + 2048), 4, C.lane * 4, so + 2048, 0, kLoadSc1);
}

// ---- blend: k-step S ------------------------------------------------------------------------------------------------
// No barrier and no operand traffic per k-step: slabs 0-12 are resident (requested before the wait for the hand-off), the
// waves run at their own pace and the two waves of a SIMD drift into opposite phases (one multiplies while the other reads).
template <int S>
__device__ __forceinline__ void role_blend_step(const RoleCtx& C, const __amdgpu_buffer_rsrc_t& feat_rsrc, unsigned feat_off,
                                                f32x16 (&acc)[3], u32x4 (&a)[3][2], u32x4 (&bq)[3][2]) {
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
    if constexpr (S + 1 < kResident && S + 1 != kRegionTSlabs) {   // the next k-step's fragments of this coordinate, under the other
                                                                   // coordinates' products (slab 9: behind the barrier, mesh_role)
      const unsigned sl = C.ring_addr + (unsigned)(role_slab_off(S + 1) + (c * 2) * 1024);
      bq[c][0] = *(lds_u32x4*)(size_t)(sl);
      bq[c][1] = *(lds_u32x4*)(size_t)(sl + 1024u);
    } else if constexpr (S + 1 >= kResident && S + 1 < kBlendKSteps) {   // the 14th k-step is not resident: L2 -> registers, a k-step of matrix work ahead
      const unsigned so = C.dirs_soff + (unsigned)((S + 1) * kSlabBytes + (c * 2) * 1024);
      bq[c][0] = __builtin_amdgcn_raw_buffer_load_b128(C.dirs_rsrc, (unsigned)(C.lane * 16), so, 0);
      bq[c][1] = __builtin_amdgcn_raw_buffer_load_b128(C.dirs_rsrc, (unsigned)(C.lane * 16), so + 1024, 0);
    }
  }
  if constexpr (S + 3 < kBlendKSteps) {     // A fragments three k-steps ahead (L2; handed over in this launch: sc1)
    const unsigned soff = feat_off + (unsigned)((S + 3) * 2 * 1024);
    a[S % 3][0] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 0), soff, kLoadSc1);
    a[S % 3][1] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 1), soff, kLoadSc1);
  }
  // issue order of the step: each coordinate's three products, then the next k-step's two fragment reads of that
  // coordinate (a whole k-step of matrix work ahead of their use), the A fragments last
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                        // MFMA
    if constexpr (S + 1 < kResident && S + 1 != kRegionTSlabs) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read
    else if constexpr (S + 1 >= kResident && S + 1 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read
  }
  if constexpr (S + 3 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // VMEM read
  __builtin_amdgcn_sched_barrier(0);
}

// ---- skinning: accumulator row R (frames 2 R + h of the unit) ---------------------------------------------------------
// The ring has four rows: row R + 4 is requested when row R has been read, three rows of work ahead of its use.
template <int R>
__device__ __forceinline__ void role_skin_row(const RoleCtx& C, const RoleLane& L, const f32x16 (&acc)[3],
                                              __amdgpu_buffer_rsrc_t cloud, unsigned row_off) {
  // the row's transforms have landed: everything but the sets of the last three rows (rows 0-2 were requested at k-step 9,
  // row 3 at the end of the blend, in front of row 0's set)
  if constexpr (R >= 3) ROLE_WAIT_VM(role_skin_ops(R - 3) + role_skin_ops(R - 2) + role_skin_ops(R - 1));
  const unsigned slot = (R % kTRing) == 3 ? L.slot3 : (unsigned)(R % kTRing) * kTRowBytes;
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
  if constexpr (R + kTRing < 16) role_dma_row(C, R + kTRing);   // into the slot this row has been read from
  asm volatile("" ::: "memory");
}

// ---- where the blend coefficients (MFMA A fragments) come from ------------------------------------------------------------------
// The A operand of the blend is [vec(R_j - I), j = 1..23 | beta | 1 1 0..] per frame: it depends on the frame's raw parameters
// only.  In round 3 wave 0 of the frame role packed it in phase C from the f64 rotations and published it 5.2 us into the launch;
// with the hand-off across XCDs (write-through drain, agent-scope add, the consumer's sc1 look: each a trip through the fabric of
// 0.4-1 us) a mesh wave saw it at 7.2 us.  Now wave 5 of the frame role computes it in phase B, in f32, from the raw parameters
// (frame_part_inl.h): ~2 us earlier.  Measured on the way and rejected (profiles/r4_*_stamps*.txt):
//   * a dedicated coefficient role, 16 blocks in front of the launch, one copy published across XCDs: no earlier than round 3
//     (fragments in LDS at 1.9 us, but stores drained at 4.5 us behind the mesh role's operand stream, signalled 4.9, seen 7.0);
//   * the hand-off kept INSIDE an XCD's L2 (every item computed once per XCD into a copy of its own, plain stores, flag words):
//     as 128 extra blocks it pushed the launch past its 512 resident workgroups (a hundred mesh workgroups entered at 5.7 us);
//     by waves 6 and 7 of the mesh workgroups (no piece of the operand stream) it took one wave 5 us per eight frames; by four
//     waves of every frame workgroup (eight frames each, no LDS, two rotations per thread) 12 k cycles of phase B — each of them
//     1,300 instructions of selects —: 33 us per step.  What was learnt about L2-local signalling: a look at a flag must be a
//     SCALAR load with glc (straight to the L2); a vector load with sc0 (workgroup scope) may hit in the CU's L1 — the first look
//     parked the line there and every later one read that copy until the wait ran out —, with sc1 it goes past the L2.

// One workgroup: vertex tile `vtile`, frames [256 group, 256 group + 256).  flags_ready: the caller has waited for the
// group's hand-off flags (or the operands are from an earlier launch).  `lds`: kRoleLdsBytes.
// unit_ctr / unit_want: this wave's unit counters (k_sweep.hip; word 0: transforms, word kUnitCoefOffset: blend coefficients) and
// the value each shows once the unit's frames have all published; wait_unit(counter) polls one (bounded), fail() marks the launch
// as incomplete.
template <typename WaitUnit, typename Fail>
__device__ __forceinline__ void mesh_role(const DevModel& M, const DevProblem& Pb, const MeshCoef& mc, float* __restrict__ cloud_f,
                                          int vtile, int group, unsigned char* lds_generic, bool beside_its_frames,
                                          const unsigned* unit_ctr, unsigned unit_want, WaitUnit wait_unit, Fail fail,
                                          int prio_early, int trickle_start, int trickle_sleep) {
  RoleCtx C;
  C.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  C.lane = threadIdx.x & 63;
  lds_u8* lds = (lds_u8*)lds_generic;
  C.ring = lds;
  C.ring_addr = (unsigned)(size_t)lds + (unsigned)C.lane * 16u;
  C.trow = lds + kRegionABytes + C.wave * kTWaveBytes;
  C.trow3 = lds + C.wave * kTSlot3Bytes;
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
  // ---- independent of every other workgroup: the tile's operand slabs, the lane's skinning weights --------------------------
  const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
  const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
  // First the nine slabs of region T (k-steps 0-8), which is all the blend needs before its barrier in front of k-step 9; the
  // four slabs of region A follow behind the first barrier, under the first k-steps.
  // A workgroup that is resident from the start of the launch runs BESIDE the frame workgroups: its stream is a trickle, not a
  // burst — requested all at once (17 MB chip-wide) it stretched the frame workgroups' table loads by 900 cycles and their
  // hand-off of the transforms by 2 us: one slab per ~0.25 us from 1.2 us on (past the frame workgroups' table loads).  A
  // workgroup dispatched later (more frames than one group) requests everything at once.
  if (C.wave < 6) {   // six pieces per slab: waves 6 and 7 have none
    if (beside_its_frames) {
      const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t_in < (unsigned long long)trickle_start) __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll 1
    for (int s = 0; s < kRegionTSlabs; ++s) {
      role_dma_slab(C, s);
      if (beside_its_frames)
        for (int z = 0; z < trickle_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
  }
  RoleLane L;
  L.w0 = wv.x; L.w1 = wv.y; L.w2 = wv.z; L.w3 = wv.w;
  const unsigned trow_addr = (unsigned)(size_t)C.trow + (unsigned)h * kRowBytes;   // LDS byte address
  L.tj0 = trow_addr + (widx & 0xffu) * 48u;
  L.tj1 = trow_addr + ((widx >> 8) & 0xffu) * 48u;
  L.tj2 = trow_addr + ((widx >> 16) & 0xffu) * 48u;
  L.tj3 = trow_addr + (widx >> 24) * 48u;
  L.slot3 = (unsigned)(size_t)C.trow3 - (unsigned)(size_t)C.trow;    // (mod 2^32: slot 3 lies below the ring)
  const unsigned stride = (unsigned)M.nVTiles * kVTile * 12;          // bytes per frame of the cloud
  L.out_off = (unsigned)h * stride + (unsigned)v * 12;
  const __amdgpu_buffer_rsrc_t cloud =
      __builtin_amdgcn_make_buffer_rsrc(cloud_f, 0, (int)((unsigned)nFT * kFTile * stride), 0x00020000);
  const unsigned feat_off = (unsigned)((active ? ftile : 0) * kBlendKSteps * 2 * 1024);

  // every wave's pieces of slabs 0-8 have landed, the control word is clear
  // (an LDS pointer, not a generic one: a FLAT store anywhere in front of the blend makes hipcc's wait-count pass drain vmcnt
  //  and lgkmcnt to 0 at the next use of any loaded register — the first product then waits for every fragment load in flight)
  volatile __attribute__((address_space(3))) unsigned* ctrl =
      (volatile __attribute__((address_space(3))) unsigned*)(lds + kRoleCtrlOff);
  if (threadIdx.x == 0) ctrl[0] = 0u;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) in a form hipcc's wait-count pass sees: region T is not "in flight" any more
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- this wave's 32-frame unit has published its BLEND COEFFICIENTS (frame role, wave 5: ~3.5 us into the launch; the
  //      transforms follow at ~6 us and are needed in front of k-step 9 only): each wave waits for ITS unit alone (sc1 looks: a
  //      round trip to the memory side) and starts its blend at once.  A wait that runs out leaves a mark the whole workgroup
  //      acts on behind the barrier of k-step 9. ------------------------------------------------------------------------------
  if (active && !wait_unit(unit_ctr + kUnitCoefOffset)) ctrl[0] = 1u;
  RSTAMP(1);
  // from here on the mesh role is the launch's critical path
  // ... but until its transforms are in (k-step 9) it must not take issue slots from the frame workgroup beside it, which still
  // owes them (at priority 3 from here, the blend stretched the frame role's phase C from 7.3 k to 9.4 k cycles and then waited
  // 6.7 k cycles for the transforms in front of k-step 9)
  if (prio_early == 0) __builtin_amdgcn_s_setprio(0);
  else if (prio_early == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio_early == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);

  f32x16 acc[3];
  u32x4 a[3][2], bq[3][2];
  const __amdgpu_buffer_rsrc_t feat_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(mc.featA, 0, nFT * kBlendKSteps * 2 * 1024, 0x00020000);
  // One straight-line block from the first fragment load to k-step 8 (a branch with loads in flight makes hipcc's wait-count
  // pass fall back to vmcnt(0) at the join — here: the first product would wait for region A to land).  Region A (slabs
  // 9-12) is requested by EVERY wave behind its first fragment loads (loads return in order), waves 6 and 7 asking for pieces
  // 0 and 1 a second time (same bytes to the same place, an L2 hit); it lands under k-steps 0-8 and is waited for together with
  // the transform counter's look below.
  RSTAMP(2);
#define RB(S) RCYC(S); role_blend_step<S>(C, feat_rsrc, feat_off, acc, a, bq)
  // Two things happen on the way: (1) in front of k-step 9 the workgroup's waves meet (barrier): every wave has read slabs 0-8
  // (region T) into registers and every wave's pieces of region A have landed, region T becomes the waves' transform rings;
  // (2) the unit's TRANSFORMS (handed over across XCDs by the frame role's wave 7, 6-8 us into the launch, + 1-2 us until a
  // look from here sees them) are needed only by the rows' LDS-DMA, which takes ~1 us to land: one look at their counter in
  // front of k-step 7, acted on behind k-step 10 (eight fragment loads are issued behind it, so it is back at vmcnt(8)), rows
  // 0-2 requested there, under the last three k-steps.  (Round 3 and the first forms of this round waited for the transforms
  // in front of k-step 9, at the barrier: with the blend starting 1.3 us earlier than then, that is where it stood — 4.6 k
  // cycles — until the slowest of the unit's 32 frames had handed over.)
  unsigned look1 = unit_want;
  if (active) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const unsigned soff = feat_off + (unsigned)(ks * 2 * 1024);
      a[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 0), soff, kLoadSc1);
      a[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 1), soff, kLoadSc1);
    }
    __builtin_amdgcn_sched_barrier(0);   // (all six fragment loads in front of the slab requests: loads return in order)
#pragma unroll
    for (int s = kRegionTSlabs; s < kResident; ++s) role_dma_slab(C, s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned sl = C.ring_addr + (unsigned)(role_slab_off(0) + (c * 2) * 1024);
      bq[c][0] = *(lds_u32x4*)(size_t)(sl);
      bq[c][1] = *(lds_u32x4*)(size_t)(sl + 1024u);
    }
    RB(0); RB(1); RB(2); RB(3); RB(4); RB(5); RB(6);
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(look1) : "v"(unit_ctr) : "memory");
    RB(7); RB(8);
    // this wave's pieces of region A have landed: everything but the look and the four fragment loads behind it
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  } else {   // (a wave without a unit still carries its pieces of region A)
#pragma unroll
    for (int s = kRegionTSlabs; s < kResident; ++s) role_dma_slab(C, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (active) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // slab 9's fragments (k-step 8 could not request them: region A was still landing)
      const unsigned sl = C.ring_addr + (unsigned)(role_slab_off(kRegionTSlabs) + (c * 2) * 1024);
      bq[c][0] = *(lds_u32x4*)(size_t)(sl);
      bq[c][1] = *(lds_u32x4*)(size_t)(sl + 1024u);
    }
    RB(9); RB(10);
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(look1)::"memory");
    if (look1 != unit_want && !wait_unit(unit_ctr)) ctrl[0] = 1u;
    else { role_dma_row(C, 0); role_dma_row(C, 1); role_dma_row(C, 2); }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(3);
    RB(11); RB(12); RB(13);
  }
#undef RB
  RCYC(14);
  // every wave has read slabs 9-12 (region A): it becomes ring slot 3
  __builtin_amdgcn_sched_barrier(0);   // (MFMAs are not memory operations: without this the last ones drift into the skinning rows)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (ctrl[0] != 0u) {   // (workgroup-uniform: written before the barrier) a unit never arrived: leave, the host re-issues
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fail();
    return;
  }
  if (!active) return;
  RSTAMP(3);
  role_dma_row(C, 3);
  asm volatile("" ::: "memory");
  // rows 0-2 have landed: everything but row 3's three pieces (the blend's own loads were waited for where they were used)
  ROLE_WAIT_VM(3);
  const unsigned row0 = (unsigned)f0 * stride;
#define RS(R) RCYC(16 + (R)); role_skin_row<R>(C, L, acc, cloud, row0 + (unsigned)(2 * (R)) * stride)
  RS(0); RS(1); RS(2); RS(3); RS(4); RS(5); RS(6); RS(7); RS(8); RS(9); RS(10); RS(11); RS(12); RS(13); RS(14); RS(15);
#undef RS
  RCYC(32);
  RSTAMP(4);
}

}  // namespace
}  // namespace bodyfit
