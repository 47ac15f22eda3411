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
    if constexpr (S + 1 < kResident) {   // the next k-step's fragments of this coordinate, under the other coordinates' products
      const lds_u8* sl = C.ring + role_slab_off(S + 1) + (c * 2) * 1024 + C.lane * 16;
      bq[c][0] = *reinterpret_cast<lds_u32x4*>(sl);
      bq[c][1] = *reinterpret_cast<lds_u32x4*>(sl + 1024);
    } else if constexpr (S + 1 < kBlendKSteps) {   // the 14th k-step is not resident: L2 -> registers, a k-step of matrix work ahead
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
    if constexpr (S + 1 < kResident) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read
    else if constexpr (S + 1 < kBlendKSteps) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read
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

// ---- coefficient role: the blend coefficients (MFMA A fragments) of 16 frames --------------------------------------------
// The A operand of the blend is [vec(R_j - I), j = 1..23 | beta | 1 1 0..] per frame: it depends on the frame's raw parameters
// only.  When the frame role produced it (round 3), a mesh wave could not start its blend before 5.2 us (table round trip,
// phase B's Rodrigues + gradient, phase C's pack, drain, signal) + detection = 7.4 us into the launch.  This role computes
// nothing else: parameters -> Rodrigues in f32 (the mesh is an f32 product split in bf16 hi + lo: the f64 rotation of the frame
// role, rounded to f32, and the f32 rotation differ by ~1e-7 relative on coefficients that multiply centimetre-scale
// directions; mesh tolerance 5e-6 m) -> fragments, write-through -> one agent-scope add of its frame count to the unit's
// coefficient counter.  Its blocks are the FIRST of the launch, two per 32-frame unit.
constexpr int kCoefFrames = 16;
constexpr int kCoefLdsFloats = kCoefFrames * 16 * kBlendKSteps;   // [16 frames][224 coefficients]
#ifdef BODYFIT_STAMPS
#define CSTAMP(i)                                                                                              \
  do {                                                                                                         \
    if (Pb.dbg && threadIdx.x == 0) {                                                                          \
      unsigned long long t_;                                                                                   \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
      Pb.dbg[kStampBase - 8192 + (size_t)blk * 16 + (i)] = t_;                                                 \
    }                                                                                                          \
  } while (0)
#else
#define CSTAMP(i)
#endif

__device__ __forceinline__ void coef_role(const DevModel& M, const DevProblem& Pb, const double* __restrict__ params,
                                          const double* __restrict__ beta, const MeshCoef& mc, int blk, unsigned* flag_base,
                                          unsigned char* lds_generic) {
  constexpr int kK = 16 * kBlendKSteps;   // 224
  float* sf = reinterpret_cast<float*>(lds_generic);
  const int tid = threadIdx.x;
  const int nJ = M.nJ, nS = M.nS, npose = 7 + 3 * (nJ - 1);
  const int f0 = blk * kCoefFrames, nf = min(kCoefFrames, Pb.F - f0);
  const int nrot = kCoefFrames * (nJ - 1);                       // (frame, joint) items: 368 <= 512, one pass
  CSTAMP(0);
  // every load first (one round trip)
  const int fr = min(tid / (nJ - 1), kCoefFrames - 1), j = tid - (tid / (nJ - 1)) * (nJ - 1);
  const double* aa = params + (size_t)(f0 + min(fr, nf - 1)) * npose + 7 + 3 * j;
  const double a0d = aa[0], a1d = aa[1], a2d = aa[2];
  const int bfr = min(tid / kMaxShape, kCoefFrames - 1), bk = tid % kMaxShape;
  double bv = 0.0;
  if (Pb.use_shape && beta && bk < nS) bv = beta[(size_t)(f0 + min(bfr, nf - 1)) * Pb.beta_stride + bk];
  if (tid < nrot) {
    const float a0 = (float)a0d, a1 = (float)a1d, a2 = (float)a2d;
    const float th2 = a0 * a0 + a1 * a1 + a2 * a2;
    float R[9];
    if (th2 > 1e-20f) {
      const float ith = rsqrtf(th2), th = th2 * ith;
      float sh, ch;
      sincosf(0.5f * th, &sh, &ch);
      const float st = 2.0f * sh * ch, omc = 2.0f * sh * sh;    // 1 - cos without cancellation
      const float w0 = a0 * ith, w1 = a1 * ith, w2 = a2 * ith;
      // R - I = sin [w]x + (1 - cos) (w w^T - I)
      R[0] = omc * (w0 * w0 - 1.0f); R[1] = omc * w0 * w1 - st * w2; R[2] = omc * w0 * w2 + st * w1;
      R[3] = omc * w1 * w0 + st * w2; R[4] = omc * (w1 * w1 - 1.0f); R[5] = omc * w1 * w2 - st * w0;
      R[6] = omc * w2 * w0 - st * w1; R[7] = omc * w2 * w1 + st * w0; R[8] = omc * (w2 * w2 - 1.0f);
    } else {   // first-order branch (the frame role's own: include/Sim3BA.h:61 through ceres::AngleAxisRotatePoint)
      R[0] = 0.0f; R[1] = -a2; R[2] = a1; R[3] = a2; R[4] = 0.0f; R[5] = -a0; R[6] = -a1; R[7] = a0; R[8] = 0.0f;
    }
    const float on = (Pb.pose_blend && fr < nf) ? 1.0f : 0.0f;
#pragma unroll
    for (int e = 0; e < 9; ++e) sf[fr * kK + 9 * j + e] = on * R[e];
  }
  if (tid < kCoefFrames * kMaxShape) sf[bfr * kK + kPoseFeat + bk] = (bfr < nf) ? (float)bv : 0.0f;
  if (tid < kCoefFrames * (kK - kPoseFeat - kMaxShape)) {     // the template's two slots (coefficient 1.0), then K's padding
    const int pf = tid / (kK - kPoseFeat - kMaxShape), pk = tid % (kK - kPoseFeat - kMaxShape);
    sf[pf * kK + kPoseFeat + kMaxShape + pk] = (pk < 2 && pf < nf) ? 1.0f : 0.0f;
  }
  if (9 * (nJ - 1) < kPoseFeat) {   // (models with fewer joints: the unused pose slots)
    for (int i = tid; i < kCoefFrames * (kPoseFeat - 9 * (nJ - 1)); i += kThreads)
      sf[(i / (kPoseFeat - 9 * (nJ - 1))) * kK + 9 * (nJ - 1) + i % (kPoseFeat - 9 * (nJ - 1))] = 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  CSTAMP(1);
  // fragments: item = (k-step, hi / lo, k-half, frame): 16 bytes = eight consecutive coefficients of one frame
  const int unit = f0 / kFTile, phi0 = f0 % kFTile;
  unsigned char* base = reinterpret_cast<unsigned char*>(mc.featA) + (size_t)unit * kBlendKSteps * 2048;
#pragma unroll
  for (int u = 0; u < (kCoefFrames * kBlendKSteps * 4 + kThreads - 1) / kThreads; ++u) {
    const int it = tid + u * kThreads;
    if (it < kCoefFrames * kBlendKSteps * 4) {
      const int kstep = it >> 6, rem = it & 63, hl = rem >> 5, h = (rem >> 4) & 1, fl = rem & 15;
      const int phi = phi0 + fl;
      // MFMA row of the frame inside its unit: accumulator register i of half-wave h holds frame 2 i + h (frame_part_inl.h)
      const int row = Pb.feat_perm ? (8 * (phi >> 3) + 4 * (phi & 1) + ((phi >> 1) & 3)) : phi;
      const float4 v0 = *reinterpret_cast<const float4*>(sf + fl * kK + kstep * 16 + 8 * h);
      const float4 v1 = *reinterpret_cast<const float4*>(sf + fl * kK + kstep * 16 + 8 * h + 4);
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
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
      const u32x4_ val = {pk[0], pk[1], pk[2], pk[3]};
      void* dst = base + (size_t)kstep * 2048 + feat_frag_off(h * 32 + row, hl);
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
    }
  }
  // hand-off (cdna guide, Guideline 16 R1): every storing wave's stores have left, then ONE add for the block's frames
  CSTAMP(2);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  CSTAMP(3);
  if (tid == 0 && nf > 0)
    (void)__hip_atomic_fetch_add(flag_base + (size_t)unit * kUnitCounterStride + kUnitCoefOffset, (unsigned)nf, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
#ifdef BODYFIT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CSTAMP(4);
#endif
}

// One workgroup: vertex tile `vtile`, frames [256 group, 256 group + 256).  flags_ready: the caller has waited for the
// group's hand-off flags (or the operands are from an earlier launch).  `lds`: kRoleLdsBytes.
// unit_ctr / unit_want: this wave's unit counters (k_sweep.hip; word 0: transforms, word kUnitCoefOffset: blend coefficients) and
// the value each shows once the unit's frames have all published; wait_unit(counter) polls one (bounded), fail() marks the launch
// as incomplete.
template <typename WaitUnit, typename Fail>
__device__ __forceinline__ void mesh_role(const DevModel& M, const DevProblem& Pb, const MeshCoef& mc, float* __restrict__ cloud_f,
                                          int vtile, int group, unsigned char* lds_generic, bool beside_its_frames,
                                          const unsigned* unit_ctr, unsigned unit_want, WaitUnit wait_unit, Fail fail) {
  RoleCtx C;
  C.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  C.lane = threadIdx.x & 63;
  lds_u8* lds = (lds_u8*)lds_generic;
  C.ring = lds;
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
  // one look at the unit's counter before anything else (loads return in order: it is back before the operand stream below):
  // a workgroup dispatched late finds its frames long handed over and never polls
  unsigned look0 = 0;
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(look0) : "v"(unit_ctr + kUnitCoefOffset) : "memory");
  // ---- independent of the frame workgroups: thirteen of the tile's fourteen operand slabs (78 KiB), the lane's skinning
  //      weights.  They land under the wait for the hand-off. -----------------------------------------------------------
  const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
  const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
  // A workgroup that is resident from the start of the launch runs BESIDE the frame workgroups it waits for: its stream is a
  // trickle, not a burst — requested all at once (17 MB chip-wide) it stretched the frame workgroups' table loads by 900
  // cycles and their hand-off by 2 us, which every mesh workgroup then waits for; nothing needs it before the hand-off:
  // one slab per ~0.25 us from 1.2 us on (past the frame workgroups' table loads).  A workgroup dispatched later (more
  // frames than one group: its frames were handed over long ago) requests everything at once.
  if (C.wave < 6) {   // six pieces per slab: waves 6 and 7 have none
    if (beside_its_frames) {
      const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t_in < 120) __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll 1
    for (int s = 0; s < kResident; ++s) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(C.dirs_rsrc, (__attribute__((address_space(3))) void*)(C.ring + role_slab_off(s) + C.piece * 1024),
                                               16, C.slab_voff, C.dirs_soff + (unsigned)s * kSlabBytes, 0, 0);
      if (beside_its_frames) __builtin_amdgcn_s_sleep(7);
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
  const __amdgpu_buffer_rsrc_t feat_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(mc.featA, 0, nFT * kBlendKSteps * 2 * 1024, 0x00020000);
  const unsigned feat_off = (unsigned)((active ? ftile : 0) * kBlendKSteps * 2 * 1024);

  // every wave's pieces of the resident slabs have landed (requested microseconds ago), the control word is clear
  volatile unsigned* ctrl = reinterpret_cast<volatile unsigned*>(lds_generic + kRoleCtrlOff);
  if (threadIdx.x == 0) ctrl[0] = 0u;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(look0)::"memory");
  // ---- this wave's 32-frame unit has published its BLEND COEFFICIENTS (the transforms follow ~2 us later and are needed in
  //      front of k-step 9 only): each wave waits for ITS unit alone and starts its blend at once (the
  //      units' slowest frames are 6.6-8.2 us after the launch's start: a wave that starts early has the SIMD to itself for the
  //      first nine k-steps; the workgroup meets again at the barrier in front of k-step 9).  A wait that runs out leaves a mark
  //      the whole workgroup acts on behind that barrier. ------------------------------------------------------------------
  if (active && look0 != unit_want && !wait_unit(unit_ctr + kUnitCoefOffset)) ctrl[0] = 1u;
  RSTAMP(1);
  // from here on the mesh role is the launch's critical path: the frame workgroup beside it is past its hand-off
  __builtin_amdgcn_s_setprio(3);

  f32x16 acc[3];
  u32x4 a[3][2], bq[3][2];
  if (active) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const unsigned soff = feat_off + (unsigned)(ks * 2 * 1024);
      a[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 0), soff, kLoadSc1);
      a[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, feat_frag_off(C.lane, 1), soff, kLoadSc1);
    }
  }
  RSTAMP(2);
  if (active) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const lds_u8* sl = C.ring + role_slab_off(0) + (c * 2) * 1024 + C.lane * 16;
      bq[c][0] = *reinterpret_cast<lds_u32x4*>(sl);
      bq[c][1] = *reinterpret_cast<lds_u32x4*>(sl + 1024);
    }
  }
#define RB(S) RCYC(S); role_blend_step<S>(C, feat_rsrc, feat_off, acc, a, bq)
  // the unit's transforms: one look at their counter four k-steps (~1.5 us) ahead of the barrier behind which the first rows are
  // requested; eight fragment loads are issued behind it, so it is back at vmcnt(8).  Normally complete; else poll (bounded).
  unsigned look1 = unit_want;
  if (active) {
    RB(0); RB(1); RB(2); RB(3); RB(4);
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(look1) : "v"(unit_ctr) : "memory");
    RB(5); RB(6); RB(7); RB(8);
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(look1)::"memory");
    if (look1 != unit_want && !wait_unit(unit_ctr)) ctrl[0] = 1u;
  }
  // k-step 9: every wave has read slabs 0-8 (region T) into registers, and slab 9 too (its reads were issued in k-step 8):
  // region T becomes the waves' transform rings, rows 0-2 are requested now and land under the last five k-steps
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (ctrl[0] != 0u) {   // (workgroup-uniform: written before the barrier) a unit never arrived: leave, the host falls back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fail();
    return;
  }
  if (active) {
    role_dma_row(C, 0); role_dma_row(C, 1); role_dma_row(C, 2);
    asm volatile("" ::: "memory");
    RB(9); RB(10); RB(11); RB(12); RB(13);
  }
#undef RB
  RCYC(14);
  // every wave has read slabs 9-12 (region A): it becomes ring slot 3
  __builtin_amdgcn_sched_barrier(0);   // (MFMAs are not memory operations: without this the last ones drift into the skinning rows)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
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
