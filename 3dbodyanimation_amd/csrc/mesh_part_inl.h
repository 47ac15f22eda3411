// mesh_part_inl.h — batched SMPL forward of all 6890 vertices, the body of k_mesh_blend_lbs (k_sweep.hip: the second launch of
// the two-launch sweep; the one-launch sweep's mesh role is mesh_role_inl.h): blendshapes (MFMA) fused with
// 24-joint linear-blend skinning, f32 out.  Replaces ark::Avatar::update()'s cloud
// (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173; src/main_single_frame.cpp:254).
//
// Per workgroup: one tile of 32 vertices (MFMA N); per wave: "units" of 32 frames (MFMA M) x that tile.
//   blend     D[frame][vertex] per coordinate = [pose feature | beta] . [posedirs | shapedirs - S_root],
//             K = 207 + 10 -> 14 k-steps of v_mfma_f32_32x32x16_bf16, both operands split hi + lo in
//             bf16, three products hi.hi + hi.lo + lo.hi (relative product error <= 2^-16 on
//             displacements of centimetres), f32 accumulation; the template rides in the same contraction on two
//             of K's seven padding slots (coefficient 1.0; one slot for its first 16 bits, one for the remainder)
//   skinning  per frame row: gather the vertex's <= 4 joint transforms (3x4 f32) from LDS, blend and apply
//             in packed f32 (v_pk_fma_f32), one 12-byte store per lane (lane = vertex -> 384 contiguous
//             bytes per half-wave); the blended vertices never touch HBM
// Eight waves per workgroup = two per SIMD.  A wave issues one instruction per 4 cycles whatever its kind, so
// the matrix pipe (288 cycles per k-step) and the vector pipe (about 35 VALU + 12 LDS reads per row) only
// run together when they are fed by different waves: each wave alternates a blend phase and a skinning phase,
// and its partner on the SIMD is in the other phase most of the time.  Both phases are straight-line code
// whose operands were requested at least one step earlier: A fragments (L2) four k-steps ahead in a register
// ring, B fragments (LDS) one k-step ahead, a row's 12 transform reads one row ahead, the next 8-frame
// quarter of transforms (L2 -> registers -> this wave's 9 KiB LDS slice) a quarter ahead.
// Data movement: the vertex tile's B operands (84 KiB, fragment order, contiguous per tile) go
// HBM -> LDS by LDS-DMA once per workgroup and serve every frame.
// LDS: 84 KiB (B) + 8 waves x 9 KiB (transforms) = 156 KiB of the CU's 160 KiB.
#pragma once
#include <hip/hip_ext.h>

#include "bodyfit_device.h"
#include "priors_inl.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // plain vector: HIP's uint4 assigns through memcpy,
                                                                  // which keeps a staging array in scratch

constexpr int kWaves = 8;
constexpr int kPieces = kBlendKSteps * 3 * 2;                    // 84 x 1 KiB, [ks][c][hi/lo]
constexpr int kBBytes = kPieces * 1024;                          // 86,016
constexpr int kRowBytes = kMaxJoints * 48;                       // one frame's 24 transforms: 1,152
constexpr int kQuarterBytes = 8 * kRowBytes;                     // 9,216
constexpr int kLdsBytes = kBBytes + kWaves * kQuarterBytes;      // 159,744
constexpr int kSkinVec = kQuarterBytes / (64 * 16);              // uint4 per lane per quarter = 9
constexpr int kStoreAux = 16;                                    // gfx940+ cache policy bits: 1 sc0, 2 nt, 16 sc1
constexpr int kAhead = 4;                                        // A-fragment ring depth, k-steps

#ifdef BODYFIT_STAMPS
constexpr size_t kStampBase = (size_t)1 << 20;   // behind k_frame_resjac's stamps in the shared diagnostic buffer
#define MSTAMP(i)                                                                             \
  do {                                                                                        \
    if (Pb.dbg && lane == 0) {                                                                \
      unsigned long long t_, c_;                                                              \
      asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_)::"memory"); \
      Pb.dbg[kStampBase + ((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = t_;                                \
      Pb.dbg[kStampBase + ((size_t)blockIdx.x * 8 + wave) * 16 + 8 + (i)] = c_;                            \
    }                                                                                         \
  } while (0)
#else
#define MSTAMP(i)
#endif

__device__ inline void lds_dma_16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct Lane {                    // per-lane constants of the skinning rows
  f32x2 w2[4];                   // skinning weights, broadcast pairs
  const unsigned char* tj[4];    // LDS address of joint i's transform in frame row 4h of the wave's slice
  unsigned out_off;              // byte offset of (frame 4h, vertex v) in the cloud
};

// one quarter of a unit: 8 frames x 24 transforms, contiguous in HBM.  skinT is allocated (and zeroed) for whole
// frame tiles, so the loads need no predicate (a predicated load is a branch, and a branch ends the slot's
// scheduling region).
struct OperandSrc {
  const unsigned char* skinT;
  const uint4* feat;                       // featA
};
__device__ __forceinline__ void skin_load(const OperandSrc& src, int ftile, int q, int lane, u32x4 (&reg)[kSkinVec]) {
  const unsigned char* g = src.skinT + ((size_t)ftile * kFTile + q * 8) * kRowBytes + lane * 16;
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) reg[i] = *reinterpret_cast<const u32x4*>(g + i * 1024);
}
// A fragments of k-step ks of frame tile ftile (hi, lo)
__device__ __forceinline__ void feat_load(const OperandSrc& src, int ftile, int ks, int lane, uint4& hi, uint4& lo) {
  const unsigned char* fa = reinterpret_cast<const unsigned char*>(src.feat) + ((size_t)ftile * kBlendKSteps + ks) * 2048;
  hi = *reinterpret_cast<const uint4*>(fa + feat_frag_off(lane, 0));
  lo = *reinterpret_cast<const uint4*>(fa + feat_frag_off(lane, 1));
}
__device__ __forceinline__ void skin_store(unsigned char* l, int lane, const u32x4 (&reg)[kSkinVec]) {
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) *reinterpret_cast<u32x4*>(l + (i * 64 + lane) * 16) = reg[i];
}

// the 12 LDS reads of row R (frame row (R & 3) + 8 (R >> 2) + 4 h of the unit; the slice holds quarter R >> 2)
template <int R>
__device__ __forceinline__ void row_fetch(const Lane& L, float4 (&t)[12]) {
  constexpr int off = (R & 3) * kRowBytes;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4* T = reinterpret_cast<const float4*>(L.tj[i] + off);
    t[3 * i + 0] = T[0]; t[3 * i + 1] = T[1]; t[3 * i + 2] = T[2];
  }
}

// blend the four transforms, apply to the blended rest vertex, store
template <int R>
__device__ __forceinline__ void row_apply(const Lane& L, const float4 (&t)[12], const f32x16 (&acc)[3],
                                          __amdgpu_buffer_rsrc_t out, unsigned row_off) {
  f32x2 b[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    b[2 * k] = L.w2[0] * f32x2{t[k].x, t[k].y};
    b[2 * k + 1] = L.w2[0] * f32x2{t[k].z, t[k].w};
  }
#pragma unroll
  for (int i = 1; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      b[2 * k] += L.w2[i] * f32x2{t[3 * i + k].x, t[3 * i + k].y};
      b[2 * k + 1] += L.w2[i] * f32x2{t[3 * i + k].z, t[3 * i + k].w};
    }
  const f32x2 pxy = {acc[0][R], acc[1][R]};       // the accumulators hold the rest vertex: the template rides in the
  const f32x2 pz1 = {acc[2][R], 1.0f};            // contraction (two padding K slots with coefficient 1)
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x2 m = b[2 * k] * pxy + b[2 * k + 1] * pz1;
    o[k] = m.x + m.y;
  }
  // unconditional store (a predicated one would split the step's basic block): the cloud's frame stride is padded to
  // whole vertex tiles and its frame count to whole frame tiles, so lanes past V or F land in padding, and each
  // half-wave writes exactly three whole 128-byte lines.  Buffer form: uniform row offset in an SGPR, lane offset in
  // a VGPR, no address arithmetic; write-through (sc1) so that the 21 MB of output drain to HBM while the kernel
  // runs instead of sitting dirty in L2 until the end-of-kernel release flushes them.
  typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;
  const u32x3 pk = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2])};
  __builtin_amdgcn_raw_buffer_store_b96(pk, out, L.out_off, row_off, kStoreAux);
}

// ---- blend phase: k-step S of the unit whose A fragments start at fa --------------------------------------------
template <int S>
__device__ __forceinline__ void blend_step(const unsigned char* sB, int lane, const OperandSrc& src, int ftile, f32x16 (&acc)[3],
                                           uint4 (&a)[kBlendKSteps][2], uint4 (&bq)[2][3][2]) {
  if constexpr (S + kAhead < kBlendKSteps) feat_load(src, ftile, S + kAhead, lane, a[S + kAhead][0], a[S + kAhead][1]);
  if constexpr (S + 1 < kBlendKSteps) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint4* bp = reinterpret_cast<const uint4*>(sB + (size_t)(((S + 1) * 3 + c) * 2) * 1024) + lane;
      bq[(S + 1) & 1][c][0] = bp[0];
      bq[(S + 1) & 1][c][1] = bp[64];
    }
  }
  const bf16x8 a_hi = __builtin_bit_cast(bf16x8, a[S][0]);
  const bf16x8 a_lo = __builtin_bit_cast(bf16x8, a[S][1]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const bf16x8 bhi = __builtin_bit_cast(bf16x8, bq[S & 1][c][0]);
    const bf16x8 blo = __builtin_bit_cast(bf16x8, bq[S & 1][c][1]);
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
  }
  // issue order: the next k-step's B reads and A loads go out under the first MFMAs of this one
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // VMEM read
  __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- skinning phase: row S of the unit at frame tile ftile ----------------------------------------------------------
template <int S>
__device__ __forceinline__ void skin_step(const Lane& L, unsigned char* sSkin, int lane, unsigned stride, const OperandSrc& src,
                                          __amdgpu_buffer_rsrc_t cloud, int ftile, const f32x16 (&acc)[3], float4 (&tq)[2][12],
                                          u32x4 (&treg)[kSkinVec]) {
  constexpr int q = S >> 2;
  if constexpr ((S & 3) == 0) {
    // quarter boundary: every read of the previous quarter has been issued (LDS serves a wave in order), so the slice
    // is overwritten with this quarter (in registers since the previous boundary) and the next one goes in flight
    skin_store(sSkin, lane, treg);
    if constexpr (q < 3) skin_load(src, ftile, q + 1, lane, treg);
    row_fetch<S>(L, tq[S & 1]);
  }
  if constexpr ((S & 3) != 3) row_fetch<S + 1>(L, tq[(S + 1) & 1]);
  const int f = ftile * kFTile + q * 8 + (S & 3);               // frame of the h = 0 half
  row_apply<S>(L, tq[S & 1], acc, cloud, (unsigned)f * stride);
  if constexpr ((S & 3) == 1 || (S & 3) == 2) {
    // issue order of the row: the NEXT row's 12 transform reads first (a whole row of VALU work ahead of their use),
    // then this row's arithmetic, then the store.  (One group alone does not pin a position; the sequence does.)
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);   // DS read
    __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);   // VALU
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);    // VMEM write
  }
  __builtin_amdgcn_sched_barrier(0);
}

// One vertex tile x all frames.  sB: the tile's operand image in LDS (84 KiB); sSkinBase: 8 x 9 KiB of transform slices.
__device__ __forceinline__ void mesh_part(const DevModel& M, const DevProblem& Pb, const MeshCoef& mc, float* __restrict__ cloud_f,
                                          int vtile, unsigned char* sB, unsigned char* sSkinBase) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int col = lane & 31, h = lane >> 5;
  const int v = vtile * kVTile + col;
  const int nFT = Pb.nFTiles;

  MSTAMP(0);
  unsigned char* sSkin = sSkinBase + wave * kQuarterBytes;             // this wave's transform slice

  // ---- stage this vertex tile's B operands: HBM -> LDS, 1 KiB per wave-instruction -------------------
  {
    const unsigned char* gp = reinterpret_cast<const unsigned char*>(M.dirsB) + (size_t)vtile * kBBytes;
#pragma unroll
    for (int i = 0; i < (kPieces + kWaves - 1) / kWaves; ++i) {
      const int p = i * kWaves + wave;
      if (p < kPieces) lds_dma_16(gp + (size_t)p * 1024 + lane * 16, sB + (size_t)p * 1024);
    }
  }
  Lane L;
  {
    const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
    const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
    const float wgt[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      L.w2[i] = f32x2{wgt[i], wgt[i]};
      L.tj[i] = sSkin + 4 * h * kRowBytes + (int)((widx >> (8 * i)) & 0xffu) * 48;
    }
  }
  const unsigned stride = (unsigned)M.nVTiles * kVTile * 12;           // bytes per frame of the cloud
  L.out_off = 4 * h * stride + (unsigned)v * 12;
  // raw buffer over the padded cloud (launch_mesh checks that it is < 4 GiB); 0x00020000 = gfx9 raw-buffer DATA_FORMAT
  const __amdgpu_buffer_rsrc_t cloud =
      __builtin_amdgcn_make_buffer_rsrc(cloud_f, 0, (int)((unsigned)nFT * kFTile * stride), 0x00020000);

  // this wave's units: frame tiles wave, wave + 8, ...
  OperandSrc src;
  src.skinT = reinterpret_cast<const unsigned char*>(mc.skinT);
  src.feat = reinterpret_cast<const uint4*>(mc.featA);
  f32x16 acc[3];
  uint4 a[kBlendKSteps][2], bq[2][3][2];
  u32x4 treg[kSkinVec];
  float4 tq[2][12];
  if (wave < nFT) {
#pragma unroll
    for (int ks = 0; ks < kAhead; ++ks) feat_load(src, wave, ks, lane, a[ks][0], a[ks][1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MSTAMP(1);
  // Waves w and w + 4 share a SIMD.  Left alone they run their phases in lock-step (both blending at half the matrix
  // rate, then both skinning against each other for LDS bandwidth); with the second wave at a higher issue priority
  // its blend finishes first and the two stay in opposite phases.
  if (wave >= 4) __builtin_amdgcn_s_setprio(2);

  for (int ftile = wave; ftile < nFT; ftile += kWaves) {
    skin_load(src, ftile, 0, lane, treg);                      // quarter 0: in flight across the blend phase
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint4* bp = reinterpret_cast<const uint4*>(sB + (size_t)(c * 2) * 1024) + lane;
      bq[0][c][0] = bp[0]; bq[0][c][1] = bp[64];
    }
#define BSTEP(S) blend_step<S>(sB, lane, src, ftile, acc, a, bq)
    BSTEP(0); BSTEP(1); BSTEP(2); BSTEP(3); BSTEP(4); BSTEP(5); BSTEP(6);
    BSTEP(7); BSTEP(8); BSTEP(9); BSTEP(10); BSTEP(11); BSTEP(12); BSTEP(13);
#undef BSTEP
    MSTAMP(2);
#define SSTEP(S) skin_step<S>(L, sSkin, lane, stride, src, cloud, ftile, acc, tq, treg)
    SSTEP(0); SSTEP(1); SSTEP(2); SSTEP(3); SSTEP(4); SSTEP(5); SSTEP(6); SSTEP(7);
    SSTEP(8); SSTEP(9); SSTEP(10); SSTEP(11); SSTEP(12);
    if (ftile + kWaves < nFT) {      // next unit's first A fragments (the transform staging registers are free now)
#pragma unroll
      for (int ks = 0; ks < kAhead; ++ks) feat_load(src, ftile + kWaves, ks, lane, a[ks][0], a[ks][1]);
    }
    SSTEP(13); SSTEP(14); SSTEP(15);
#undef SSTEP
    MSTAMP(3);
  }
  if (wave >= 4) __builtin_amdgcn_s_setprio(0);
}

}  // namespace
}  // namespace bodyfit
