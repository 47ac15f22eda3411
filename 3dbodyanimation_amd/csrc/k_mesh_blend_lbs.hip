// k_mesh_blend_lbs.hip — batched SMPL forward of all 6890 vertices: blendshapes (MFMA) fused with
// 24-joint linear-blend skinning, f32 out.  Replaces ark::Avatar::update()'s cloud
// (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173; src/main_single_frame.cpp:254).
//
// Per workgroup: one tile of 32 vertices (MFMA N); per wave: "units" of 32 frames (MFMA M) x that tile.
//   blend     D[frame][vertex] per coordinate = [pose feature | beta] . [posedirs | shapedirs - S_root],
//             K = 207 + 10 -> 14 k-steps of v_mfma_f32_32x32x16_bf16, both operands split hi + lo in
//             bf16, three products hi.hi + hi.lo + lo.hi (relative product error <= 2^-16 on
//             displacements of centimetres), f32 accumulation; the template is added in f32 afterwards
//   skinning  per frame row: gather the vertex's <= 4 joint transforms (3x4 f32) from LDS, blend, apply,
//             one 12-byte store per lane (lane = vertex -> 384 contiguous bytes per half-wave)
// Software pipeline (one wave per SIMD, 4 per workgroup, up to 512 registers each): the k-loop of unit
// n+1 and the 16 skinning rows of unit n are ONE straight-line body of 16 slots, slot s = k-step s
// (9 MFMAs, 288 matrix-pipe cycles) + row s (about 50 vector instructions): the MFMA only holds the vector
// issue port for 8 of its 32 cycles, so the row's VALU work runs underneath.  Everything a slot consumes was
// requested at least one slot earlier: A fragments (L2) six k-steps ahead in a register ring, B fragments
// (LDS) one k-step ahead, the row's 12 transform reads (LDS) one row ahead, the transforms of the next
// 8-frame quarter (L2 -> registers -> this wave's private LDS double buffer) a whole quarter ahead.
// Data movement: the vertex tile's B operands (84 KiB, fragment order, contiguous per tile) go
// HBM -> LDS by LDS-DMA once per workgroup and serve every frame; nothing blended touches HBM.
// LDS: 84 KiB (B) + 4 waves x 2 x 9 KiB (transforms) = 156 KiB of the CU's 160 KiB.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // plain vector: HIP's uint4 assigns through memcpy,
                                                                  // which keeps a staging array in scratch

constexpr int kWaves = 4;
constexpr int kPieces = kBlendKSteps * 3 * 2;                    // 84 x 1 KiB, [ks][c][hi/lo]
constexpr int kBBytes = kPieces * 1024;                          // 86,016
constexpr int kRowBytes = kMaxJoints * 48;                       // one frame's 24 transforms: 1,152
constexpr int kQuarterBytes = 8 * kRowBytes;                     // 9,216
constexpr int kLdsBytes = kBBytes + kWaves * 2 * kQuarterBytes;  // 159,744
constexpr int kSkinVec = kQuarterBytes / (64 * 16);              // uint4 per lane per quarter = 9
constexpr int kAhead = 6;                                        // A-fragment ring depth, k-steps

#ifdef BODYFIT_STAMPS
#define MSTAMP(i)                                                                             \
  do {                                                                                        \
    if (Pb.dbg && lane == 0) {                                                                \
      unsigned long long t_, c_;                                                              \
      asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_)::"memory"); \
      Pb.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = t_;                                \
      Pb.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 8 + (i)] = c_;                            \
    }                                                                                         \
  } while (0)
#else
#define MSTAMP(i)
#endif

__device__ inline void lds_dma_16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct Lane {                    // per-lane constants of the skinning rows
  float vt[3];                   // template (centred on the rest root joint)
  f32x2 w2[4];                   // skinning weights, broadcast pairs
  const unsigned char* tj[4];    // LDS address of joint i's transform in frame row 4h of buffer 0
  unsigned out_off;              // byte offset of (frame 4h, vertex v) in the cloud
  unsigned char* dump;           // where lanes past the last vertex store (padding behind the cloud)
  bool v_ok;
};

// one quarter of a unit: 8 frames x 24 transforms, contiguous in HBM.  skinT is allocated (and zeroed) for whole
// frame tiles, so the loads need no predicate (a predicated load is a branch, and a branch ends the slot's
// scheduling region).
__device__ __forceinline__ void skin_load(const unsigned char* skinT, int ftile, int q, int lane, u32x4 (&reg)[kSkinVec]) {
  const unsigned char* g = skinT + ((size_t)ftile * kFTile + q * 8) * kRowBytes + lane * 16;
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) reg[i] = *reinterpret_cast<const u32x4*>(g + i * 1024);
}
__device__ __forceinline__ void skin_store(unsigned char* l, int lane, const u32x4 (&reg)[kSkinVec]) {
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) *reinterpret_cast<u32x4*>(l + (i * 64 + lane) * 16) = reg[i];
}

// the 12 LDS reads of row R (frame row (R & 3) + 8 (R >> 2) + 4 h of the unit) — buffer parity (R >> 2) & 1
template <int R>
__device__ __forceinline__ void row_fetch(const Lane& L, float4 (&t)[12]) {
  constexpr int off = ((R >> 2) & 1) * kQuarterBytes + (R & 3) * kRowBytes;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4* T = reinterpret_cast<const float4*>(L.tj[i] + off);
    t[3 * i + 0] = T[0]; t[3 * i + 1] = T[1]; t[3 * i + 2] = T[2];
  }
}

// blend the four transforms, apply to the blended rest vertex, store
template <int R>
__device__ __forceinline__ void row_apply(const Lane& L, const float4 (&t)[12], const f32x16 (&acc)[3],
                                          unsigned char* out_row) {
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
  const f32x2 pxy = {acc[0][R] + L.vt[0], acc[1][R] + L.vt[1]};
  const f32x2 pz1 = {acc[2][R] + L.vt[2], 1.0f};
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x2 m = b[2 * k] * pxy + b[2 * k + 1] * pz1;
    o[k] = m.x + m.y;
  }
  // unconditional store (a predicated one splits the slot's basic block and with it the MFMA/VALU interleave):
  // frames past F land in the cloud's tile padding, lanes past V in the dump row
  struct alignas(4) F3 { float x, y, z; };
  unsigned char* dst = L.v_ok ? out_row + L.out_off : L.dump;
  *reinterpret_cast<F3*>(dst) = F3{o[0], o[1], o[2]};
}

struct Unit {            // where one unit's operands live
  const uint4* fa;       // A fragments: featA + ftile * 14 * 2 * 64 + lane
  int ftile;
};

// One slot of the pipeline.  kM: k-step S of unit `um` accumulates into accN; kE: row S of unit `ue` is
// skinned from accC.
template <int S, bool kM, bool kE>
__device__ __forceinline__ void slot(const Lane& L, const unsigned char* sB, unsigned char* sSkin, int lane, int F, int V,
                                     const unsigned char* skinT, unsigned char* cloud, const Unit& um, const Unit& um_next,
                                     const Unit& ue, int ft_q1, int q_q1, int ft_q2, int q_q2, f32x16 (&accN)[3],
                                     const f32x16 (&accC)[3], uint4 (&a)[kBlendKSteps][2], uint4 (&bq)[2][3][2],
                                     float4 (&tq)[2][12], u32x4 (&treg)[kSkinVec]) {
  constexpr int q = S >> 2;
  if constexpr (kE && (S & 3) == 0) {
    // quarter boundary: the next quarter's transforms (in registers since the previous boundary) go to
    // the other buffer, the one after that goes in flight.  (ft_q1, q_q1) / (ft_q2, q_q2) name the quarters
    // one / two after quarter q of unit ue; they may belong to the wave's next unit.
    skin_store(sSkin + ((q + 1) & 1) * kQuarterBytes, lane, treg);
    const int ft2 = (q + 2 < 4) ? ue.ftile : ft_q2;
    skin_load(skinT, ft2, (q + 2) & 3, lane, treg);
    (void)ft_q1; (void)q_q1; (void)q_q2;
  }
  if constexpr (kM && S < kBlendKSteps) {
    constexpr int sa = S + kAhead;
    if constexpr (sa < kBlendKSteps) {
      a[sa][0] = um.fa[(size_t)sa * 128]; a[sa][1] = um.fa[(size_t)sa * 128 + 64];
    } else {
      a[sa - kBlendKSteps][0] = um_next.fa[(size_t)(sa - kBlendKSteps) * 128];
      a[sa - kBlendKSteps][1] = um_next.fa[(size_t)(sa - kBlendKSteps) * 128 + 64];
    }
    constexpr int sb = (S + 1) % kBlendKSteps;       // B is the same for every unit
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint4* bp = reinterpret_cast<const uint4*>(sB + (size_t)((sb * 3 + c) * 2) * 1024) + lane;
      bq[(S + 1) & 1][c][0] = bp[0];
      bq[(S + 1) & 1][c][1] = bp[64];
    }
  }
  if constexpr (kE) row_fetch<(S + 1) & 15>(L, tq[(S + 1) & 1]);
  if constexpr (kM && S < kBlendKSteps) {
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
        accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, z, 0, 0, 0);
      } else {
        accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, accN[c], 0, 0, 0);
      }
      accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, blo, accN[c], 0, 0, 0);
      accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bhi, accN[c], 0, 0, 0);
    }
  }
  if constexpr (kE) {
    const int f = ue.ftile * kFTile + q * 8 + (S & 3);               // frame of the h = 0 half
    unsigned char* out_row = cloud + (size_t)f * V * 12;
    row_apply<S>(L, tq[S & 1], accC, out_row);
  }
  if constexpr (kM && kE && S < kBlendKSteps) {
    // issue order inside the slot: every MFMA is followed by the LDS reads and vector work that fit under it
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // VALU
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <bool kM, bool kE>
__device__ __forceinline__ void body(const Lane& L, const unsigned char* sB, unsigned char* sSkin, int lane, int F, int V,
                                     const unsigned char* skinT, unsigned char* cloud, const Unit& um, const Unit& um_next,
                                     const Unit& ue, int ft_next, f32x16 (&accN)[3], const f32x16 (&accC)[3],
                                     uint4 (&a)[kBlendKSteps][2], uint4 (&bq)[2][3][2], float4 (&tq)[2][12],
                                     u32x4 (&treg)[kSkinVec]) {
#define SLOT(S) slot<S, kM, kE>(L, sB, sSkin, lane, F, V, skinT, cloud, um, um_next, ue, 0, 0, ft_next, 0, accN, accC, a, bq, tq, treg)
  SLOT(0); SLOT(1); SLOT(2); SLOT(3); SLOT(4); SLOT(5); SLOT(6); SLOT(7);
  SLOT(8); SLOT(9); SLOT(10); SLOT(11); SLOT(12); SLOT(13); SLOT(14); SLOT(15);
#undef SLOT
}

__global__ __launch_bounds__(64 * kWaves, 1) void k_mesh_blend_lbs(DevModel M, DevProblem Pb, MeshCoef mc,
                                                                   float* __restrict__ cloud_f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int vtile = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int col = lane & 31, h = lane >> 5;
  const int v = vtile * kVTile + col;
  const int V = M.V, F = Pb.F, nFT = Pb.nFTiles;
  unsigned char* cloud = reinterpret_cast<unsigned char*>(cloud_f);
  const unsigned char* skinT = reinterpret_cast<const unsigned char*>(mc.skinT);

  MSTAMP(0);
  unsigned char* sB = lds;                                             // [ks][c][hi/lo][64][16 B]
  unsigned char* sSkin = lds + kBBytes + wave * 2 * kQuarterBytes;     // this wave's two quarter buffers

  // ---- stage this vertex tile's B operands: HBM -> LDS, 1 KiB per wave-instruction -------------------
  {
    const unsigned char* gp = reinterpret_cast<const unsigned char*>(M.dirsB) + (size_t)vtile * kBBytes;
#pragma unroll
    for (int i = 0; i < kPieces / kWaves; ++i) {
      const int p = i * kWaves + wave;
      lds_dma_16(gp + (size_t)p * 1024 + lane * 16, sB + (size_t)p * 1024);
    }
  }
  Lane L;
#pragma unroll
  for (int c = 0; c < 3; ++c) L.vt[c] = M.vtB[((size_t)vtile * 3 + c) * 32 + col];
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
  L.out_off = (unsigned)(((size_t)4 * h * V + v) * 12);
  L.v_ok = v < V;
  L.dump = cloud + (size_t)nFT * kFTile * V * 12 + lane * 12;

  // this wave's units: frame tiles wave, wave + 4, ...
  const uint4* feat = reinterpret_cast<const uint4*>(mc.featA) + lane;
  auto unit = [&](int n) {
    const int ft = min(wave + n * kWaves, nFT - 1);                    // clamped: prefetches past the end re-read
    return Unit{feat + (size_t)ft * kBlendKSteps * 2 * 64, ft};
  };
  const int n_units = (nFT - wave + kWaves - 1) / kWaves;              // may be 0
  f32x16 accN[3], accC[3];
  uint4 a[kBlendKSteps][2], bq[2][3][2];
  u32x4 treg[kSkinVec];
  float4 tq[2][12];
  Unit u0 = unit(0);
  if (n_units > 0) {
#pragma unroll
    for (int ks = 0; ks < kAhead; ++ks) { a[ks][0] = u0.fa[(size_t)ks * 128]; a[ks][1] = u0.fa[(size_t)ks * 128 + 64]; }
    skin_load(skinT, u0.ftile, 0, lane, treg);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MSTAMP(1);
  if (n_units <= 0) return;

  // prologue: blend of unit 0, then bootstrap the transform pipeline
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint4* bp = reinterpret_cast<const uint4*>(sB + (size_t)(c * 2) * 1024) + lane;
    bq[0][c][0] = bp[0]; bq[0][c][1] = bp[64];
  }
  {
    const Unit u1 = unit(1);
    body<true, false>(L, sB, sSkin, lane, F, V, skinT, cloud, u0, u1, u0, 0, accN, accC, a, bq, tq, treg);
  }
  skin_store(sSkin, lane, treg);                                       // quarter 0 -> buffer 0
  skin_load(skinT, u0.ftile, 1, lane, treg);                        // quarter 1 in flight
  row_fetch<0>(L, tq[0]);
  MSTAMP(2);

  for (int n = 0; n < n_units; ++n) {
#pragma unroll
    for (int c = 0; c < 3; ++c) accC[c] = accN[c];
    const Unit ue = unit(n), um = unit(n + 1), um_next = unit(n + 2);
    const int ft_next = um.ftile;     // clamped past the end: those quarters are staged but never read
    if (n + 1 < n_units)
      body<true, true>(L, sB, sSkin, lane, F, V, skinT, cloud, um, um_next, ue, ft_next, accN, accC, a, bq, tq, treg);
    else
      body<false, true>(L, sB, sSkin, lane, F, V, skinT, cloud, um, um_next, ue, ft_next, accN, accC, a, bq, tq, treg);
  }
  MSTAMP(3);
}

}  // namespace

void launch_mesh(const DevModel& M, const DevProblem& P, const MeshCoef& mc, float* d_cloud, hipStream_t s) {
  if (P.F <= 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mesh_blend_lbs),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_mesh_blend_lbs, dim3(M.nVTiles), dim3(64 * kWaves), kLdsBytes, s, M, P, mc, d_cloud);
}

}  // namespace bodyfit
