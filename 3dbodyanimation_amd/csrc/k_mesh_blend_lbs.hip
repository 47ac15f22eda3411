// k_mesh_blend_lbs.hip — batched SMPL forward of all 6890 vertices: blendshapes (MFMA) fused with
// 24-joint linear-blend skinning, f32 out.  Replaces ark::Avatar::update()'s cloud
// (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173; src/main_single_frame.cpp:254).
//
// Per workgroup: one tile of 32 vertices (MFMA N), all frames in tiles of 32 (MFMA M), K = blend
// coefficients.  D[frame][vertex] per coordinate, accumulated in f32:
//   template        v_template (centred on the rest root joint), added to the accumulators in the epilogue
//   shape blend     v_mfma_f32_32x32x2_f32, K = 10 -> 5 steps, exact f32 (per-frame beta supported)
//   pose blend      v_mfma_f32_32x32x16_bf16, K = 207 -> 13 steps, operands split hi+lo in bf16 and
//                   three products hi.hi + hi.lo + lo.hi (relative product error <= 2^-16), 5.3x the
//                   f32-MFMA rate
// Data movement (the 17 MB posedirs stream is the only large read):
//   * the vertex tile's B operands (78 KiB pose hi/lo + 1.9 KiB shape, stored in fragment order at
//     upload) go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB contiguous per wave-instruction),
//     once per workgroup, and are reused by all four waves for every frame tile;
//   * A operands (pose-feature hi/lo, beta) come from L2 in fragment order, one 1 KiB load per k-step;
//   * the skinning transforms of 8 frames (9 KiB, contiguous) are register-staged per wave (loads for the
//     next 8 frames issued before the current 8 are skinned, written to LDS afterwards) and gathered by
//     joint id with ds_read_b128; the blended vertices never touch HBM;
//   * output rows are 384 contiguous bytes per frame (lane = vertex).
// Eight waves per workgroup = two per SIMD, so one wave's LDS-DMA / store-retire waits (CDNA4 counts
// stores in vmcnt) are covered by its partner's MFMA or skinning VALU work.
// LDS: 79.9 KiB (B) + 3.75 KiB (shape) + 8 x 9 KiB (transforms) = 153.75 KiB of the CU's 160 KiB.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kPosePieces = 3 * kPoseKSteps * 2;                 // 78 x 1 KiB
constexpr int kPoseBytes = kPosePieces * 1024;                   // 79,872
constexpr int kShapeFloats = 3 * kShapeKSteps * 64;              // 960 f32 = 3,840 B
constexpr int kSkinRows = 8;                                     // frames per epilogue quarter
constexpr int kWaves = 8;
constexpr int kSkinBytes = kSkinRows * kMaxJoints * 48;          // 9,216
constexpr int kLdsBytes = kPoseBytes + kShapeFloats * 4 + kWaves * kSkinBytes;   // 157,440

#ifdef BODYFIT_STAMPS
#define MSTAMP(i)                                                                             \
  do {                                                                                        \
    if (Pb.dbg && (threadIdx.x & 63) == 0) {                                                  \
      unsigned long long t_;                                                                  \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      Pb.dbg[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_;                  \
    }                                                                                         \
  } while (0)
#else
#define MSTAMP(i)
#endif

__device__ inline void lds_dma_16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// One output row of the skinning epilogue.  R (compile-time) = accumulator register = frame row
// (R & 3) + 8 (R >> 2) + 4 h of the tile; the row's 24 transforms sit at Trow in this wave's LDS slice.
template <int R>
__device__ __forceinline__ void skin_row(const f32x16 (&acc)[3], const float (&vt)[3], const unsigned char* Trow,
                                         const int (&jo)[4], const float (&wgt)[4], float* __restrict__ o, bool live) {
  float4 t0 = make_float4(0, 0, 0, 0), t1 = t0, t2 = t0;
#pragma unroll
  for (int i = 0; i < kMeshNnz; ++i) {
    const float4* T = reinterpret_cast<const float4*>(Trow + jo[i]);
    const float w = wgt[i];
    const float4 a0 = T[0], a1 = T[1], a2 = T[2];
    t0.x += w * a0.x; t0.y += w * a0.y; t0.z += w * a0.z; t0.w += w * a0.w;
    t1.x += w * a1.x; t1.y += w * a1.y; t1.z += w * a1.z; t1.w += w * a1.w;
    t2.x += w * a2.x; t2.y += w * a2.y; t2.z += w * a2.z; t2.w += w * a2.w;
  }
  const float px = acc[0][R] + vt[0], py = acc[1][R] + vt[1], pz = acc[2][R] + vt[2];
  if (live) {
    o[0] = t0.x * px + t0.y * py + t0.z * pz + t0.w;
    o[1] = t1.x * px + t1.y * py + t1.z * pz + t1.w;
    o[2] = t2.x * px + t2.y * py + t2.z * pz + t2.w;
  }
}

constexpr int kSkinVec = kSkinBytes / (64 * 16);   // uint4 per lane per 8-frame block = 9

// issue the loads of one 8-frame block of skinning transforms (contiguous in HBM) into registers
__device__ __forceinline__ void skin_load(const unsigned char* g, int nbytes, int lane, uint4 (&reg)[kSkinVec]) {
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) {
    const int off = (i * 64 + lane) * 16;
    reg[i] = (off < nbytes) ? *reinterpret_cast<const uint4*>(g + off) : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void skin_store(unsigned char* l, int lane, const uint4 (&reg)[kSkinVec]) {
#pragma unroll
  for (int i = 0; i < kSkinVec; ++i) *reinterpret_cast<uint4*>(l + (i * 64 + lane) * 16) = reg[i];
}

__global__ __launch_bounds__(64 * kWaves, 2) void k_mesh_blend_lbs(DevModel M, DevProblem Pb, MeshCoef mc,
                                                         float* __restrict__ cloud) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int vtile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, h = lane >> 5;
  const int v = vtile * kVTile + col;
  const int V = M.V, nJ = M.nJ, F = Pb.F;
  const bool pose = Pb.pose_blend && M.P > 0;

  MSTAMP(0);
  unsigned char* sPose = lds;                                           // [ks][c][hl][64][16 B]
  float* sShape = reinterpret_cast<float*>(lds + kPoseBytes);           // [c][ks][64]
  unsigned char* sSkin = lds + kPoseBytes + kShapeFloats * 4 + wave * kSkinBytes;

  // ---- stage this vertex tile's B operands: HBM -> LDS, 1 KiB per wave-instruction -----------------
  {
    const unsigned char* gp = reinterpret_cast<const unsigned char*>(M.dirsB) + (size_t)vtile * kPoseBytes;
    if (pose) {
      for (int p = wave; p < kPosePieces; p += kWaves) {
        // LDS piece p = (ks, c, hl) k-step-major; global layout is [c][ks][hl]
        const int ks = p / 6, c = (p % 6) >> 1, hl = p & 1;
        const int gpiece = (c * kPoseKSteps + ks) * 2 + hl;
        lds_dma_16(gp + (size_t)gpiece * 1024 + lane * 16, sPose + (size_t)p * 1024);
      }
    }
    const float* gs = M.sdB + (size_t)vtile * kShapeFloats;
    for (int i = threadIdx.x; i < kShapeFloats; i += 64 * kWaves) sShape[i] = gs[i];
  }
  float vt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) vt[c] = M.vtB[((size_t)vtile * 3 + c) * 32 + col];
  const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
  const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
  const float wgt[4] = {wv.x, wv.y, wv.z, wv.w};
  int jo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) jo[i] = (int)((widx >> (8 * i)) & 0xffu) * 48;
  const uint4* feat = reinterpret_cast<const uint4*>(mc.featA);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MSTAMP(1);

  for (int ftile = wave; ftile < Pb.nFTiles; ftile += kWaves) {
    // first 8-frame block of skinning transforms: loads in flight across the whole MFMA section
    const unsigned char* gskin = reinterpret_cast<const unsigned char*>(mc.skinT) + (size_t)ftile * 32 * nJ * 48;
    const int frames_left = F - ftile * 32;
    uint4 treg[kSkinVec];
    skin_load(gskin, min(kSkinRows, frames_left) * nJ * 48, lane, treg);
    // A fragments (pose-feature hi/lo, L2) run kAhead k-steps ahead of the MFMAs in a register ring
    constexpr int kAhead = 5;
    const uint4* fa = feat + ((size_t)ftile * kPoseKSteps * 2) * 64 + lane;
    uint4 ahi[kPoseKSteps], alo[kPoseKSteps];
    if (pose) {
#pragma unroll
      for (int ks = 0; ks < kAhead; ++ks) { ahi[ks] = fa[(size_t)ks * 128]; alo[ks] = fa[(size_t)ks * 128 + 64]; }
    }

    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;   // the template is added in the epilogue (a 16-wide
                                                       // loop-invariant C-in per coordinate costs 48 VGPRs)

    // shape blend, exact f32
#pragma unroll
    for (int ks = 0; ks < kShapeKSteps; ++ks) {
      const float a = mc.betaA[((size_t)ftile * kShapeKSteps + ks) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float b = sShape[(c * kShapeKSteps + ks) * 64 + lane];
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      }
    }
    // pose blend, bf16 hi/lo split; A fragments straight from L2, B fragments from LDS
    if (pose) {
#pragma unroll
      for (int ks = 0; ks < kPoseKSteps; ++ks) {
        if (ks + kAhead < kPoseKSteps) {
          ahi[ks + kAhead] = fa[(size_t)(ks + kAhead) * 128];
          alo[ks + kAhead] = fa[(size_t)(ks + kAhead) * 128 + 64];
        }
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ahi[ks]);
        const bf16x8 a_lo = __builtin_bit_cast(bf16x8, alo[ks]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint4* bp = reinterpret_cast<const uint4*>(sPose + (size_t)((ks * 3 + c) * 2) * 1024) + lane;
          const bf16x8 bhi = __builtin_bit_cast(bf16x8, bp[0]);
          const bf16x8 blo = __builtin_bit_cast(bf16x8, bp[64]);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, blo, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bhi, acc[c], 0, 0, 0);
        }
        asm volatile("" ::: "memory");   // keep the A ring at kAhead k-steps (no further hoisting: VGPR budget)
      }
    }

    MSTAMP(2);
    // skinning epilogue, one frame row per step (lane = vertex): accumulator register r holds frame row
    // (r & 3) + 8 (r >> 2) + 4 h, so registers 4q..4q+3 cover rows 8q..8q+7 = one LDS-DMA'd quarter.
    // The runtime loop + switch keeps every row's 12 transform reads in its own scheduling region
    // (a fully unrolled epilogue hoists 192 VGPRs of reads and spills at two waves per SIMD).
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
      const int q = r >> 2, rr = r & 3;
      if (rr == 0) {
        // this block's transforms: registers -> this wave's LDS slice; next block's loads go in flight
        skin_store(sSkin, lane, treg);
        if (q < 3)
          skin_load(gskin + (size_t)(q + 1) * kSkinRows * nJ * 48,
                    max(0, min(kSkinRows, frames_left - (q + 1) * kSkinRows)) * nJ * 48, lane, treg);
        __builtin_amdgcn_wave_barrier();
        if (q == 0) MSTAMP(3);
      }
      const int f = ftile * kFTile + q * 8 + rr + 4 * h;
      const unsigned char* Trow = sSkin + (rr + 4 * h) * (nJ * 48);
      float* o = cloud + ((size_t)f * V + v) * 3;
      const bool live = f < F && v < V;
      switch (r) {
        case 0: skin_row<0>(acc, vt, Trow, jo, wgt, o, live); break;
        case 1: skin_row<1>(acc, vt, Trow, jo, wgt, o, live); break;
        case 2: skin_row<2>(acc, vt, Trow, jo, wgt, o, live); break;
        case 3: skin_row<3>(acc, vt, Trow, jo, wgt, o, live); break;
        case 4: skin_row<4>(acc, vt, Trow, jo, wgt, o, live); break;
        case 5: skin_row<5>(acc, vt, Trow, jo, wgt, o, live); break;
        case 6: skin_row<6>(acc, vt, Trow, jo, wgt, o, live); break;
        case 7: skin_row<7>(acc, vt, Trow, jo, wgt, o, live); break;
        case 8: skin_row<8>(acc, vt, Trow, jo, wgt, o, live); break;
        case 9: skin_row<9>(acc, vt, Trow, jo, wgt, o, live); break;
        case 10: skin_row<10>(acc, vt, Trow, jo, wgt, o, live); break;
        case 11: skin_row<11>(acc, vt, Trow, jo, wgt, o, live); break;
        case 12: skin_row<12>(acc, vt, Trow, jo, wgt, o, live); break;
        case 13: skin_row<13>(acc, vt, Trow, jo, wgt, o, live); break;
        case 14: skin_row<14>(acc, vt, Trow, jo, wgt, o, live); break;
        default: skin_row<15>(acc, vt, Trow, jo, wgt, o, live); break;
      }
      if (rr == 3) __builtin_amdgcn_wave_barrier();   // every lane has read this block before it is overwritten
    }
    __builtin_amdgcn_wave_barrier();
    MSTAMP(4);
  }
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
