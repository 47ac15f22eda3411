// k_mesh_blend_lbs.hip — batched SMPL forward of all 6890 vertices: blendshapes (MFMA) fused with
// 24-joint linear-blend skinning, f32 out.  Replaces ark::Avatar::update()'s cloud
// (call sites include/Sim3BA.h:371,538; include/MultiFrameBA.h:53,173; src/main_single_frame.cpp:254).
//
// Per workgroup: one tile of 32 vertices (MFMA N), all frames in tiles of 32 (MFMA M), K = blend
// coefficients.  D[frame][vertex] per coordinate, accumulated in f32:
//   C-in            v_template (centred on the rest root joint), exact
//   shape blend     v_mfma_f32_32x32x2_f32, K = 10 -> 5 steps, exact f32 (per-frame beta supported)
//   pose blend      v_mfma_f32_32x32x16_bf16, K = 207 -> 13 steps, operands split hi+lo in bf16 and
//                   three products hi.hi + hi.lo + lo.hi (relative product error <= 2^-16), 5.3x the
//                   f32-MFMA rate; posedirs were pre-split and stored in B-fragment order at upload so
//                   every wave load is one contiguous 1 KiB (coalesced 16 B/lane)
// The accumulator layout puts the vertex on the lane and 16 frames in registers, so the skinning
// epilogue keeps each lane's 4 packed weights in registers, gathers the 3x4 transforms of its frames,
// and writes 384 contiguous bytes per frame row.  The blend result never touches HBM.
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void k_mesh_blend_lbs(DevModel M, DevProblem Pb, MeshCoef mc,
                                                         float* __restrict__ cloud) {
  const int vtile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, h = lane >> 5;
  const int v = vtile * kVTile + col;
  const int V = M.V, nJ = M.nJ, F = Pb.F;

  float vt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) vt[c] = M.vtB[((size_t)vtile * 3 + c) * 32 + col];
  const uint32_t widx = M.wIdx[(size_t)vtile * 32 + col];
  const float4 wv = reinterpret_cast<const float4*>(M.wVal)[(size_t)vtile * 32 + col];
  const float wgt[4] = {wv.x, wv.y, wv.z, wv.w};
  const uint4* dirs = reinterpret_cast<const uint4*>(M.dirsB);
  const uint4* feat = reinterpret_cast<const uint4*>(mc.featA);

  for (int ftile = wave; ftile < Pb.nFTiles; ftile += 4) {
    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = vt[c];

    // shape blend, exact f32
#pragma unroll
    for (int ks = 0; ks < kShapeKSteps; ++ks) {
      const float a = mc.betaA[((size_t)ftile * kShapeKSteps + ks) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float b = M.sdB[(((size_t)vtile * 3 + c) * kShapeKSteps + ks) * 64 + lane];
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      }
    }
    // pose blend, bf16 hi/lo split
    if (Pb.pose_blend && M.P > 0) {
#pragma unroll 1
      for (int ks = 0; ks < kPoseKSteps; ++ks) {
        const size_t fa = (((size_t)ftile * kPoseKSteps + ks) * 2) * 64 + lane;
        const bf16x8 ahi = __builtin_bit_cast(bf16x8, feat[fa]);
        const bf16x8 alo = __builtin_bit_cast(bf16x8, feat[fa + 64]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t fb = ((((size_t)vtile * 3 + c) * kPoseKSteps + ks) * 2) * 64 + lane;
          const bf16x8 bhi = __builtin_bit_cast(bf16x8, dirs[fb]);
          const bf16x8 blo = __builtin_bit_cast(bf16x8, dirs[fb + 64]);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[c], 0, 0, 0);
        }
      }
    }
    // skinning epilogue: lane = vertex, register = frame row (r&3) + 8 (r>>2) + 4 h
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = ftile * kFTile + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (f < F && v < V) {
        const float4* T = reinterpret_cast<const float4*>(mc.skinT + (size_t)f * nJ * 12);
        float4 t0 = make_float4(0, 0, 0, 0), t1 = t0, t2 = t0;
#pragma unroll
        for (int i = 0; i < kMeshNnz; ++i) {
          const int j = (widx >> (8 * i)) & 0xffu;
          const float w = wgt[i];
          const float4 a0 = T[j * 3 + 0], a1 = T[j * 3 + 1], a2 = T[j * 3 + 2];
          t0.x += w * a0.x; t0.y += w * a0.y; t0.z += w * a0.z; t0.w += w * a0.w;
          t1.x += w * a1.x; t1.y += w * a1.y; t1.z += w * a1.z; t1.w += w * a1.w;
          t2.x += w * a2.x; t2.y += w * a2.y; t2.z += w * a2.z; t2.w += w * a2.w;
        }
        const float px = acc[0][r], py = acc[1][r], pz = acc[2][r];
        float* o = cloud + ((size_t)f * V + v) * 3;
        o[0] = t0.x * px + t0.y * py + t0.z * pz + t0.w;
        o[1] = t1.x * px + t1.y * py + t1.z * pz + t1.w;
        o[2] = t2.x * px + t2.y * py + t2.z * pz + t2.w;
      }
    }
  }
}

}  // namespace

void launch_mesh(const DevModel& M, const DevProblem& P, const MeshCoef& mc, float* d_cloud, hipStream_t s) {
  if (P.F <= 0) return;
  hipLaunchKernelGGL(k_mesh_blend_lbs, dim3(M.nVTiles), dim3(256), 0, s, M, P, mc, d_cloud);
}

}  // namespace bodyfit
