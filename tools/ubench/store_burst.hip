// Microbenchmark: how long does the mesh kernel's output store pattern take on its own?
// 216 workgroups x 8 waves, each wave stores 16 rows of 2 x 384 B (12 B per lane) into a [F][6912][3] f32 cloud,
// F = 256 (21 MB).  Variants: dwordx3 / dwordx4-contiguous, plain / sc1 / nt; optional dependent-load chain.
// build: hipcc -O3 --offload-arch=gfx950 store_burst.hip -o store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int AUX, bool X4, bool TILE_MAJOR = false>
__global__ __launch_bounds__(512) void k_store(float* cloud, int F, int nVT, unsigned long long* stamps, int spin) {
  const int vtile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned stride = (unsigned)nVT * 32 * 12;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cloud, 0, (int)((unsigned)F * stride), 0x00020000);
  unsigned long long t0 = 0, t1 = 0;
  if (lane == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int ftile = wave; ftile * 32 < F; ftile += 8) {
    for (int r = 0; r < 16; ++r) {
      const int f = ftile * 32 + (r >> 2) * 8 + (r & 3);
      float acc = (float)lane;
      for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;     // stand-in for the row's VALU work
      if (X4) {
        const unsigned off = (unsigned)f * stride / 12 * 16 / 4 * 0 + ((unsigned)(f * nVT + vtile) * 64 + lane) * 16;
        u32x4 v = {__float_as_uint(acc), 1u, 2u, 3u};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off % ((unsigned)F * stride - 16), 0, AUX);
      } else {
        u32x3 v = {__float_as_uint(acc), 1u, 2u};
        if (TILE_MAJOR) {   // cloud as [vtile][frame][32][3]: a workgroup's stores cover one contiguous region
          const unsigned off = ((unsigned)vtile * (unsigned)F + (unsigned)f + 4 * (lane >> 5)) * 384 + (lane & 31) * 12;
          __builtin_amdgcn_raw_buffer_store_b96(v, rs, off, 0, AUX);
        } else {
          const unsigned off = 4 * (lane >> 5) * stride + (unsigned)(vtile * 32 + (lane & 31)) * 12;
          __builtin_amdgcn_raw_buffer_store_b96(v, rs, off, (unsigned)f * stride, AUX);
        }
      }
    }
  }
  if (lane == 0) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    stamps[(blockIdx.x * 8 + wave) * 2] = t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t1;
  }
}

template <int AUX, bool X4, bool TM = false>
void run(const char* name, float* d, int F, int nVT, unsigned long long* ds, int spin) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<AUX, X4, TM>), dim3(nVT), dim3(512), 0, 0, d, F, nVT, ds, spin);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_store<AUX, X4, TM>), dim3(nVT), dim3(512), 0, 0, d, F, nVT, ds, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nVT * 8 * 2);
  hipMemcpy(h.data(), ds, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0; double wsum = 0; int n = 0;
  for (int i = 0; i < nVT * 8; ++i) { if (F > i % 8 * 32) { lo = lo < h[2*i] ? lo : h[2*i]; hi = hi > h[2*i+1] ? hi : h[2*i+1]; wsum += (h[2*i+1]-h[2*i]) / 100.0; ++n; } }
  const double mb = (double)F * nVT * 32 * 12 / 1e6;
  printf("%-28s spin %4d: %.2f us per launch (events), in-kernel span %.2f us, mean wave %.2f us, %.1f MB -> %.2f TB/s over the span\n",
         name, spin, ms / 20 * 1e3, (hi - lo) / 100.0, wsum / n, mb, mb / ((hi - lo) / 100.0) / 1e6 * 1e0);
}

int main() {
  const int F = 256, nVT = 216;
  float* d; hipMalloc(&d, (size_t)F * nVT * 32 * 12 * 2);
  unsigned long long* ds; hipMalloc(&ds, nVT * 8 * 2 * 8);
  for (int spin : {0, 20}) {
    run<0, false>("dwordx3 plain", d, F, nVT, ds, spin);
    run<16, false>("dwordx3 sc1", d, F, nVT, ds, spin);
    run<2, false>("dwordx3 nt", d, F, nVT, ds, spin);
    run<0, true>("dwordx4 contiguous plain", d, F, nVT, ds, spin);
    run<16, true>("dwordx4 contiguous sc1", d, F, nVT, ds, spin);
    run<0, false, true>("dwordx3 tile-major plain", d, F, nVT, ds, spin);
    run<16, false, true>("dwordx3 tile-major sc1", d, F, nVT, ds, spin);
  }
  return 0;
}
