// Microbenchmark: do the mesh kernel's output stores interfere with its LDS gather?
// 216 workgroups x 8 waves; per wave 16 rows: 12 ds_read_b128 gathers (per-lane offsets into a 9 KiB slice), packed-f32
// work, one 768-B buffer store (real address / one shared dump row / no store).  Reports the in-kernel span.
// build: hipcc -O3 --offload-arch=gfx950 store_lds.hip -o store_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;

template <int MODE>   // 0 no store, 1 dump row, 2 real addresses
__global__ __launch_bounds__(512) void k(float* cloud, int F, int nVT, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int vtile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* slice = lds + wave * 9216;
  for (int i = lane; i < 9216 / 4; i += 64) reinterpret_cast<float*>(slice)[i] = 0.001f * i;
  __syncthreads();
  const unsigned stride = (unsigned)nVT * 32 * 12;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cloud, 0, (int)((unsigned)F * stride), 0x00020000);
  const unsigned char* tj[4];
  for (int i = 0; i < 4; ++i) tj[i] = slice + 4 * (lane >> 5) * 1152 + ((lane * 7 + i * 5) % 24) * 48;
  unsigned long long t0 = 0, t1 = 0;
  if (lane == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float acc0 = lane, acc1 = 1.0f, acc2 = 2.0f;
  for (int ftile = wave; ftile * 32 < F; ftile += 8) {
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
      float4 t[12];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4* T = reinterpret_cast<const float4*>(tj[i] + (r & 3) * 1152);
        t[3 * i] = T[0]; t[3 * i + 1] = T[1]; t[3 * i + 2] = T[2];
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        acc0 = acc0 * 0.5f + t[i].x * t[i].y; acc1 = acc1 * 0.5f + t[i].z * t[i].w; acc2 = acc2 * 0.5f + t[i].x * t[i].w;
      }
      const int f = ftile * 32 + (r >> 2) * 8 + (r & 3);
      u32x3 v = {__float_as_uint(acc0), __float_as_uint(acc1), __float_as_uint(acc2)};
      if (MODE == 2) {
        const unsigned off = 4 * (lane >> 5) * stride + (unsigned)(vtile * 32 + (lane & 31)) * 12;
        __builtin_amdgcn_raw_buffer_store_b96(v, rs, off, (unsigned)f * stride, 16);
      } else if (MODE == 1) {
        __builtin_amdgcn_raw_buffer_store_b96(v, rs, (unsigned)lane * 12, 0, 0);
      } else if (acc0 == 12345.678f) {
        __builtin_amdgcn_raw_buffer_store_b96(v, rs, (unsigned)lane * 12, 0, 0);
      }
    }
  }
  if (lane == 0) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    stamps[(blockIdx.x * 8 + wave) * 2] = t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t1;
  }
}

template <int MODE>
void run(const char* name, float* d, int F, int nVT, unsigned long long* ds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 9216);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), dim3(nVT), dim3(512), 8 * 9216, 0, d, F, nVT, ds);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(nVT * 8 * 2);
  (void)hipMemcpy(h.data(), ds, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0; double wsum = 0;
  for (int i = 0; i < nVT * 8; ++i) { lo = lo < h[2*i] ? lo : h[2*i]; hi = hi > h[2*i+1] ? hi : h[2*i+1]; wsum += (h[2*i+1]-h[2*i]) / 100.0; }
  printf("%-22s in-kernel span %.2f us, mean wave %.2f us\n", name, (hi - lo) / 100.0, wsum / (nVT * 8));
}

int main() {
  const int F = 256, nVT = 216;
  float* d; (void)hipMalloc(&d, (size_t)F * nVT * 32 * 12 * 2);
  unsigned long long* ds; (void)hipMalloc(&ds, nVT * 8 * 2 * 8);
  run<0>("gather, no store", d, F, nVT, ds);
  run<1>("gather + dump store", d, F, nVT, ds);
  run<2>("gather + real store", d, F, nVT, ds);
  return 0;
}
