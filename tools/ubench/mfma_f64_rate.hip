// Issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on one SIMD (one wave, independent accumulator chains), in shader
// cycles per instruction (s_memtime).  hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) double d4;
__global__ void k(double* out, unsigned long long* t, int n) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  unsigned long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  asm volatile("s_nop 0" ::"v"(a0[0]), "v"(a1[0]), "v"(a2[0]), "v"(a3[0]));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double f0 = x, f1 = y, f2 = x + 1, f3 = y + 1, f4 = x + 2, f5 = y + 2, f6 = x + 3, f7 = y + 3;
  for (int i = 0; i < n; ++i) {
    f0 = __builtin_fma(f0, y, x); f1 = __builtin_fma(f1, y, x); f2 = __builtin_fma(f2, y, x); f3 = __builtin_fma(f3, y, x);
    f4 = __builtin_fma(f4, y, x); f5 = __builtin_fma(f5, y, x); f6 = __builtin_fma(f6, y, x); f7 = __builtin_fma(f7, y, x);
  }
  asm volatile("s_nop 0" ::"v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
}
int main() {
  double* o; unsigned long long* t;
  hipMalloc(&o, 64 * 8); hipMalloc(&t, 16);
  const int n = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, n);
  unsigned long long h[2];
  hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  std::printf("v_mfma_f64_16x16x4_f64: %.1f cycles each (2048 flop)   v_fma_f64: %.1f cycles each (128 flop)\n", (double)h[0] / (4.0 * n),
              (double)h[1] / (8.0 * n));
  return 0;
}
