// micro-benchmark: v_mfma_f64_16x16x4_f64 issue interval / dependent latency on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) double d4;
template <int NACC>
__global__ void k(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (double)(iters * NACC);
}
int main() {
  double* d; hipMalloc(&d, 1024); double h[65];
  k<1><<<1, 64>>>(d, 1000); hipMemcpy(h, d, 65 * 8, hipMemcpyDeviceToHost); printf("1 accumulator  (dependent): %.1f cycles/MFMA\n", h[64]);
  k<5><<<1, 64>>>(d, 1000); hipMemcpy(h, d, 65 * 8, hipMemcpyDeviceToHost); printf("5 accumulators (independent): %.1f cycles/MFMA\n", h[64]);
  k<8><<<1, 64>>>(d, 1000); hipMemcpy(h, d, 65 * 8, hipMemcpyDeviceToHost); printf("8 accumulators (independent): %.1f cycles/MFMA\n", h[64]);
  return 0;
}
