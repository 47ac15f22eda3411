// Diagnostic: issue interval (s_memtime cycles per instruction) of ONE wave issuing independent f64 VALU instructions on gfx950:
// v_fma_f64, v_fmac_f64_dpp row_newbcast, v_mul_f64, v_readlane_b32 pairs + v_fma with an SGPR operand.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/valu_f64_rate tools/ubench/valu_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  double a[16], b = 1.0000001, c = 1e-9;
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x;
  unsigned long long t0, t1;
#define TIME(BODY, SLOT)                                                               \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
  for (int it = 0; it < iters; ++it) { BODY }                                          \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");            \
  if (threadIdx.x == 0) cyc[SLOT] = t1 - t0;
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
  TIME(REP16(FMA), 0)
#define DPP(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
  TIME(REP16(DPP), 1)
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  TIME(REP16(MUL), 2)
#define F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(((float*)a)[2 * i]) : "v"((float)b), "v"((float)c));
  TIME(REP16(F32), 3)
  // dependent chain: one accumulator
#define DEP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
  TIME(REP16(DEP), 4)
#define DEPDPP(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(b), "v"(c));
  TIME(REP16(DEPDPP), 5)
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[threadIdx.x] = s;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const char* names[6] = {"v_fma_f64 (16 independent)", "v_fmac_f64_dpp row_newbcast (16 independent)", "v_mul_f64 (16 independent)",
                          "v_fma_f32 (16 independent)", "v_fma_f64 dependent chain", "v_fmac_f64_dpp dependent chain"};
  for (int i = 0; i < 6; ++i) printf("%-48s %6.2f cycles per instruction\n", names[i], (double)h[i] / (16.0 * iters));
  return 0;
}
