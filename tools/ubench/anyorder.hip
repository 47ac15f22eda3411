// Does hipExtAnyOrderLaunch let a kernel start before its predecessor in the same stream has finished?
// Two kernels of 128 workgroups each that spin for ~20 us; serial = ~40 us + boundary, overlapped = ~20 us.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 20; ++i) {
        hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s, nullptr, nullptr, 0, 2000ull, d);
        hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s, nullptr, nullptr, flag, 2000ull, d + 1);
      }
      hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("flag %d: %.1f us per pair\n", flag, us / 20);
    }
  }
  return 0;
}
