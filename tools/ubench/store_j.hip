// Microbenchmark: the Jacobian panel's store pattern on its own.  256 workgroups x 512 threads, one frame's [50][86] f64 panel
// (34,400 B, contiguous per frame) each, written the way the frame role's sweep writes it and as 16-byte column pairs.
//   x2   thread = (joint column kc of 69, keypoint group g of 7): two 8-byte stores per keypoint (rows 2k, 2k + 1), then the
//        17 Sim3 + shape columns by a flat loop, 8 bytes per lane                               (frame_part_inl.h, phase F)
//   x4   thread = (column pair pc of 43, keypoint group g of 11): two 16-byte stores per keypoint
// each plain / sc1 (write-through).  build: hipcc -O3 --offload-arch=gfx950 store_j.hip -o store_j
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) double d2;
template <int SC1> __device__ __forceinline__ void st8(double* p, double v) {
  if (SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); else *p = v;
}
template <int SC1> __device__ __forceinline__ void st16(double* p, d2 v) {
  if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); else *reinterpret_cast<d2*>(p) = v;
}
template <int SC1, bool X4>
__global__ __launch_bounds__(512) void k_store(double* J, int spin) {
  const int tid = threadIdx.x;
  double* Jf = J + (size_t)blockIdx.x * 50 * 86;
  double x = (double)tid;
  for (int i = 0; i < spin; ++i) x = x * 1.0001 + 0.5;
  if (!X4) {
    if (tid < 7 * 69) {
      const int g = tid / 69, kc = tid - g * 69;
      for (int kk = g; kk < 25; kk += 7) {
        st8<SC1>(Jf + (size_t)(2 * kk) * 86 + 7 + kc, x);
        st8<SC1>(Jf + (size_t)(2 * kk + 1) * 86 + 7 + kc, x + 1.0);
      }
    }
    for (int e = tid; e < 25 * 17; e += 512) {
      const int kk = e / 17, c = e - kk * 17, col = c < 7 ? c : 76 + (c - 7);
      st8<SC1>(Jf + (size_t)(2 * kk) * 86 + col, x);
      st8<SC1>(Jf + (size_t)(2 * kk + 1) * 86 + col, x + 1.0);
    }
  } else {
    if (tid < 11 * 43) {
      const int g = tid / 43, pc = tid - g * 43;
      for (int kk = g; kk < 25; kk += 11) {
        st16<SC1>(Jf + (size_t)(2 * kk) * 86 + 2 * pc, d2{x, x + 2.0});
        st16<SC1>(Jf + (size_t)(2 * kk + 1) * 86 + 2 * pc, d2{x + 1.0, x + 3.0});
      }
    }
  }
}
template <int SC1, bool X4> void run(const char* name, double* d, int spin) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<SC1, X4>), dim3(256), dim3(512), 0, 0, d, spin);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_store<SC1, X4>), dim3(256), dim3(512), 0, 0, d, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-26s %.2f us per launch (8.8 MB; launch-to-launch, includes the ~1.5 us boundary)\n", name, ms / 50 * 1e3);
}
int main() {
  double* d; hipMalloc(&d, (size_t)256 * 50 * 86 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, false>("x2 plain", d, 0); run<1, false>("x2 sc1", d, 0); run<0, true>("x4 pairs plain", d, 0); run<1, true>("x4 pairs sc1", d, 0);
  }
  return 0;
}
