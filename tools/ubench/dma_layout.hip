// Diagnostic: LDS image written by buffer_load ... lds (MUBUF LDS-DMA) for 4-, 12- and 16-byte pieces on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 -o dma_layout dma_layout.hip ; prints where word i of the source landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
#define KERNEL(SZ)                                                                                                   \
  __global__ void k##SZ(const unsigned* src, unsigned* out, int imm_test) {                                          \
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];                                                 \
    for (int i = threadIdx.x; i < 2048; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;                  \
    __syncthreads();                                                                                                 \
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 1 << 20, 0x00020000); \
    lds_u8* l = (lds_u8*)lds;                                                                                        \
    if (imm_test)                                                                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(l + 1024), SZ, threadIdx.x * SZ, 0, 1024, 0); \
    else                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(l + 1024), SZ, threadIdx.x * SZ, 0, 0, 0); \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
    __syncthreads();                                                                                                 \
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = reinterpret_cast<unsigned*>(lds)[i];                       \
  }
KERNEL(4)
KERNEL(12)
KERNEL(16)
template <typename K>
void run(K kern, int SZ, const unsigned* d_src, unsigned* d_out, int imm) {
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_src, d_out, imm);
  std::vector<unsigned> h(2048);
  hipMemcpy(h.data(), d_out, 8192, hipMemcpyDeviceToHost);
  printf("size %d imm %d: ", SZ, imm);
  int shown = 0;
  for (int i = 0; i < 2048 && shown < 40; ++i)
    if (h[i] != 0xdeadbeefu) { printf("[w%d]=%u ", i, h[i]); ++shown; }
  int n = 0; for (int i = 0; i < 2048; ++i) n += h[i] != 0xdeadbeefu;
  printf("... total %d words written\n", n);
}
int main() {
  std::vector<unsigned> s(1 << 18);
  for (size_t i = 0; i < s.size(); ++i) s[i] = (unsigned)i;
  unsigned *d_src, *d_out;
  hipMalloc(&d_src, s.size() * 4); hipMalloc(&d_out, 8192);
  hipMemcpy(d_src, s.data(), s.size() * 4, hipMemcpyHostToDevice);
  run(k4, 4, d_src, d_out, 0); run(k12, 12, d_src, d_out, 0); run(k16, 16, d_src, d_out, 0);
  run(k4, 4, d_src, d_out, 1); run(k16, 16, d_src, d_out, 1);
  return 0;
}
