// Micro-benchmark and bit-for-bit check of the 16 x 16 diagonal-block factorisation (dense_inl.h diag_factor16_acc): the shipped
// form (round 5: acc16b) against round 4's (diag_factor16_acc_r4, kept in dense_inl.h as the reference).
// Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -I3dbodyanimation_amd/csrc -o tools/ubench/diag16 tools/ubench/diag16.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../3dbodyanimation_amd/csrc/dense_inl.h"

// one wave per block; A [16][16] symmetric positive definite, row-major; out: a[4], b[4] per lane, inv_col, ok; cycles
template <int V>
__global__ __launch_bounds__(64) void k_diag(const double* __restrict__ A, int nvalid, int reps, double* __restrict__ out,
                                             unsigned long long* __restrict__ cyc) {
  const int lane = threadIdx.x, m = lane & 15, kk = lane >> 4;
  const double* Ab = A + (size_t)blockIdx.x * 256;
  const double s0 = Ab[(kk + 0) * 16 + m], s1 = Ab[(kk + 4) * 16 + m], s2 = Ab[(kk + 8) * 16 + m], s3 = Ab[(kk + 12) * 16 + m];
  double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, inv = 0.0;
  bool ok = true;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    double a[4] = {s0, s1, s2, s3};
    double b[4] = {kk == m ? 1.0 : 0.0, kk + 4 == m ? 1.0 : 0.0, kk + 8 == m ? 1.0 : 0.0, kk + 12 == m ? 1.0 : 0.0};
    if constexpr (V == 1) ok = bodyfit::diag_factor16_acc_r4(a, b, lane, inv, nvalid);
    else ok = bodyfit::diag_factor16_acc(a, b, lane, inv, nvalid);
    r0 = a[0]; r1 = a[1]; r2 = a[2]; r3 = a[3]; r4 = b[0]; r5 = b[1]; r6 = b[2]; r7 = b[3];
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double* o = out + (size_t)blockIdx.x * (64 * 9 + 1);
  o[lane * 9 + 0] = r0; o[lane * 9 + 1] = r1; o[lane * 9 + 2] = r2; o[lane * 9 + 3] = r3;
  o[lane * 9 + 4] = r4; o[lane * 9 + 5] = r5; o[lane * 9 + 6] = r6; o[lane * 9 + 7] = r7;
  o[lane * 9 + 8] = inv;
  if (lane == 0) { o[64 * 9] = ok ? 1.0 : 0.0; cyc[blockIdx.x] = t1 - t0; }
}

int main() {
  const int NB = 64;
  std::vector<double> A((size_t)NB * 256);
  unsigned long long sd = 12345;
  auto u = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return (double)(sd >> 11) / 9007199254740992.0 - 0.5; };
  for (int b = 0; b < NB; ++b) {
    double G[16][16];
    for (auto& row : G) for (double& v : row) v = u();
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = (i == j) ? 0.5 + (b % 7) * 0.3 : 0.0;
        for (int k = 0; k < 16; ++k) s += G[i][k] * G[j][k];
        A[(size_t)b * 256 + i * 16 + j] = s;
      }
  }
  double *dA, *d1, *d2; unsigned long long *c1, *c2;
  const size_t no = (size_t)NB * (64 * 9 + 1);
  hipMalloc(&dA, A.size() * 8); hipMalloc(&d1, no * 8); hipMalloc(&d2, no * 8); hipMalloc(&c1, NB * 8); hipMalloc(&c2, NB * 8);
  std::vector<double> o1(no), o2(no);
  std::vector<unsigned long long> h1(NB), h2(NB);
  int bad_total = 0;
  for (int nvalid : {16, 12, 6, 1}) {
    std::vector<double> Ap = A;   // identity padding beyond nvalid
    for (int b = 0; b < NB; ++b)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
          if (i >= nvalid || j >= nvalid) Ap[(size_t)b * 256 + i * 16 + j] = (i == j) ? 1.0 : 0.0;
    hipMemcpy(dA, Ap.data(), Ap.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_diag<1>, dim3(NB), dim3(64), 0, 0, dA, nvalid, 1, d1, c1);
    hipLaunchKernelGGL(k_diag<2>, dim3(NB), dim3(64), 0, 0, dA, nvalid, 1, d2, c2);
    hipDeviceSynchronize();
    hipMemcpy(o1.data(), d1, no * 8, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, no * 8, hipMemcpyDeviceToHost);
    int bad = 0; double worst_vs_host = 0.0;
    for (int b = 0; b < NB; ++b) {
      const double* p1 = o1.data() + (size_t)b * (64 * 9 + 1); const double* p2 = o2.data() + (size_t)b * (64 * 9 + 1);
      // host Cholesky of the block
      double L[16][16] = {};
      for (int j = 0; j < 16; ++j) {
        double d = Ap[(size_t)b * 256 + j * 16 + j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < 16; ++i) { double s = Ap[(size_t)b * 256 + i * 16 + j]; for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k]; L[i][j] = s / L[j][j]; }
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, kk = lane >> 4;
        for (int q = 0; q < 4; ++q) {
          const int r = kk + 4 * q;
          if (r >= m) {   // L
            if (std::memcmp(&p1[lane * 9 + q], &p2[lane * 9 + q], 8)) ++bad;
            worst_vs_host = std::fmax(worst_vs_host, std::fabs(p2[lane * 9 + q] - L[r][m]));
          }
          if (m >= r && std::memcmp(&p1[lane * 9 + 4 + q], &p2[lane * 9 + 4 + q], 8)) ++bad;   // appended rows x L^-T
        }
        if (std::memcmp(&p1[lane * 9 + 8], &p2[lane * 9 + 8], 8)) ++bad;
      }
      if (p1[64 * 9] != p2[64 * 9]) ++bad;
    }
    printf("nvalid %2d: words that differ between round 4's form and round 5's: %d; |L - host Cholesky| max %.2e\n", nvalid, bad, worst_vs_host);
    bad_total += bad;
  }
  // timing: one wave per CU-ish (64 blocks), 200 factorisations each
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_diag<1>, dim3(NB), dim3(64), 0, 0, dA, 16, 200, d1, c1);
    hipLaunchKernelGGL(k_diag<2>, dim3(NB), dim3(64), 0, 0, dA, 16, 200, d2, c2);
    hipDeviceSynchronize();
    hipMemcpy(h1.data(), c1, NB * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), c2, NB * 8, hipMemcpyDeviceToHost);
    double m1 = 0, m2 = 0;
    for (int b = 0; b < NB; ++b) { m1 += h1[b]; m2 += h2[b]; }
    printf("cycles per 16 x 16 block (s_memtime, mean over %d waves, 200 blocks each): round 4 %.0f = %.0f per pivot | round 5 %.0f = %.0f per pivot\n",
           NB, m1 / NB / 200, m1 / NB / 200 / 16, m2 / NB / 200, m2 / NB / 200 / 16);
  }
  return bad_total ? 1 : 0;
}
