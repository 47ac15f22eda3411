// Diagnostic: where one k_cr_factor workgroup (the window LM's 80 x 80 block factorisation with 96 appended rows) spends its
// cycles.  Includes the product's translation unit with -DBODYFIT_CR_STAMPS (wave 0 writes s_memtime per phase).
// Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBODYFIT_CR_STAMPS -I3dbodyanimation_amd/csrc -o tools/ubench/cr_factor_phases tools/ubench/cr_factor_phases.hip
#include "../../3dbodyanimation_amd/csrc/k_window_lm.hip"

#include <cstdio>
#include <cstring>
#include <vector>
namespace bodyfit { std::atomic<long> g_launch_count{0}; }
int main() {
  using namespace bodyfit;
  const int F = 3, WBk = kWinBlock, WRk = kWinRhs;
  WinBuf W{};
  std::vector<double> D((size_t)F * WBk * WBk, 0.0), U((size_t)F * WBk * WBk, 0.0), R((size_t)F * WRk * WBk, 0.0);
  for (int f = 0; f < F; ++f)
    for (int i = 0; i < WBk; ++i)
      for (int k = 0; k < WBk; ++k) {
        D[((size_t)f * WBk + i) * WBk + k] = (i == k ? 100.0 : 0.0) + 1.0 / (1.0 + i + k);
        U[((size_t)f * WBk + i) * WBk + k] = 0.01 * ((i * 7 + k * 3) % 11) - 0.05;
      }
  for (auto& v : R) v = 0.5;
  auto up = [](const std::vector<double>& h) { double* d; hipMalloc(&d, h.size() * 8); hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); return d; };
  W.D = up(D); W.U = up(U); W.Rt = up(R);
  hipMalloc(&W.L, D.size() * 8); hipMalloc(&W.Pt, D.size() * 8); hipMalloc(&W.Qt, D.size() * 8); hipMalloc(&W.Yt, R.size() * 8);
  hipMalloc(&W.Li, (size_t)F * 5 * 256 * 8); hipMalloc(&W.fail, 4); hipMemset(W.fail, 0, 4);
  const int elim_h[3] = {1, 0, 2};
  int* d_elim; hipMalloc(&d_elim, 12); hipMemcpy(d_elim, elim_h, 12, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) launch_cr_factor(W, d_elim, 1, 0);
  hipDeviceSynchronize();
  unsigned long long st[64];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cr_stamps), sizeof(st));
  int fail = 0; hipMemcpy(&fail, W.fail, 4, hipMemcpyDeviceToHost);
  printf("fail %d\nload blocks -> LDS           %6llu cycles\ndiagonal block 0              %6llu\n", fail, st[1] - st[0], st[2] - st[1]);
  unsigned long long prev = st[2];
  for (int p = 0; p < 5; ++p) {
    printf("panel %d: solve %6llu", p, st[3 + 4 * p] - prev);
    if (p < 4) printf(" | wave 0: next diagonal tile update %6llu, its factorisation %6llu | whole update phase %6llu\n",
                      st[4 + 4 * p] - st[3 + 4 * p], st[5 + 4 * p] - st[4 + 4 * p], st[6 + 4 * p] - st[3 + 4 * p]);
    else printf(" | (last panel) %6llu\n", st[6 + 4 * p] - st[3 + 4 * p]);
    prev = st[6 + 4 * p];
  }
  printf("store blocks                  %6llu\ntotal                         %6llu cycles\n", st[24] - st[22], st[24] - st[0]);
  // the outputs of frame 1, as a checksum that is sensitive to every bit (compare two builds of the kernel)
  auto sum = [&](const double* d, size_t off, size_t n) {
    std::vector<double> h(n);
    hipMemcpy(h.data(), d + off, n * 8, hipMemcpyDeviceToHost);
    unsigned long long x = 1469598103934665603ull;
    for (double v : h) { unsigned long long b; memcpy(&b, &v, 8); x = (x ^ b) * 1099511628211ull; }
    return x;
  };
  {   // launch to launch, 200 launches (what the stamps cannot see: the time in front of the first stamp)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) launch_cr_factor(W, d_elim, 1, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("launch to launch: %.2f us per k_cr_factor\n", ms / 200 * 1e3);
    }
  }
  const size_t bb = (size_t)WBk * WBk;
  printf("outputs: L %016llx  Pt %016llx  Qt %016llx  Yt %016llx  Li %016llx\n", sum(W.L, bb, bb), sum(W.Pt, bb, bb), sum(W.Qt, bb, bb),
         sum(W.Yt, (size_t)WRk * WBk, (size_t)WRk * WBk), sum(W.Li, 5 * 256, 5 * 256));
  return 0;
}
