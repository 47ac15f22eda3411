// dense_inl.h — small dense f64 building blocks shared by the device solvers (k_lm_batched.hip, k_window_lm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace bodyfit {
namespace {

// f64 value of lane `src` (wave-uniform lane id): two v_readlane_b32
__device__ __forceinline__ double readlane_f64w(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Cholesky of a 16 x 16 diagonal block in the registers of ONE wave, right-looking, with rows appended below that receive
// L^-T from the right at no extra instruction: lane = row (lane & 15) of group (lane >> 4); group 0 holds the block's own
// rows, the other groups whatever rows the caller appends (here: the identity, which comes out as L^-T, so the panel solve
// becomes a product on the matrix cores).  Column values travel by v_readlane (the source lane is uniform).
// Software-pipelined by hand: pivot c + 1 is final after the FIRST update of step c, so its broadcast and its reciprocal
// square root (hardware estimate + two Newton steps: the long dependent chain of a step) are issued there and run under the
// remaining updates of step c.  inv_out[c] (lane-uniform) = 1 / L_cc.
// nvalid: pivots to process; the columns beyond are identity padding (unit pivots, zero couplings) and are passed through.
__device__ __forceinline__ bool diag_factor16(double (&av)[16], int rr, bool own_rows, double (&inv_out)[16], int nvalid = 16) {
  auto rsq_nr = [](double piv) {
    double inv = __builtin_amdgcn_rsq(piv);
    inv = inv * (1.5 - 0.5 * piv * inv * inv);
    return inv * (1.5 - 0.5 * piv * inv * inv);
  };
  bool okp = true;
  double piv = readlane_f64w(av[0], 0);
  double inv = rsq_nr(piv);
#pragma unroll
  for (int c = 0; c < 16; ++c) inv_out[c] = 1.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < nvalid) {   // (uniform)
      if (!(piv > 0.0) || !(piv < 1e300)) okp = false;
      inv_out[c] = inv;
      const double l = (own_rows && rr == c) ? piv * inv : av[c] * inv;
      av[c] = l;
      double piv_n = 1.0, inv_n = 1.0;
      if (c + 1 < 16) {
        av[c + 1] -= l * readlane_f64w(l, c + 1);
        piv_n = readlane_f64w(av[c + 1], c + 1);
        inv_n = rsq_nr(piv_n);
      }
      // (no row predicate: above the diagonal this writes values nothing reads)
#pragma unroll
      for (int k = c + 2; k < 16; ++k) av[k] -= l * readlane_f64w(l, k);
      piv = piv_n; inv = inv_n;
    }
  }
  return okp;
}

}  // namespace
}  // namespace bodyfit
