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


// ---- the same factorisation with the column values travelling by DPP (row_newbcast: lane k of every 16-lane row to the whole
//      row) instead of v_readlane: one v_fmac_f64_dpp per (pivot, column) where the readlane form needs two v_readlane_b32 and
//      a v_fma_f64 (and SGPR-hazard s_nops).  A row broadcast stays inside its 16 lanes, so the appended row of a lane is a
//      second register set bv of the SAME lane (the caller passes the identity there and gets L^-T back); every 16-lane group
//      of the wave runs the same code on its own data (the callers load the same block into all four).
//      The whole pivot step is a sequence of one-instruction asm statements in issue order: the reciprocal square root of
//      the NEXT pivot (hardware estimate + two Newton steps, the dependent chain of a step) is interleaved by hand with the
//      rank-1 updates of the current one.  hipcc does not see into asm statements, so the two hazards of gfx940+ that apply
//      are kept by construction: a VGPR written by a VALU instruction is read through DPP two or more instructions later, and
//      the result of a transcendental is read by an ordinary VALU instruction one or more instructions later.
namespace dpp16 {
template <int K>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& col, const double& mul) {   // acc += col[lane K of the row] * mul
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(col), "v"(mul), "n"(K));
}
template <int K>
__device__ __forceinline__ void mov_bcast(double& dst, const double& src) {
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src), "n"(K));
}
__device__ __forceinline__ void mul(double& d, const double& a, const double& b) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void neg(double& d, const double& a) { asm volatile("v_mul_f64 %0, -1.0, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void half(double& d, const double& a) { asm volatile("v_mul_f64 %0, 0.5, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void rsq(double& d, const double& a) { asm volatile("v_rsq_f64 %0, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void nr(double& e, const double& r, const double& c15) {   // e = 1.5 - r * e
  asm volatile("v_fma_f64 %0, -%1, %0, %2" : "+v"(e) : "v"(r), "v"(c15));
}
__device__ __forceinline__ void nop2() { asm volatile("s_nop 1"); }

// rank-1 updates number I .. I + N - 1 of pivot C: update u covers column C + 2 + u / 2 of av (u even) or bv (u odd);
// (column C + 1 of both sets and column C + 2 of av are issued by the caller, around the next pivot's broadcast)
template <int C, int U>
__device__ __forceinline__ void update_one(double (&av)[16], double (&bv)[16], const double& nl, const double& nlb) {
  constexpr int u = U + 1;                       // u = 0 (av[C + 2]) belongs to the caller
  constexpr int k = C + 2 + u / 2;
  if constexpr (k < 16) {
    if constexpr (u % 2 == 0) fmac_bcast<k>(av[k], av[C], nl);
    else fmac_bcast<k>(bv[k], av[C], nlb);
  }
}
template <int C>
__device__ __forceinline__ void pivot_step(double (&av)[16], double (&bv)[16], const double& inv, double& piv_n, double& inv_n,
                                           const double& c15) {
  double nl, nlb, r, h, e;
  mul(av[C], av[C], inv);          // l   (the pivot's own lane: piv / sqrt(piv))
  mul(bv[C], bv[C], inv);          // l of the appended row
  neg(nl, av[C]);
  neg(nlb, bv[C]);
  if constexpr (C + 1 < 16) {
    fmac_bcast<C + 1>(av[C + 1], av[C], nl);            // (av[C] was written three instructions ago)
    fmac_bcast<C + 1>(bv[C + 1], av[C], nlb);
    if constexpr (C + 2 < 16) fmac_bcast<C + 2>(av[C + 2], av[C], nl); else nop2();
    mov_bcast<C + 1>(piv_n, av[C + 1]);                 // the next pivot (av[C + 1] was written two instructions ago)
    rsq(r, piv_n);
    update_one<C, 0>(av, bv, nl, nlb);
    half(h, piv_n);
    update_one<C, 1>(av, bv, nl, nlb); update_one<C, 2>(av, bv, nl, nlb);
    mul(e, h, r);                                       // (r: one or more instructions after the transcendental)
    update_one<C, 3>(av, bv, nl, nlb); update_one<C, 4>(av, bv, nl, nlb);
    nr(e, r, c15);
    update_one<C, 5>(av, bv, nl, nlb); update_one<C, 6>(av, bv, nl, nlb);
    mul(r, r, e);
    update_one<C, 7>(av, bv, nl, nlb); update_one<C, 8>(av, bv, nl, nlb);
    mul(e, h, r);
    update_one<C, 9>(av, bv, nl, nlb); update_one<C, 10>(av, bv, nl, nlb);
    nr(e, r, c15);
    update_one<C, 11>(av, bv, nl, nlb); update_one<C, 12>(av, bv, nl, nlb);
    mul(inv_n, r, e);
    update_one<C, 13>(av, bv, nl, nlb); update_one<C, 14>(av, bv, nl, nlb); update_one<C, 15>(av, bv, nl, nlb);
    update_one<C, 16>(av, bv, nl, nlb); update_one<C, 17>(av, bv, nl, nlb); update_one<C, 18>(av, bv, nl, nlb);
    update_one<C, 19>(av, bv, nl, nlb); update_one<C, 20>(av, bv, nl, nlb); update_one<C, 21>(av, bv, nl, nlb);
    update_one<C, 22>(av, bv, nl, nlb); update_one<C, 23>(av, bv, nl, nlb); update_one<C, 24>(av, bv, nl, nlb);
    update_one<C, 25>(av, bv, nl, nlb); update_one<C, 26>(av, bv, nl, nlb);
  }
}
template <int C>
__device__ __forceinline__ void steps_from(double (&av)[16], double (&bv)[16], double (&inv_out)[16], int nvalid, double piv, double inv,
                                           const double& c15, bool& okp) {
  if constexpr (C < 16) {
    if (C < nvalid) {   // (uniform)
      if (!(piv > 0.0) || !(piv < 1e300)) okp = false;
      inv_out[C] = inv;
      double piv_n = 1.0, inv_n = 1.0;
      pivot_step<C>(av, bv, inv, piv_n, inv_n, c15);
      steps_from<C + 1>(av, bv, inv_out, nvalid, piv_n, inv_n, c15, okp);
    }
  }
}
}  // namespace dpp16

// av: row (lane & 15) of the block; bv: the appended row of the same lane.  On return av holds the row of L (lower part),
// bv the appended row times L^-T, inv_out[c] = 1 / L_cc (row-uniform).  Pivots >= nvalid are identity padding.
__device__ __forceinline__ bool diag_factor16_dpp(double (&av)[16], double (&bv)[16], double (&inv_out)[16], int nvalid = 16) {
  const double c15 = 1.5;
  bool okp = true;
#pragma unroll
  for (int c = 0; c < 16; ++c) inv_out[c] = 1.0;
  double piv, r, h, e, inv;
  dpp16::nop2();                                   // (av[0] may have been written by the instruction before)
  dpp16::mov_bcast<0>(piv, av[0]);
  dpp16::rsq(r, piv);
  dpp16::half(h, piv);
  dpp16::mul(e, h, r);
  dpp16::nr(e, r, c15);
  dpp16::mul(r, r, e);
  dpp16::mul(e, h, r);
  dpp16::nr(e, r, c15);
  dpp16::mul(inv, r, e);
  dpp16::steps_from<0>(av, bv, inv_out, nvalid, piv, inv, c15, okp);
  return okp;
}

}  // namespace
}  // namespace bodyfit
