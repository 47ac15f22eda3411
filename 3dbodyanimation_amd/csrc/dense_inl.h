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


// ---- the same factorisation with the block spread over ALL 64 lanes (round 4) -------------------------------------------------
// diag_factor16_dpp keeps one row per lane: a rank-1 update of the block is one instruction per COLUMN, sixteen useful lanes
// each, 31 f64 instructions per pivot at the start of a block; measured 330-416 cycles per pivot (tools/ubench/
// cr_factor_phases.hip), 42 % of a window-LM iteration and 72 % of the batched LM.  Here the block lives in the layout of an
// f64 16 x 16 MFMA accumulator, four registers per lane:
//     a[q], lane (m = lane & 15, kk = lane >> 4)  =  A[kk + 4 q][m]          (the FULL symmetric block, both triangles)
//     b[q]                                         =  the appended rows (identity -> L^-T), same layout
// A rank-1 update  A[r][m] -= A[r][c] A[c][m] / piv  of the WHOLE block is then four instructions (one per q), sixty-four
// useful lanes each:   a[q] += row_newbcast:c(a[q]) * w,   w[m] = -A[c][m] / piv for m > c, 0 otherwise
// — the broadcast inside a 16-lane row hands every column m the row's entry of column c, and w is row c of A (by symmetry its
// entries ARE column c), copied from the row of lanes it lives in to all four with two lane-swap instructions (gfx950:
// v_permlane16_swap / v_permlane32_swap).  Eight updates per pivot (A and the appended rows) whatever the pivot, and NO scaling
// inside the loop: the columns stay unscaled (A~[r][m] = L[r][m] L[m][m]; the appended rows likewise, see the derivation in
// DESIGN.md 6 round 4) and are multiplied by 1 / L[m][m] = rsqrt(A~[m][m]) once at the end.  Per pivot: 8 f64 FMAs through DPP,
// a reciprocal with two Newton steps, one product, ~10 cheap 32-bit moves / swaps / selects.
// On return a[q] holds L (valid for rows >= column), b[q] the appended rows times L^-T (valid for columns >= row), inv_col
// (every lane of column m) 1 / L[m][m].  Columns >= nvalid must hold identity padding (unit diagonal, zero couplings).
namespace acc16 {
// One-instruction asm statements in issue order, as in dpp16 above: hipcc keeps them where they are written, so the NEXT
// pivot's dependent chain (copy row C + 1 to all rows -> pivot -> reciprocal + two Newton steps -> w) is interleaved by hand
// with the current pivot's eight updates.  Hazards kept by construction: a VGPR written by a VALU instruction is read through
// DPP two or more instructions later; the result of v_rcp_f64 is read one or more instructions later
// (tools/check_dpp_hazards.py checks the first on the shipped ISA at every build).
struct Chain {            // the next pivot's chain state
  unsigned tl, th, ul, uh;  // halves of the two copies being swapped
  double v, piv, r, e, w;
};
__device__ __forceinline__ void mov2(unsigned& dl, unsigned& dh, const double& x) {
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(dl), "=&v"(dh) : "v"(__builtin_bit_cast(uint2, x).x), "v"(__builtin_bit_cast(uint2, x).y));
}
template <int C>
__device__ __forceinline__ void fmac_row_bcast(double& acc, const double& w) {   // acc += acc[lane C of the 16-lane row] * w
  if constexpr (C == 0)   // (hipcc may copy the freshly loaded block into the registers it picked for the loop right in front of
                          //  the first update: two wait states between such a copy and the DPP read)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "n"(C));
  else
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "n"(C));
}
// stage 1: both copies of the register that holds row CN; first swap (rows inside the pairs {0,1}, {2,3})
template <int CN>
__device__ __forceinline__ void chain_swap16(Chain& ch, const double& src) {
  mov2(ch.tl, ch.th, src);
  mov2(ch.ul, ch.uh, src);
  if constexpr ((CN & 1) == 0)   // t = [x0 x0 x2 x2]
    asm volatile("v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(ch.tl), "+v"(ch.th), "+v"(ch.ul), "+v"(ch.uh));
  else                           // t = [x1 x1 x3 x3]
    asm volatile("v_permlane16_swap_b32 %2, %0\n\tv_permlane16_swap_b32 %3, %1" : "+v"(ch.tl), "+v"(ch.th), "+v"(ch.ul), "+v"(ch.uh));
}
// stage 2: second swap (across the halves): v = row CN % 4 in every row
template <int CN>
__device__ __forceinline__ void chain_swap32(Chain& ch) {
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(ch.ul), "=&v"(ch.uh) : "v"(ch.tl), "v"(ch.th));
  if constexpr ((CN & 3) < 2)
    asm volatile("v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(ch.tl), "+v"(ch.th), "+v"(ch.ul), "+v"(ch.uh));
  else
    asm volatile("v_permlane32_swap_b32 %2, %0\n\tv_permlane32_swap_b32 %3, %1" : "+v"(ch.tl), "+v"(ch.th), "+v"(ch.ul), "+v"(ch.uh));
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(reinterpret_cast<uint2&>(ch.v).x), "=&v"(reinterpret_cast<uint2&>(ch.v).y) : "v"(ch.tl), "v"(ch.th));
}
template <int CN>
__device__ __forceinline__ void chain_pivot(Chain& ch) {   // (ch.v was written two or more instructions ago)
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(ch.piv) : "v"(ch.v), "n"(CN));
}
__device__ __forceinline__ void chain_rcp(Chain& ch) { asm volatile("v_rcp_f64 %0, %1" : "=v"(ch.r) : "v"(ch.piv)); }
__device__ __forceinline__ void chain_err(Chain& ch) { asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(ch.e) : "v"(ch.piv), "v"(ch.r)); }
__device__ __forceinline__ void chain_nr(Chain& ch) { asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(ch.r) : "v"(ch.e)); }
// w[m] = -A[CN][m] / piv for the columns m > CN still to be updated, 0 for the others (gt: this lane's m > CN)
__device__ __forceinline__ void chain_w(Chain& ch, bool gt) {
  asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(ch.w) : "v"(ch.r), "v"(ch.v));
  ch.w = gt ? ch.w : 0.0;
}

// pivot C: its w is in `cur`; the chain of pivot C + 1 is built in `nxt` under the eight updates
template <int C>
__device__ __forceinline__ void pivot(double (&a)[4], double (&b)[4], int m, const Chain& cur, Chain& nxt) {
  constexpr int CN = C + 1;
  constexpr int Q1 = (CN < 16) ? CN / 4 : 0;
  constexpr int o0 = (Q1 + 1) & 3, o1 = (Q1 + 2) & 3, o2 = (Q1 + 3) & 3;
  fmac_row_bcast<C>(a[Q1], cur.w);                       // the register that holds row C + 1 first: the next chain starts on it
  if constexpr (CN < 16) {
    chain_swap16<CN>(nxt, a[Q1]);
    fmac_row_bcast<C>(a[o0], cur.w);
    chain_swap32<CN>(nxt);
    fmac_row_bcast<C>(a[o1], cur.w);
    fmac_row_bcast<C>(a[o2], cur.w);
    chain_pivot<CN>(nxt);
    fmac_row_bcast<C>(b[0], cur.w);
    chain_rcp(nxt);
    fmac_row_bcast<C>(b[1], cur.w);
    chain_err(nxt);
    chain_nr(nxt);
    fmac_row_bcast<C>(b[2], cur.w);
    chain_err(nxt);
    chain_nr(nxt);
    fmac_row_bcast<C>(b[3], cur.w);
    chain_w(nxt, m > CN);
  } else {
    fmac_row_bcast<C>(a[o0], cur.w); fmac_row_bcast<C>(a[o1], cur.w); fmac_row_bcast<C>(a[o2], cur.w);
    fmac_row_bcast<C>(b[0], cur.w); fmac_row_bcast<C>(b[1], cur.w); fmac_row_bcast<C>(b[2], cur.w); fmac_row_bcast<C>(b[3], cur.w);
  }
}
template <int C>
__device__ __forceinline__ void pivots_from(double (&a)[4], double (&b)[4], int m, int nvalid, Chain& cur) {
  if constexpr (C < 16) {
    if (C < nvalid) {   // (uniform)
      Chain nxt;
      pivot<C>(a, b, m, cur, nxt);
      if constexpr (C + 1 < 16) pivots_from<C + 1>(a, b, m, nvalid, nxt);
    }
  }
}
}  // namespace acc16

// ---- round 5: the same factorisation, bit for bit, with the DEAD updates left out -----------------------------------------------
// tools/ubench/diag16.hip (one wave alone): acc16 takes 228 cycles per pivot for 15 f64 instructions (8.5 cycles of issue each)
// + 14 32-bit ones (4 each) = 184 cycles of issue — the block is ISSUE-bound on its one wave, not bound by its dependent chain.
// (Measured and rejected first: taking the reciprocal chain off the row replication — the next two rows kept replicated in
//  registers and updated like the block, the pivot from p' = x1 - x0 (u r): seven dependent steps instead of sixteen, the same
//  bits, and 269 cycles per pivot: three more f64 instructions.)  So: fewer instructions.  Of the eight rank-1 updates of a
// pivot, those of a register none of whose entries is read again are skipped:
//     a[q] once its rows 4 q .. 4 q + 3 are all <= C   (left of the diagonal they are final, right of it never read),
//     b[q] while its rows are all > C                  (the appended rows start as the identity: b[r][C] is still exactly 0),
// which leaves 4 or 5 updates per pivot (4.5 on average) and none at the last.  Same operands in the same operations for every entry that is read: bit-identical to acc16 (diag16: 0 words).
namespace acc16c {
using acc16::Chain;
using acc16::fmac_row_bcast;
// update number I of pivot C, in issue order: the live registers of the block (the one that holds row C + 1 first: the next
// pivot's chain starts on it), then the appended rows' registers that can be non-zero in column C
template <int C, int I>
__device__ __forceinline__ void upd(double (&a)[4], double (&b)[4], const double& w) {
  constexpr int qa0 = (C + 1) / 4;                    // first register of the block with a row > C (it holds row C + 1)
  constexpr int na = (C + 1 < 16) ? 4 - qa0 : 0;
  constexpr int nb = C / 4 + 1;                       // registers of the appended rows with a row <= C
  if constexpr (C + 1 < 16) {                         // (pivot 15: w is zero everywhere)
    if constexpr (I < na) fmac_row_bcast<C>(a[qa0 + I], w);
    else if constexpr (I < na + nb) fmac_row_bcast<C>(b[I - na], w);
  }
}
template <int C>
__device__ __forceinline__ void pivot(double (&a)[4], double (&b)[4], int m, const Chain& cur, Chain& nxt) {
  constexpr int CN = C + 1;
  if constexpr (CN < 16) {
    constexpr int n_upd = (4 - CN / 4) + (C / 4 + 1);   // 4 or 5
    upd<C, 0>(a, b, cur.w);                           // a[(C + 1) / 4]: row C + 1
    acc16::chain_swap16<CN>(nxt, a[CN / 4]);
    upd<C, 1>(a, b, cur.w);
    acc16::chain_swap32<CN>(nxt);
    upd<C, 2>(a, b, cur.w); upd<C, 3>(a, b, cur.w);   // (two instructions between the last write of nxt.v and its DPP read)
    acc16::chain_pivot<CN>(nxt);
    acc16::chain_rcp(nxt);
    if constexpr (n_upd > 4) upd<C, 4>(a, b, cur.w); else asm volatile("s_nop 0");   // (one between the reciprocal and its first use)
    acc16::chain_err(nxt);
    acc16::chain_nr(nxt);
    acc16::chain_err(nxt);
    acc16::chain_nr(nxt);
    acc16::chain_w(nxt, m > CN);
  }
}
template <int C>
__device__ __forceinline__ void pivots_from(double (&a)[4], double (&b)[4], int m, int nvalid, Chain& cur) {
  if constexpr (C < 16) {
    if (C < nvalid) {   // (uniform)
      Chain nxt;
      acc16c::pivot<C>(a, b, m, cur, nxt);   // (qualified: Chain lives in acc16, whose pivot would be found too)
      if constexpr (C + 1 < 16) acc16c::pivots_from<C + 1>(a, b, m, nvalid, nxt);
    }
  }
}
}  // namespace acc16c

// the common tail: the diagonal of the unscaled factor -> 1 / L[m][m] per column, scale, status
__device__ __forceinline__ bool diag_factor16_finish(double (&a)[4], double (&b)[4], int m, double& inv_col) {
  // column m's pivot sits in register m / 4, row m % 4, lane column m
  double d = (m < 4) ? a[0] : (m < 8 ? a[1] : (m < 12 ? a[2] : a[3]));
  {
    const int src = ((m & 3) << 4) | m;       // lane (m, m % 4)
    const long long bits = __double_as_longlong(d);
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(bits & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(bits >> 32));
    d = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
  const bool okp = __all((d > 0.0) && (d < 1e300));
  double inv = __builtin_amdgcn_rsq(d);
  inv = inv * (1.5 - 0.5 * d * inv * inv);
  inv = inv * (1.5 - 0.5 * d * inv * inv);
  inv_col = inv;
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] *= inv; b[q] *= inv; }
  return okp;
}

__device__ __forceinline__ bool diag_factor16_acc(double (&a)[4], double (&b)[4], int lane, double& inv_col, int nvalid = 16) {
  const int m = lane & 15;
  {
    acc16::Chain c0;                                   // pivot 0's chain, nothing to hide it under
    acc16::chain_swap16<0>(c0, a[0]);
    acc16::chain_swap32<0>(c0);
    asm volatile("s_nop 1");
    acc16::chain_pivot<0>(c0);
    acc16::chain_rcp(c0);
    asm volatile("s_nop 0");
    acc16::chain_err(c0); acc16::chain_nr(c0); acc16::chain_err(c0); acc16::chain_nr(c0);
    acc16::chain_w(c0, m > 0);
    if (nvalid > 0) acc16c::pivots_from<0>(a, b, m, nvalid, c0);
  }
  return diag_factor16_finish(a, b, m, inv_col);
}

// round 4's form (acc16: all eight updates at every pivot).  Not used by the product's kernels any more;
// kept as the bit-for-bit reference of the form above (tools/ubench/diag16.hip, tests/test_gpu_window_lm.py).
__device__ __forceinline__ bool diag_factor16_acc_r4(double (&a)[4], double (&b)[4], int lane, double& inv_col, int nvalid = 16) {
  const int m = lane & 15;
  {
    acc16::Chain c0;                                   // pivot 0's chain, nothing to hide it under
    acc16::chain_swap16<0>(c0, a[0]);
    acc16::chain_swap32<0>(c0);
    asm volatile("s_nop 1");
    acc16::chain_pivot<0>(c0);
    acc16::chain_rcp(c0);
    asm volatile("s_nop 0");
    acc16::chain_err(c0); acc16::chain_nr(c0); acc16::chain_err(c0); acc16::chain_nr(c0);
    acc16::chain_w(c0, m > 0);
    if (nvalid > 0) acc16::pivots_from<0>(a, b, m, nvalid, c0);
  }
  return diag_factor16_finish(a, b, m, inv_col);
}

}  // namespace
}  // namespace bodyfit
