// host_solver.cpp — the outer loop the reference leaves to ceres::Solve, restated for hosts without Ceres.
//
// The reference configures Ceres as: trust-region Levenberg-Marquardt, DENSE_QR, HuberLoss(3.0) on the
// reprojection blocks, bounds on the scale, max_num_iterations, everything else default
// (include/Sim3BA.h:406-407,449-451,472-479,641-647; include/MultiFrameBA.h:63-64,144-151).  Ceres is
// not available here, so this file follows Ceres 1.14's documented algorithm:
//   * Triggs corrector with rho'' <= 0 for Huber: residuals and Jacobian rows scaled by sqrt(rho')
//   * Jacobi column scaling 1 / (1 + ||J_col||) fixed at the first iterate
//   * LM step: (J^T J + diag(clamp(diag(J^T J), 1e-6, 1e32)) / radius) d = -J^T r
//   * step quality rho = cost change / model cost change; accept if rho > 1e-3;
//     radius /= max(1/3, 1 - (2 rho - 1)^3) on success, radius /= 2^k after k consecutive failures
//   * termination: |dcost|/cost < 1e-6, max |g| < 1e-10, |d| < 1e-8 (|x| + 1e-8), or max iterations
//   * box bounds by projecting the candidate point
// Every evaluation is one sweep of the HIP evaluator (bodyfit_evaluate_batch); the only CPU arithmetic
// here is the linear algebra of the normal equations.  DENSE_QR is replaced by a block-tridiagonal
// Cholesky (76x76 frame blocks coupled by the temporal term) with a Schur complement on the shared
// shape block: same minimiser, O(F 76^3) instead of O(m n^2).
#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bodyfit.h"
#include "solver_view.h"

namespace {

constexpr int NP = BODYFIT_FRAME_PARAMS;

// ---- small dense helpers (row-major) ------------------------------------------------------------
// contiguous dot product on AVX2 (every MI355X host has it): two 4-lane accumulators, fixed association
inline double dot4(const double* a, const double* b, int n) {
  __m256d s0 = _mm256_setzero_pd(), s1 = _mm256_setzero_pd();
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    s0 = _mm256_fmadd_pd(_mm256_loadu_pd(a + k), _mm256_loadu_pd(b + k), s0);
    s1 = _mm256_fmadd_pd(_mm256_loadu_pd(a + k + 4), _mm256_loadu_pd(b + k + 4), s1);
  }
  if (k + 4 <= n) { s0 = _mm256_fmadd_pd(_mm256_loadu_pd(a + k), _mm256_loadu_pd(b + k), s0); k += 4; }
  s0 = _mm256_add_pd(s0, s1);
  const __m128d lo = _mm256_castpd256_pd128(s0), hi = _mm256_extractf128_pd(s0, 1);
  const __m128d q = _mm_add_pd(lo, hi);
  double s = _mm_cvtsd_f64(q) + _mm_cvtsd_f64(_mm_unpackhi_pd(q, q));
  for (; k < n; ++k) s += a[k] * b[k];
  return s;
}
bool chol_inplace(double* A, int n) {  // lower Cholesky in the lower triangle
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j] - dot4(A + j * n, A + j * n, j);
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < n; ++i) A[i * n + j] = (A[i * n + j] - dot4(A + i * n, A + j * n, j)) * inv;
  }
  return true;
}
// X <- L^{-1} X, X is n x m; lower_rhs: X is lower triangular on entry (and stays so), columns > i are skipped
void fwd_solve(const double* L, int n, double* X, int m, bool lower_rhs = false) {
  for (int i = 0; i < n; ++i) {
    const int mc = lower_rhs ? std::min(m, i + 1) : m;
    double* Xi = X + (size_t)i * m;
    for (int k = 0; k < i; ++k) {
      const double l = L[i * n + k];
      if (l == 0.0) continue;
      const double* Xk = X + (size_t)k * m;
      const int kc = lower_rhs ? std::min(mc, k + 1) : mc;
      for (int c = 0; c < kc; ++c) Xi[c] -= l * Xk[c];
    }
    const double inv = 1.0 / L[i * n + i];
    for (int c = 0; c < mc; ++c) Xi[c] *= inv;
  }
}
// X <- L^{-T} X
void bwd_solve(const double* L, int n, double* X, int m) {
  for (int i = n - 1; i >= 0; --i) {
    const double inv = 1.0 / L[i * n + i];
    for (int c = 0; c < m; ++c) X[i * m + c] *= inv;
    for (int k = 0; k < i; ++k) {
      const double l = L[i * n + k];
      if (l == 0.0) continue;
      for (int c = 0; c < m; ++c) X[k * m + c] -= l * X[i * m + c];
    }
  }
}

// Independent problems (one per frame in the batched single-frame mode) are solved on host threads: the
// per-problem normal equations are 86 x 86, far too small to be worth a round trip each, and there are
// hundreds of them per LM iteration.
template <typename Fn>
void parallel_groups(size_t n, Fn&& fn) {
  const size_t hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min<size_t>(hw, 64), n);
  if (nt <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(nt);
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      for (size_t i = t; i < n; i += nt) fn(i);
    });
  for (auto& x : th) x.join();
}

struct Group {  // frames [f0, f1) sharing one LM state (and one beta block when present)
  int f0 = 0, f1 = 0;
  double radius = 1e4, decrease_factor = 2.0;
  double cost = 0.0;
  bool active = true;
  int termination = 1;  // 0 convergence, 1 no convergence (iterations), 2 failure
  int iterations = 0, n_ok = 0, n_bad = 0;
  double initial_cost = 0.0;
  std::vector<double> scale;  // Jacobi column scaling, [nf*NP + nb]
  std::string why;
};

struct Ctx {
  bodyfit_problem* p;
  bodyfit_layout lay;
  int F, nS, nb;           // nb = size of the beta block (0 if absent)
  bool beta_per_frame;
  double beta_pose, beta_shape, lambda_t, huber;
  int prior_rows;          // 0, 69 or 70
  const double* prec_cho;  // [K][69][69] when the GMM prior is on
  std::vector<int> kp_off; // per-frame keypoint offsets
};

double huber_rho(double delta, double s, double* rho1) {
  const double b = delta * delta;
  if (delta > 0.0 && s > b) {
    const double rt = std::sqrt(s);
    *rho1 = delta / rt;
    return 2.0 * delta * rt - b;
  }
  *rho1 = 1.0;
  return s;
}

// cost of a group from the residual vector (1/2 sum rho)
double group_cost(const Ctx& c, const Group& g, const double* r) {
  double cost = 0.0;
  for (int k = c.kp_off[g.f0]; k < c.kp_off[g.f1]; ++k) {
    double r1;
    cost += 0.5 * huber_rho(c.huber, r[2 * k] * r[2 * k] + r[2 * k + 1] * r[2 * k + 1], &r1);
  }
  const int D = NP - 7;
  if (c.prior_rows > 0) {
    const double* rp = r + c.lay.reproj_rows;
    for (int f = g.f0; f < g.f1; ++f)
      for (int i = 0; i < c.prior_rows; ++i) cost += 0.5 * rp[(size_t)f * c.prior_rows + i] * rp[(size_t)f * c.prior_rows + i];
  }
  if (c.lay.shape_rows > 0) {
    const double* rs = r + c.lay.reproj_rows + (size_t)c.F * c.prior_rows;
    if (c.beta_per_frame) {
      for (int f = g.f0; f < g.f1; ++f)
        for (int i = 0; i < c.nS; ++i) cost += 0.5 * rs[(size_t)f * c.nS + i] * rs[(size_t)f * c.nS + i];
    } else {
      for (int i = 0; i < c.nS; ++i) cost += 0.5 * rs[i] * rs[i];
    }
  }
  if (c.lay.temporal_rows > 0) {
    const double* rt = r + c.lay.reproj_rows + (size_t)c.F * c.prior_rows + c.lay.shape_rows;
    for (int f = g.f0; f + 1 < g.f1; ++f)
      for (int i = 0; i < 6 + D; ++i) cost += 0.5 * rt[(size_t)f * (6 + D) + i] * rt[(size_t)f * (6 + D) + i];
  }
  return cost;
}

// Normal equations of one group in block form.
struct Normal {
  int nf, nb;
  std::vector<double> A;   // [nf][NP*NP] diagonal blocks
  std::vector<double> E;   // [nf-1][NP]   diagonal of the (diagonal) coupling blocks  A_{f,f+1} = diag(E_f)
  std::vector<double> B;   // [nf][NP*nb]  frame x beta
  std::vector<double> C;   // [nb*nb]
  std::vector<double> g;   // [nf*NP + nb] gradient J^T r
};

// Hblk (optional): the reprojection part built on the device, [F][87][88] lower panels (k_frame_normal); J is then unused
void build_normal(const Ctx& c, const Group& g, const double* r, const double* J, const int* comp, Normal& N,
                  const double* Hblk = nullptr) {
  const int nf = g.f1 - g.f0, nb = c.nb, ncols = c.lay.n_cols, D = NP - 7;
  N.nf = nf; N.nb = nb;
  N.A.assign((size_t)nf * NP * NP, 0.0);
  N.E.assign((size_t)std::max(0, nf - 1) * NP, 0.0);
  N.B.assign((size_t)nf * NP * std::max(nb, 1), 0.0);
  N.C.assign((size_t)std::max(nb, 1) * std::max(nb, 1), 0.0);
  N.g.assign((size_t)nf * NP + nb, 0.0);
  for (int lf = 0; lf < nf; ++lf) {
    const int f = g.f0 + lf;
    double* A = &N.A[(size_t)lf * NP * NP];
    double* B = &N.B[(size_t)lf * NP * std::max(nb, 1)];
    double* gf = &N.g[(size_t)lf * NP];
    double* gb = &N.g[(size_t)nf * NP];
    if (Hblk) {
      const double* H = Hblk + (size_t)f * 87 * 88;
      for (int i = 0; i < NP; ++i) {
        for (int j = 0; j <= i; ++j) A[i * NP + j] += H[i * 88 + j];   // (+=: the previous frame's temporal pair is already in)
        gf[i] += H[ncols * 88 + i];
      }
      for (int ib = 0; ib < nb; ++ib) {
        const double* Hr = H + (size_t)(NP + ib) * 88;
        for (int j = 0; j < NP; ++j) B[j * nb + ib] += Hr[j];
        for (int jb = 0; jb <= ib; ++jb) N.C[ib * nb + jb] += Hr[NP + jb];
        gb[ib] += H[ncols * 88 + NP + ib];
      }
    }
    for (int k = c.kp_off[f]; !Hblk && k < c.kp_off[f + 1]; ++k) {
      const double r0 = r[2 * k], r1 = r[2 * k + 1];
      double rho1;
      huber_rho(c.huber, r0 * r0 + r1 * r1, &rho1);
      for (int row = 0; row < 2; ++row) {
        const double* Jr = J + (size_t)(2 * k + row) * ncols;
        const double rr = (row ? r1 : r0) * rho1;
        // sparsity: a keypoint row touches Sim3 + a few chain joints + beta
        int nz[NP + 16], nnz = 0;
        for (int i = 0; i < ncols; ++i)
          if (Jr[i] != 0.0) nz[nnz++] = i;
        for (int a = 0; a < nnz; ++a) {
          const int ia = nz[a];
          const double ja = Jr[ia] * rho1;
          if (ia < NP) gf[ia] += Jr[ia] * rr; else gb[ia - NP] += Jr[ia] * rr;
          for (int b2 = 0; b2 <= a; ++b2) {
            const int ib = nz[b2];
            const double v = ja * Jr[ib];
            if (ia < NP) A[ia * NP + ib] += v;                       // ib <= ia < NP (lower triangle)
            else if (ib < NP) B[ib * nb + (ia - NP)] += v;           // frame x beta
            else N.C[(ia - NP) * nb + (ib - NP)] += v;               // beta x beta (lower)
          }
        }
      }
    }
    // pose prior
    if (c.prior_rows > 0) {
      const double* rp = r + c.lay.reproj_rows + (size_t)f * c.prior_rows;
      const double bp = c.beta_pose;
      if (c.prec_cho) {
        // J = beta_p L_k^T on the top 69 rows (include/Sim3BA.h:298-299): J^T J = beta_p^2 L L^T, J^T r = beta_p L r
        const double* L = c.prec_cho + (size_t)comp[f] * D * D;
        for (int i = 0; i < D; ++i) {
          double gi = 0.0;
          for (int k = 0; k <= i; ++k) gi += L[i * D + k] * rp[k];
          gf[7 + i] += bp * gi;
          for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            const int kmax = std::min(i, j);
            for (int k = 0; k <= kmax; ++k) s += L[i * D + k] * L[j * D + k];
            A[(7 + i) * NP + 7 + j] += bp * bp * s;
          }
        }
      } else {
        for (int i = 0; i < D; ++i) { A[(7 + i) * NP + 7 + i] += bp * bp; gf[7 + i] += bp * rp[i]; }
      }
    }
    // per-frame shape prior
    if (c.lay.shape_rows > 0 && c.beta_per_frame) {
      const double* rs = r + c.lay.reproj_rows + (size_t)c.F * c.prior_rows + (size_t)f * c.nS;
      for (int i = 0; i < nb; ++i) { N.C[i * nb + i] += c.beta_shape * c.beta_shape; gb[i] += c.beta_shape * rs[i]; }
    }
    // temporal link f -> f+1 inside the group: rows lambda (a_f[src] - a_{f+1}[src])
    if (c.lay.temporal_rows > 0 && lf + 1 < nf) {
      const double* rt = r + c.lay.reproj_rows + (size_t)c.F * c.prior_rows + c.lay.shape_rows + (size_t)f * (6 + D);
      const double lam = c.lambda_t;
      double* A2 = &N.A[(size_t)(lf + 1) * NP * NP];
      double* g2 = &N.g[(size_t)(lf + 1) * NP];
      for (int i = 0; i < 6 + D; ++i) {
        const int src = (i < 3) ? (4 + i) : (i < 6 ? (1 + (i - 3)) : (7 + (i - 6)));
        A[src * NP + src] += lam * lam;
        A2[src * NP + src] += lam * lam;
        N.E[(size_t)lf * NP + src] = -lam * lam;
        gf[src] += lam * rt[i];
        g2[src] -= lam * rt[i];
      }
    }
  }
  if (c.lay.shape_rows > 0 && !c.beta_per_frame) {
    const double* rs = r + c.lay.reproj_rows + (size_t)c.F * c.prior_rows;
    double* gb = &N.g[(size_t)nf * NP];
    for (int i = 0; i < nb; ++i) { N.C[i * nb + i] += c.beta_shape * c.beta_shape; gb[i] += c.beta_shape * rs[i]; }
  }
  // mirror lower -> upper
  for (int lf = 0; lf < nf; ++lf) {
    double* A = &N.A[(size_t)lf * NP * NP];
    for (int i = 0; i < NP; ++i)
      for (int j = 0; j < i; ++j) A[j * NP + i] = A[i * NP + j];
  }
  for (int i = 0; i < nb; ++i)
    for (int j = 0; j < i; ++j) N.C[j * nb + i] = N.C[i * nb + j];
}

// c[0..len) -= a0 b0 + a1 b1 + a2 b2 + a3 b3 (rank-4 row update)
inline void axpy4(double* c, int len, const double* a, const double* b0, const double* b1, const double* b2,
                  const double* b3) {
  const __m256d a0 = _mm256_set1_pd(a[0]), a1 = _mm256_set1_pd(a[1]), a2 = _mm256_set1_pd(a[2]), a3 = _mm256_set1_pd(a[3]);
  int j = 0;
  for (; j + 4 <= len; j += 4) {
    __m256d v = _mm256_loadu_pd(c + j);
    v = _mm256_fnmadd_pd(a0, _mm256_loadu_pd(b0 + j), v);
    v = _mm256_fnmadd_pd(a1, _mm256_loadu_pd(b1 + j), v);
    v = _mm256_fnmadd_pd(a2, _mm256_loadu_pd(b2 + j), v);
    v = _mm256_fnmadd_pd(a3, _mm256_loadu_pd(b3 + j), v);
    _mm256_storeu_pd(c + j, v);
  }
  for (; j < len; ++j) c[j] -= a[0] * b0[j] + a[1] * b1[j] + a[2] * b2[j] + a[3] * b3[j];
}

// two rows at once against the same four b rows (the b loads are shared): c0 over len0, c1 over len1 >= len0
inline void axpy4x2(double* c0, int len0, const double* a, double* c1, int len1, const double* e, const double* b0,
                    const double* b1, const double* b2, const double* b3) {
  const __m256d a0 = _mm256_set1_pd(a[0]), a1 = _mm256_set1_pd(a[1]), a2 = _mm256_set1_pd(a[2]), a3 = _mm256_set1_pd(a[3]);
  const __m256d e0 = _mm256_set1_pd(e[0]), e1 = _mm256_set1_pd(e[1]), e2 = _mm256_set1_pd(e[2]), e3 = _mm256_set1_pd(e[3]);
  int j = 0;
  for (; j + 4 <= len0; j += 4) {
    const __m256d v0 = _mm256_loadu_pd(b0 + j), v1 = _mm256_loadu_pd(b1 + j), v2 = _mm256_loadu_pd(b2 + j),
                  v3 = _mm256_loadu_pd(b3 + j);
    __m256d x = _mm256_loadu_pd(c0 + j), y = _mm256_loadu_pd(c1 + j);
    x = _mm256_fnmadd_pd(a0, v0, x); y = _mm256_fnmadd_pd(e0, v0, y);
    x = _mm256_fnmadd_pd(a1, v1, x); y = _mm256_fnmadd_pd(e1, v1, y);
    x = _mm256_fnmadd_pd(a2, v2, x); y = _mm256_fnmadd_pd(e2, v2, y);
    x = _mm256_fnmadd_pd(a3, v3, x); y = _mm256_fnmadd_pd(e3, v3, y);
    _mm256_storeu_pd(c0 + j, x);
    _mm256_storeu_pd(c1 + j, y);
  }
  for (int q = j; q < len0; ++q) c0[q] -= a[0] * b0[q] + a[1] * b1[q] + a[2] * b2[q] + a[3] * b3[q];
  for (int q = j; q < len1; ++q) c1[q] -= e[0] * b0[q] + e[1] * b1[q] + e[2] * b2[q] + e[3] * b3[q];
}

// B = A^T for 76 x 76 row-major matrices, 4 x 4 blocks in registers
inline void transpose76(const double* A, double* B) {
  for (int i = 0; i < NP; i += 4)
    for (int j = 0; j < NP; j += 4) {
      const __m256d r0 = _mm256_loadu_pd(A + (size_t)i * NP + j), r1 = _mm256_loadu_pd(A + (size_t)(i + 1) * NP + j);
      const __m256d r2 = _mm256_loadu_pd(A + (size_t)(i + 2) * NP + j), r3 = _mm256_loadu_pd(A + (size_t)(i + 3) * NP + j);
      const __m256d t0 = _mm256_unpacklo_pd(r0, r1), t1 = _mm256_unpackhi_pd(r0, r1);
      const __m256d t2 = _mm256_unpacklo_pd(r2, r3), t3 = _mm256_unpackhi_pd(r2, r3);
      _mm256_storeu_pd(B + (size_t)j * NP + i, _mm256_permute2f128_pd(t0, t2, 0x20));
      _mm256_storeu_pd(B + (size_t)(j + 1) * NP + i, _mm256_permute2f128_pd(t1, t3, 0x20));
      _mm256_storeu_pd(B + (size_t)(j + 2) * NP + i, _mm256_permute2f128_pd(t0, t2, 0x31));
      _mm256_storeu_pd(B + (size_t)(j + 3) * NP + i, _mm256_permute2f128_pd(t1, t3, 0x31));
    }
}

// Right-looking Cholesky of the 76 x 76 matrix M (lower triangle, row-major, leading dimension 76), four columns at a
// time, with two borders that receive L^{-T} from the right as the factorisation proceeds: Eb (76 rows, row r starts
// as E_r e_r^T and ends as row r of diag(E) L^{-T}, which is zero left of column r; may be null) and Yb (m rows).
bool bordered_chol(double* M, double* Eb, double* Yb, int m) {
  static_assert(NP % 4 == 0, "column blocks of 4");
  double Lt[4][NP];   // the current panel transposed: Lt[k][j] = M[j][jb + k]
  for (int jb = 0; jb < NP; jb += 4) {
    // diagonal 4 x 4 block
    for (int j = jb; j < jb + 4; ++j) {
      double d = M[(size_t)j * NP + j];
      for (int k = jb; k < j; ++k) d -= M[(size_t)j * NP + k] * M[(size_t)j * NP + k];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      M[(size_t)j * NP + j] = d;
      const double inv = 1.0 / d;
      for (int i = j + 1; i < jb + 4; ++i) {
        double v = M[(size_t)i * NP + j];
        for (int k = jb; k < j; ++k) v -= M[(size_t)i * NP + k] * M[(size_t)j * NP + k];
        M[(size_t)i * NP + j] = v * inv;
      }
    }
    const double* D = M + (size_t)jb * NP + jb;   // the factored diagonal block (lower, leading dimension 76)
    // panel: rows below solve x L11^T = a
    const double i0 = 1.0 / D[0], i1 = 1.0 / D[NP + 1], i2 = 1.0 / D[2 * NP + 2], i3 = 1.0 / D[3 * NP + 3];
    const double d10 = D[NP], d20 = D[2 * NP], d21 = D[2 * NP + 1], d30 = D[3 * NP], d31 = D[3 * NP + 1], d32 = D[3 * NP + 2];
    auto panel_row = [&](double* row) {
      const double x0 = row[jb] * i0;
      const double x1 = (row[jb + 1] - x0 * d10) * i1;
      const double x2 = (row[jb + 2] - x0 * d20 - x1 * d21) * i2;
      const double x3 = (row[jb + 3] - x0 * d30 - x1 * d31 - x2 * d32) * i3;
      row[jb] = x0; row[jb + 1] = x1; row[jb + 2] = x2; row[jb + 3] = x3;
    };
    const int j1 = jb + 4, rest = NP - j1;
    for (int i = j1; i < NP; ++i) {
      panel_row(M + (size_t)i * NP);
      for (int k = 0; k < 4; ++k) Lt[k][i] = M[(size_t)i * NP + jb + k];
    }
    const int er = Eb ? std::min(NP, jb + 4) : 0;   // border rows r < jb + 4 have entered their non-zero part
    for (int r = 0; r < er; ++r) panel_row(Eb + (size_t)r * NP);
    for (int c = 0; c < m; ++c) panel_row(Yb + (size_t)c * NP);
    if (rest == 0) break;
    // trailing update with the transposed panel
    const double *t0 = Lt[0] + j1, *t1 = Lt[1] + j1, *t2 = Lt[2] + j1, *t3 = Lt[3] + j1;
    int i = j1;
    for (; i + 1 < NP; i += 2)
      axpy4x2(M + (size_t)i * NP + j1, i - j1 + 1, M + (size_t)i * NP + jb, M + (size_t)(i + 1) * NP + j1, i - j1 + 2,
              M + (size_t)(i + 1) * NP + jb, t0, t1, t2, t3);
    if (i < NP) axpy4(M + (size_t)i * NP + j1, i - j1 + 1, M + (size_t)i * NP + jb, t0, t1, t2, t3);
    int r = 0;
    for (; r + 1 < er; r += 2)
      axpy4x2(Eb + (size_t)r * NP + j1, rest, Eb + (size_t)r * NP + jb, Eb + (size_t)(r + 1) * NP + j1, rest,
              Eb + (size_t)(r + 1) * NP + jb, t0, t1, t2, t3);
    if (r < er) axpy4(Eb + (size_t)r * NP + j1, rest, Eb + (size_t)r * NP + jb, t0, t1, t2, t3);
    int c = 0;
    for (; c + 1 < m; c += 2)
      axpy4x2(Yb + (size_t)c * NP + j1, rest, Yb + (size_t)c * NP + jb, Yb + (size_t)(c + 1) * NP + j1, rest,
              Yb + (size_t)(c + 1) * NP + jb, t0, t1, t2, t3);
    if (c < m) axpy4(Yb + (size_t)c * NP + j1, rest, Yb + (size_t)c * NP + jb, t0, t1, t2, t3);
  }
  return true;
}

// Solve (S H S + diag(clamp(diag(S H S)))/radius) ds = -S g for the scaled step, return d = S ds and the
// model cost change  -d^T (g + 1/2 H d).  `free_mask` zeroes constant parameters.
bool solve_step(const Normal& N, const std::vector<double>& scale, const unsigned char* constant, double radius,
                std::vector<double>& d, double* model_change) {
  const int nf = N.nf, nb = N.nb, n = nf * NP + nb;
  std::vector<double> A(N.A), B(N.B), C(N.C), E(N.E), rhs(n);
  auto is_const = [&](int idx) { return constant && idx < nf * NP && constant[idx % NP]; };
  // scale, damp, pin constants
  for (int lf = 0; lf < nf; ++lf) {
    double* Af = &A[(size_t)lf * NP * NP];
    const double* s = &scale[(size_t)lf * NP];
    for (int i = 0; i < NP; ++i)
      for (int j = 0; j < NP; ++j) Af[i * NP + j] *= s[i] * s[j];
    if (nb)
      for (int i = 0; i < NP; ++i)
        for (int j = 0; j < nb; ++j) B[((size_t)lf * NP + i) * nb + j] *= s[i] * scale[(size_t)nf * NP + j];
    if (lf + 1 < nf)
      for (int i = 0; i < NP; ++i) E[(size_t)lf * NP + i] *= s[i] * scale[(size_t)(lf + 1) * NP + i];
    for (int i = 0; i < NP; ++i) rhs[lf * NP + i] = -N.g[lf * NP + i] * s[i];
    for (int i = 0; i < NP; ++i) {
      if (constant && constant[i]) {
        for (int j = 0; j < NP; ++j) { Af[i * NP + j] = 0.0; Af[j * NP + i] = 0.0; }
        Af[i * NP + i] = 1.0;
        for (int j = 0; j < nb; ++j) B[((size_t)lf * NP + i) * nb + j] = 0.0;
        if (lf + 1 < nf) E[(size_t)lf * NP + i] = 0.0;
        if (lf > 0) E[(size_t)(lf - 1) * NP + i] = 0.0;
        rhs[lf * NP + i] = 0.0;
      } else {
        const double dg = std::min(std::max(Af[i * NP + i], 1e-6), 1e32);
        Af[i * NP + i] += dg / radius;
      }
    }
  }
  for (int i = 0; i < nb; ++i) {
    for (int j = 0; j < nb; ++j) C[i * nb + j] *= scale[(size_t)nf * NP + i] * scale[(size_t)nf * NP + j];
    rhs[nf * NP + i] = -N.g[nf * NP + i] * scale[(size_t)nf * NP + i];
  }
  for (int i = 0; i < nb; ++i) C[i * nb + i] += std::min(std::max(C[i * nb + i], 1e-6), 1e32) / radius;
  (void)is_const;

  // block-tridiagonal Cholesky T = L L^T:  Ld_f = chol(A_f - Ls_{f-1} Ls_{f-1}^T),  Ls_f = diag(E_f) Ld_f^{-T}, and the
  // forward substitution of the nb + 1 right-hand sides [B | rhs], all three as ONE bordered factorisation per frame:
  // the rows of diag(E_f) and of [B | rhs]^T are appended below A_f and receive L^{-T} from the right while the top
  // 76 x 76 block is factorised (right-looking, 4 columns at a time, rank-4 row updates on AVX2).
  const int m = nb + 1;
  static thread_local std::vector<double> wsLs, wsYt, wsLsT;
  wsLs.resize((size_t)std::max(1, nf - 1) * NP * NP);   // Ls_f: block (f+1, f), row i is zero left of column i
  wsYt.resize((size_t)nf * m * NP);                     // (L^{-1} [B | rhs])^T: [frame][column][76]
  wsLsT.resize((size_t)NP * NP);
  double* const Ls = wsLs.data();
  double* const Yt = wsYt.data();
  double* const LsT = wsLsT.data();                     // transpose of the latest Ls
  for (int lf = 0; lf < nf; ++lf) {
    double* Af = &A[(size_t)lf * NP * NP];              // factorised in place (lower triangle)
    const bool has_next = lf + 1 < nf;
    double* Eb = has_next ? Ls + (size_t)lf * NP * NP : nullptr;   // rows of diag(E) -> Ls
    double* Yb = Yt + (size_t)lf * m * NP;                         // rows of [B | rhs]^T -> Yt
    for (int c = 0; c < nb; ++c)
      for (int i = 0; i < NP; ++i) Yb[(size_t)c * NP + i] = B[((size_t)lf * NP + i) * nb + c];
    for (int i = 0; i < NP; ++i) Yb[(size_t)nb * NP + i] = rhs[lf * NP + i];
    if (lf > 0) {
      // A_f -= Ls Ls^T and [B | rhs]^T -= (Ls Yp)^T as rank-4 row updates against Ls^T (LsT[k][j] = Ls[j][k], zero for
      // j > k; Ls[i][k] is zero for k < i, so the k groups may start at a multiple of 4 below i)
      const double* Lp = Ls + (size_t)(lf - 1) * NP * NP;
      const double* Yp = Yt + (size_t)(lf - 1) * m * NP;
      for (int i = 0; i < NP; ++i) {
        double* Mi = Af + (size_t)i * NP;
        for (int k = i & ~3; k < NP; k += 4)
          axpy4(Mi, i + 1, Lp + (size_t)i * NP + k, LsT + (size_t)k * NP, LsT + (size_t)(k + 1) * NP,
                LsT + (size_t)(k + 2) * NP, LsT + (size_t)(k + 3) * NP);
      }
      for (int c = 0; c < m; ++c)
        for (int k = 0; k < NP; k += 4)
          axpy4(Yb + (size_t)c * NP, k + 4, Yp + (size_t)c * NP + k, LsT + (size_t)k * NP, LsT + (size_t)(k + 1) * NP,
                LsT + (size_t)(k + 2) * NP, LsT + (size_t)(k + 3) * NP);
    }
    if (has_next) {
      std::memset(Eb, 0, sizeof(double) * NP * NP);
      for (int i = 0; i < NP; ++i) Eb[(size_t)i * NP + i] = E[(size_t)lf * NP + i];
    }
    if (!bordered_chol(Af, Eb, Yb, m)) return false;
    if (has_next) transpose76(Eb, LsT);
  }
  // Schur complement on beta: S = C - Yb^T Yb, rb = rhs_b - Yb^T y
  std::vector<double> db(nb, 0.0);
  if (nb) {
    std::vector<double> S(C), rb(nb);
    for (int i = 0; i < nb; ++i) rb[i] = rhs[nf * NP + i];
    for (int lf = 0; lf < nf; ++lf) {
      const double* Yf = Yt + (size_t)lf * m * NP;
      for (int i = 0; i < nb; ++i) {
        rb[i] -= dot4(Yf + (size_t)i * NP, Yf + (size_t)nb * NP, NP);
        for (int j = 0; j <= i; ++j) S[i * nb + j] -= dot4(Yf + (size_t)i * NP, Yf + (size_t)j * NP, NP);
      }
    }
    if (!chol_inplace(S.data(), nb)) return false;
    fwd_solve(S.data(), nb, rb.data(), 1);
    bwd_solve(S.data(), nb, rb.data(), 1);
    db = rb;
  }
  // back substitution:  z = y - Yb db ;  x = L^{-T} z  (block bidiagonal)
  std::vector<double> ds(n, 0.0);
  std::vector<double> z((size_t)nf * NP);
  for (int lf = 0; lf < nf; ++lf) {
    const double* Yf = Yt + (size_t)lf * m * NP;
    for (int i = 0; i < NP; ++i) {
      double v = Yf[(size_t)nb * NP + i];
      for (int j = 0; j < nb; ++j) v -= Yf[(size_t)j * NP + i] * db[j];
      z[lf * NP + i] = v;
    }
  }
  for (int lf = nf - 1; lf >= 0; --lf) {
    double* zf = &z[(size_t)lf * NP];
    if (lf + 1 < nf) {
      const double* Lsf = Ls + (size_t)lf * NP * NP;   // block (lf+1, lf): contributes Ls^T x_{lf+1}
      const double* xn = &ds[(size_t)(lf + 1) * NP];
      for (int i = 0; i < NP; ++i) {
        const double xi = xn[i];
        for (int k = i; k < NP; ++k) zf[k] -= Lsf[i * NP + k] * xi;
      }
    }
    bwd_solve(&A[(size_t)lf * NP * NP], NP, zf, 1);
    for (int i = 0; i < NP; ++i) ds[lf * NP + i] = zf[i];
  }
  for (int i = 0; i < nb; ++i) ds[nf * NP + i] = db[i];
  // unscale and model change with the UNDAMPED, unscaled system:  -d^T g - 1/2 d^T H d
  d.assign(n, 0.0);
  for (int i = 0; i < n; ++i) d[i] = ds[i] * scale[i];
  double dg = 0.0, dHd = 0.0;
  for (int i = 0; i < n; ++i) dg += d[i] * N.g[i];
  for (int lf = 0; lf < nf; ++lf) {
    const double* Af = &N.A[(size_t)lf * NP * NP];
    const double* x = &d[(size_t)lf * NP];
    for (int i = 0; i < NP; ++i) {
      double s = dot4(Af + (size_t)i * NP, x, NP);
      for (int j = 0; j < nb; ++j) s += 2.0 * N.B[((size_t)lf * NP + i) * nb + j] * d[nf * NP + j];
      if (lf + 1 < nf) s += 2.0 * N.E[(size_t)lf * NP + i] * d[(size_t)(lf + 1) * NP + i];
      dHd += x[i] * s;
    }
  }
  for (int i = 0; i < nb; ++i)
    for (int j = 0; j < nb; ++j) dHd += d[nf * NP + i] * N.C[i * nb + j] * d[nf * NP + j];
  *model_change = -dg - 0.5 * dHd;
  return true;
}

}  // namespace

extern "C" int bodyfit_solve(bodyfit_problem* p, double* frame_params, double* beta,
                             const unsigned char* param_constant, int independent_frames,
                             const bodyfit_fit_options* opt_in, bodyfit_fit_summary* summaries, int n_summaries) {
  if (!p || !frame_params) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: null argument");
  bodyfit_solver_view view;
  if (bodyfit_internal_solver_view(p, &view) != BODYFIT_OK) return BODYFIT_ERR_INVALID;
  bodyfit_fit_options opt;
  opt.max_iters = 100; opt.scale_lo = 0.3; opt.scale_hi = 3.0; opt.verbose = 0; opt.solver = 0;
  if (opt_in) opt = *opt_in;
  Ctx c;
  c.p = p;
  if (bodyfit_problem_layout(p, &c.lay) != BODYFIT_OK) return BODYFIT_ERR_INVALID;
  c.F = view.n_frames; c.nS = view.n_shape;
  c.nb = c.lay.n_cols > NP ? view.n_shape : 0;
  c.beta_per_frame = view.beta_per_frame != 0;
  c.beta_pose = view.beta_pose; c.beta_shape = view.beta_shape; c.lambda_t = view.lambda_temporal;
  c.huber = view.huber_delta;
  c.prior_rows = c.lay.prior_rows_per_frame;
  c.prec_cho = view.has_gmm ? view.prec_cho : nullptr;
  c.kp_off.assign(view.kp_offset, view.kp_offset + c.F + 1);
  if (view.n_joints != 24 || view.temporal_halo)   // (a shard with a halo row goes through bodyfit_solve_sharded)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: needs a 24-joint model and a whole window (no halo)");
  if (c.nb && !beta) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: beta required (shape block present)");
  if (independent_frames && c.F > 1 && (c.lambda_t > 0.0 || (c.nb && !c.beta_per_frame)))
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID,
                                 "bodyfit_solve: independent frames cannot share beta or temporal links");
  if (!independent_frames && c.F > 1 && c.nb && c.beta_per_frame)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: one problem over all frames needs a shared beta");

  // independent frames: the whole LM loop runs on the device (no per-iteration host round trip)
  const bool device_ok = (independent_frames || c.F == 1) && c.lambda_t == 0.0 && (!c.nb || c.beta_per_frame || c.F == 1) &&
                         view.max_kp_per_frame <= 32;
  if (opt.solver == 2 && !device_ok)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: the device loop handles independent frames with <= 32 keypoints");
  if (device_ok && opt.solver != 1)
    return bodyfit_internal_solve_batched_device(p, frame_params, beta, param_constant, &opt, summaries, n_summaries);

  // one problem over all frames with a shared beta: the device-resident window LM (block cyclic reduction over the
  // frames, k_window_lm.hip) when the problem has the shape it is built for; solver 1 keeps the host loop
  const bool window_ok = !independent_frames && c.nb == view.n_shape && view.n_shape == 10 && !c.beta_per_frame &&
                         !view.has_gmm && view.max_kp_per_frame <= 32;
  if (opt.solver == 3 && !window_ok)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_solve: the device window loop needs a shared 10-coefficient beta, the L2 pose prior and <= 32 keypoints per frame");
  // (below a dozen frames the chain is short enough for the host's sequential factorisation to win: measured per LM iteration
  //  on MI355X, host / device: 20 frames 0.60 / 0.49 ms, 103 frames 2.7 / 0.69 ms, 1024 frames 54 / 1.8 ms)
  const int window_min = std::getenv("BODYFIT_WINDOW_MIN") ? std::atoi(std::getenv("BODYFIT_WINDOW_MIN")) : 12;
  if (window_ok && (opt.solver == 3 || (opt.solver == 0 && c.F >= window_min)))
    return bodyfit_internal_solve_window_device(p, frame_params, beta, param_constant, &opt, summaries, nullptr);

  const int F = c.F, nb = c.nb;
  std::vector<Group> groups;
  if (independent_frames) {
    groups.resize(F);
    for (int f = 0; f < F; ++f) { groups[f].f0 = f; groups[f].f1 = f + 1; }
  } else {
    groups.resize(1);
    groups[0].f0 = 0; groups[0].f1 = F;
  }
  const size_t nbeta = nb ? (size_t)(c.beta_per_frame ? F * nb : nb) : 0;
  std::vector<double> x(frame_params, frame_params + (size_t)F * NP), xb(nbeta);
  if (nbeta) std::memcpy(xb.data(), beta, nbeta * sizeof(double));
  std::vector<double> r((size_t)c.lay.total_rows), rn((size_t)c.lay.total_rows);
  // normal equations: the reprojection part comes from the device as per-frame panels (k_frame_normal) whenever the
  // frames fit its 64-row staging; otherwise J is copied back and the Gram products are formed here
  const bool device_normals = view.max_kp_per_frame <= 32 && std::getenv("BODYFIT_HOST_NORMALS") == nullptr;
  std::vector<double> J(device_normals ? 0 : (size_t)c.lay.reproj_rows * c.lay.n_cols);
  std::vector<double> Hblk(device_normals ? (size_t)F * 87 * 88 : 0);
  std::vector<int> comp(F, 0), compn(F, 0);
  std::vector<double> xn(x), xbn(xb);
  auto eval_full = [&]() -> int {
    return device_normals
               ? bodyfit_internal_frame_normals(p, x.data(), nbeta ? xb.data() : nullptr, r.data(), comp.data(), Hblk.data())
               : bodyfit_evaluate_batch(p, x.data(), nbeta ? xb.data() : nullptr, r.data(), J.data(), comp.data(), 1);
  };
  int rc = eval_full();
  if (rc) return rc;
  int n_sweeps = 1;
  std::vector<Normal> normals(groups.size());
  std::vector<char> normal_valid(groups.size(), 0), has_cand(groups.size(), 0), was_active(groups.size(), 0);
  std::vector<std::vector<double>> steps(groups.size());
  std::vector<double> model_change(groups.size(), 0.0);
  auto beta_of = [&](const Group& g, std::vector<double>& vb) -> double* {
    return nb ? vb.data() + (c.beta_per_frame ? (size_t)g.f0 * nb : 0) : nullptr;
  };
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    Group& g = groups[gi];
    g.cost = g.initial_cost = group_cost(c, g, r.data());
    if (!std::isfinite(g.cost)) { g.active = false; g.termination = 2; g.why = "initial cost is not finite"; }
  }
  // BODYFIT_TIMING=1: where the host loop's wall time goes (diagnostic print at the end of the solve)
  const bool timing = std::getenv("BODYFIT_TIMING") != nullptr;
  double t_build = 0, t_solve = 0, t_eval_r = 0, t_eval_j = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int it = 0; it < opt.max_iters; ++it) {
    bool any_active = false, any_cand = false;
    xn = x; xbn = xb;
    auto process_group = [&](size_t gi) {
      Group& g = groups[gi];
      has_cand[gi] = 0;
      was_active[gi] = 0;
      if (!g.active) return;
      was_active[gi] = 1;
      const int nf = g.f1 - g.f0, n = nf * NP + nb;
      if (!normal_valid[gi]) {
        const double t0 = timing ? now() : 0.0;
        build_normal(c, g, r.data(), J.data(), comp.data(), normals[gi], device_normals ? Hblk.data() : nullptr);
        normal_valid[gi] = 1;
        if (timing && groups.size() == 1) t_build += now() - t0;
      }
      const Normal& N = normals[gi];
      if (g.scale.empty()) {   // Jacobi scaling from the first iterate
        g.scale.assign(n, 1.0);
        for (int lf = 0; lf < nf; ++lf)
          for (int i = 0; i < NP; ++i) g.scale[lf * NP + i] = 1.0 / (1.0 + std::sqrt(N.A[(size_t)lf * NP * NP + i * NP + i]));
        for (int i = 0; i < nb; ++i) g.scale[nf * NP + i] = 1.0 / (1.0 + std::sqrt(N.C[i * nb + i]));
      }
      // gradient tolerance (projected for the bounded scale)
      double gmax = 0.0;
      for (int lf = 0; lf < nf; ++lf)
        for (int i = 0; i < NP; ++i) {
          if (param_constant && param_constant[i]) continue;
          double gi2 = N.g[lf * NP + i];
          if (i == 0) {
            const double s0 = x[(size_t)(g.f0 + lf) * NP];
            const double proj = std::min(std::max(s0 - gi2, opt.scale_lo), opt.scale_hi);
            gi2 = s0 - proj;
          }
          gmax = std::max(gmax, std::fabs(gi2));
        }
      for (int i = 0; i < nb; ++i) gmax = std::max(gmax, std::fabs(N.g[nf * NP + i]));
      if (gmax <= 1e-10) { g.active = false; g.termination = 0; g.why = "gradient tolerance"; return; }
      const double ts0 = timing ? now() : 0.0;
      const bool step_ok = solve_step(N, g.scale, param_constant, g.radius, steps[gi], &model_change[gi]);
      if (timing && groups.size() == 1) t_solve += now() - ts0;
      if (!step_ok) {
        g.radius /= g.decrease_factor; g.decrease_factor *= 2.0; ++g.n_bad; ++g.iterations;
        if (g.radius < 1e-32) { g.active = false; g.termination = 2; g.why = "trust region collapsed"; }
        return;
      }
      std::vector<double>& d = steps[gi];
      // candidate, projected onto the scale bounds; the model change is re-evaluated for the projected step
      bool projected = false;
      for (int lf = 0; lf < nf; ++lf) {
        const size_t o = (size_t)(g.f0 + lf) * NP;
        const double s_new = std::min(std::max(x[o] + d[lf * NP], opt.scale_lo), opt.scale_hi);
        if (s_new != x[o] + d[lf * NP]) { d[lf * NP] = s_new - x[o]; projected = true; }
        for (int i = 0; i < NP; ++i) xn[o + i] = x[o + i] + d[lf * NP + i];
      }
      if (nb) {
        double* b0 = beta_of(g, xb);
        double* b1 = beta_of(g, xbn);
        for (int i = 0; i < nb; ++i) b1[i] = b0[i] + d[nf * NP + i];
      }
      if (projected) {
        double dg = 0.0, dHd = 0.0;
        for (int i = 0; i < n; ++i) dg += d[i] * N.g[i];
        for (int lf = 0; lf < nf; ++lf) {
          const double* Af = &N.A[(size_t)lf * NP * NP];
          for (int i = 0; i < NP; ++i) {
            double s = 0.0;
            for (int j = 0; j < NP; ++j) s += Af[i * NP + j] * d[lf * NP + j];
            for (int j = 0; j < nb; ++j) s += 2.0 * N.B[((size_t)lf * NP + i) * nb + j] * d[nf * NP + j];
            if (lf + 1 < nf) s += 2.0 * N.E[(size_t)lf * NP + i] * d[(lf + 1) * NP + i];
            dHd += d[lf * NP + i] * s;
          }
        }
        for (int i = 0; i < nb; ++i)
          for (int j = 0; j < nb; ++j) dHd += d[nf * NP + i] * N.C[i * nb + j] * d[nf * NP + j];
        model_change[gi] = -dg - 0.5 * dHd;
      }
      // parameter tolerance
      double dn = 0.0, xnorm = 0.0;
      for (int i = 0; i < n; ++i) dn += d[i] * d[i];
      for (int lf = 0; lf < nf; ++lf)
        for (int i = 0; i < NP; ++i) xnorm += x[(size_t)(g.f0 + lf) * NP + i] * x[(size_t)(g.f0 + lf) * NP + i];
      if (nb) { const double* b0 = beta_of(g, xb); for (int i = 0; i < nb; ++i) xnorm += b0[i] * b0[i]; }
      if (std::sqrt(dn) <= 1e-8 * (std::sqrt(xnorm) + 1e-8)) {
        g.active = false; g.termination = 0; g.why = "parameter tolerance";
        for (int lf = 0; lf < nf; ++lf)
          for (int i = 0; i < NP; ++i) xn[(size_t)(g.f0 + lf) * NP + i] = x[(size_t)(g.f0 + lf) * NP + i];
        if (nb) std::memcpy(beta_of(g, xbn), beta_of(g, xb), nb * sizeof(double));
        return;
      }
      has_cand[gi] = 1;
    };
    parallel_groups(groups.size(), process_group);
    for (size_t gi = 0; gi < groups.size(); ++gi) { any_active |= was_active[gi] != 0; any_cand |= has_cand[gi] != 0; }
    if (!any_active) break;
    if (!any_cand) continue;
    const double te0 = timing ? now() : 0.0;
    rc = bodyfit_evaluate_batch(p, xn.data(), nbeta ? xbn.data() : nullptr, rn.data(), nullptr, compn.data(), 0);
    if (rc) return rc;
    if (timing) t_eval_r += now() - te0;
    ++n_sweeps;
    bool any_accept = false;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      if (!has_cand[gi]) continue;
      Group& g = groups[gi];
      ++g.iterations;
      const double new_cost = group_cost(c, g, rn.data());
      const double change = g.cost - new_cost;
      const double rho = change / model_change[gi];
      const int nf = g.f1 - g.f0;
      if (std::isfinite(new_cost) && model_change[gi] > 0.0 && rho > 1e-3) {
        for (int lf = 0; lf < nf; ++lf)
          std::memcpy(&x[(size_t)(g.f0 + lf) * NP], &xn[(size_t)(g.f0 + lf) * NP], NP * sizeof(double));
        if (nb) std::memcpy(beta_of(g, xb), beta_of(g, xbn), nb * sizeof(double));
        const double old_cost = g.cost;
        g.cost = new_cost;
        const double t = 2.0 * rho - 1.0;
        g.radius = std::min(1e16, g.radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
        g.decrease_factor = 2.0;
        ++g.n_ok;
        normal_valid[gi] = 0;
        any_accept = true;
        if (std::fabs(change) < 1e-6 * old_cost) { g.active = false; g.termination = 0; g.why = "function tolerance"; }
      } else {
        g.radius /= g.decrease_factor; g.decrease_factor *= 2.0; ++g.n_bad;
        if (g.radius < 1e-32) { g.active = false; g.termination = 2; g.why = "trust region collapsed"; }
      }
      if (opt.verbose && groups.size() == 1)
        std::printf("[bodyfit] it %3d cost %.6e change %.3e rho %.3f radius %.3e\n", g.iterations, g.cost, change, rho, g.radius);
    }
    if (any_accept) {
      const double tj0 = timing ? now() : 0.0;
      rc = eval_full();
      if (rc) return rc;
      if (timing) t_eval_j += now() - tj0;
      ++n_sweeps;
    }
  }
  if (timing)
    std::fprintf(stderr, "[bodyfit timing] F=%d sweeps=%d  residual sweeps %.2f ms  jacobian sweeps %.2f ms  build_normal %.2f ms  solve_step %.2f ms\n",
                 F, n_sweeps, t_eval_r * 1e3, t_eval_j * 1e3, t_build * 1e3, t_solve * 1e3);
  std::memcpy(frame_params, x.data(), x.size() * sizeof(double));
  if (nbeta) std::memcpy(beta, xb.data(), nbeta * sizeof(double));
  if (summaries) {
    for (int i = 0; i < n_summaries && i < (int)groups.size(); ++i) {
      const Group& g = groups[i];
      bodyfit_fit_summary& s = summaries[i];
      s.iterations = g.iterations; s.termination = g.termination;
      s.usable = (g.termination != 2) ? 1 : 0;   // ceres::Solver::Summary::IsSolutionUsable
      s.n_successful = g.n_ok; s.n_unsuccessful = g.n_bad;
      s.initial_cost = g.initial_cost; s.final_cost = g.cost;
      s.n_sweeps = n_sweeps; s.n_sweeps_issued = n_sweeps;
    }
  }
  return BODYFIT_OK;
}
