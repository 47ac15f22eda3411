// bodyfit_oracle.cpp — CPU restatement (f64, C++17, OpenMP) of the reference hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
// of bench.py may load this library.  The product (3dbodyanimation_amd/) never links, imports or
// calls anything in oracle/.
//
// *** PARITY UNPINNED ***  The reference (jonH34400/3DBodyAnimation @ 2025-08-08) ships no tests,
// golden vectors or known-answer fixtures for this path, and cannot be built here (its SMPL library
// `external/avatar` is an empty submodule; Ceres/Eigen/OpenCV are absent).  This file follows the
// cited reference lines and the *published* algorithms of the absent third-party pieces
// (Ceres 1.14 rotation.h AngleAxisRotatePoint, HuberLoss, Triggs corrector; SMPL forward; the
// SMPLify max-mixture pose prior that sxyu/avatar's GaussianMixture implements).  It is pinned only
// by self-consistency: analytic Jacobian == dual-number autodiff == central finite differences.
//
// What follows which reference lines (all paths relative to /root/reference):
//   AngleAxisRotatePoint            Ceres 1.14 rotation.h (called at include/Sim3BA.h:61,77,177,216)
//   Jet<4>                          ceres::Jet under DynamicAutoDiffCostFunction (stride 4)
//                                   include/Sim3BA.h:420,581; include/MultiFrameBA.h:90
//   ReprojFunctor::operator()       include/Sim3BA.h:34-88 (ReprojCost), :126-227 (ReprojCostShape)
//   pose_prior_eval                 include/Sim3BA.h:263-315 (PosePriorAAAnalytic::Evaluate)
//   shape prior / temporal          include/Sim3BA.h:331-343, include/MultiFrameBA.h:20-28
//   rest offsets                    include/Sim3BA.h:367-392,533-555; include/MultiFrameBA.h:53-60
//   smpl_forward                    ark::Avatar::update() call sites include/Sim3BA.h:371,538;
//                                   include/MultiFrameBA.h:53,173 (SMPL public definition)
//   gmm_*                           ark::GaussianMixture uses at include/Sim3BA.h:257,266,280,288;
//                                   file format scripts/convert_gmm_to_avatar.py:14-29
//   huber                           ceres::HuberLoss(3.0) include/Sim3BA.h:407,570; MultiFrameBA.h:64
//   mean_pixel_error                include/Utils.h:102-115
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------------------------
// Dual numbers: value + N partials (ceres::Jet analogue).
// ----------------------------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
};
template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) {
  Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline bool operator>(const Jet<N>& x, const Jet<N>& y) { return x.a > y.a; }
template <int N> inline Jet<N> jsqrt(const Jet<N>& x) {
  Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a;
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <int N> inline Jet<N> jsin(const Jet<N>& x) {
  Jet<N> r; r.a = std::sin(x.a); const double d = std::cos(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <int N> inline Jet<N> jcos(const Jet<N>& x) {
  Jet<N> r; r.a = std::cos(x.a); const double d = -std::sin(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jcos(double x) { return std::cos(x); }

// ----------------------------------------------------------------------------------------------
// Ceres 1.14 AngleAxisRotatePoint, restated.  Rodrigues when theta^2 > DBL_EPSILON, otherwise the
// first-order form pt + aa x pt.  Safe for result == pt (the reference always calls it in place).
// ----------------------------------------------------------------------------------------------
template <typename T>
inline void AngleAxisRotatePoint(const T aa[3], const T pt[3], T result[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = jsqrt(theta2);
    const T costheta = jcos(theta);
    const T sintheta = jsin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                      w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    const T r0 = pt[0] * costheta + wxp[0] * sintheta + w[0] * tmp;
    const T r1 = pt[1] * costheta + wxp[1] * sintheta + w[1] * tmp;
    const T r2 = pt[2] * costheta + wxp[2] * sintheta + w[2] * tmp;
    result[0] = r0; result[1] = r1; result[2] = r2;
  } else {
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                      aa[0] * pt[1] - aa[1] * pt[0]};
    const T r0 = pt[0] + wxp[0];
    const T r1 = pt[1] + wxp[1];
    const T r2 = pt[2] + wxp[2];
    result[0] = r0; result[1] = r1; result[2] = r2;
  }
}

// Rotation matrix (row-major) with exactly the same branch behaviour: columns = rotated basis vectors.
template <typename T>
inline void AngleAxisToMatrix(const T aa[3], T R[9]) {
  for (int c = 0; c < 3; ++c) {
    T e[3] = {T(0.0), T(0.0), T(0.0)};
    e[c] = T(1.0);
    T o[3];
    AngleAxisRotatePoint(aa, e, o);
    R[0 * 3 + c] = o[0]; R[1 * 3 + c] = o[1]; R[2 * 3 + c] = o[2];
  }
}

// ----------------------------------------------------------------------------------------------
// Model
// ----------------------------------------------------------------------------------------------
struct Model {
  int V = 0, nJ = 0, nS = 0, P = 0, nL = 0;
  std::vector<double> v_template;   // [V][3]
  std::vector<double> shapedirs;    // [V][3][nS]
  std::vector<double> posedirs;     // [V][3][P]
  std::vector<double> j_regressor;  // [nJ][V]
  std::vector<double> weights;      // [V][nJ]
  std::vector<int> parent;          // [nJ]
  std::vector<int> landmark_vid;    // [nL]
  // sparse keypoint regressors over the POSED vertices (keypoint id >= nJ + nL): position = sum_i w_i posed(v_i)
  std::vector<int> kpreg_off, kpreg_vid;
  std::vector<double> kpreg_w;
  // derived
  std::vector<double> J0;      // [nJ][3]   initialJointPos = j_regressor . v_template
  std::vector<double> S;       // [3 nJ][nS] jointShapeReg   = j_regressor . shapedirs
  std::vector<double> offset;  // [nJ][3]   include/Sim3BA.h:372-392
};

void derive(Model& m) {
  const int V = m.V, nJ = m.nJ, nS = m.nS;
  m.J0.assign(nJ * 3, 0.0);
  m.S.assign(3 * nJ * nS, 0.0);
  for (int j = 0; j < nJ; ++j) {
    const double* row = &m.j_regressor[(size_t)j * V];
    for (int v = 0; v < V; ++v) {
      const double w = row[v];
      if (w == 0.0) continue;
      for (int a = 0; a < 3; ++a) {
        m.J0[j * 3 + a] += w * m.v_template[(size_t)v * 3 + a];
        for (int c = 0; c < nS; ++c)
          m.S[(size_t)(3 * j + a) * nS + c] += w * m.shapedirs[((size_t)v * 3 + a) * nS + c];
      }
    }
  }
  // include/Sim3BA.h:372-392 : root to origin, offset[j] = base[j] - base[parent[j]], offset[0] = 0
  std::vector<double> base(m.J0);
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) base[j * 3 + a] -= m.J0[a];
  m.offset.assign(nJ * 3, 0.0);
  for (int j = 1; j < nJ; ++j) {
    const int pj = m.parent[j];
    for (int a = 0; a < 3; ++a)
      m.offset[j * 3 + a] = pj >= 0 ? base[j * 3 + a] - base[pj * 3 + a] : base[j * 3 + a];
  }
}

// ----------------------------------------------------------------------------------------------
// Keypoint functor.  id < nJ : the reference's FK-chain joint (ReprojCost / ReprojCostShape).
//                    id >= nJ: vertex landmark (id - nJ) through the SMPL forward of that vertex.
// params[] in the reference's block order: [scale(1), rootAA(3), rootT(3), jointAA[1..nJ-1](3 each),
// (beta(nS))]  — include/Sim3BA.h:36-40,128-133.
// ----------------------------------------------------------------------------------------------
struct KpCtx {
  const Model* m;
  int id;
  double u_obs, v_obs, fx, fy, cx, cy;
  const double* R0;  // 3x3 row-major
  bool use_shape;    // jointShapeReg passed (ReprojCostShape with betaShape > 0) or not
  bool pose_blend;   // landmarks only
};

template <typename T>
void fk_joint_body(const KpCtx& c, T const* const* params, T pos[3]) {
  const Model& m = *c.m;
  const int nJ = m.nJ, nS = m.nS, jid = c.id;
  auto jointAA = [&](int j) -> const T* { return params[3 + (j - 1)]; };
  // include/Sim3BA.h:142
  for (int a = 0; a < 3; ++a) pos[a] = T(m.offset[jid * 3 + a]);
  // include/Sim3BA.h:145-170
  if (c.use_shape && nS > 0) {
    const T* w = params[3 + (nJ - 1)];
    T dj[3] = {T(0.0), T(0.0), T(0.0)}, dp[3] = {T(0.0), T(0.0), T(0.0)};
    for (int k = 0; k < nS; ++k) {
      for (int a = 0; a < 3; ++a) dj[a] += T(m.S[(size_t)(3 * jid + a) * nS + k]) * w[k];
      if (m.parent[jid] >= 0) {
        const int pj = m.parent[jid];
        for (int a = 0; a < 3; ++a) dp[a] += T(m.S[(size_t)(3 * pj + a) * nS + k]) * w[k];
      }
    }
    for (int a = 0; a < 3; ++a) pos[a] += (dj[a] - dp[a]);
  }
  // include/Sim3BA.h:173-207
  int cur = jid;
  while (m.parent[cur] != -1 && m.parent[cur] != 0) {
    const int p = m.parent[cur];
    AngleAxisRotatePoint(jointAA(p), pos, pos);
    if (c.use_shape && nS > 0) {
      const T* w = params[3 + (nJ - 1)];
      T d1[3] = {T(0.0), T(0.0), T(0.0)}, d2[3] = {T(0.0), T(0.0), T(0.0)};
      for (int k = 0; k < nS; ++k) {
        for (int a = 0; a < 3; ++a) d1[a] += T(m.S[(size_t)(3 * p + a) * nS + k]) * w[k];
        const int pp = m.parent[p];
        if (pp >= 0)
          for (int a = 0; a < 3; ++a) d2[a] += T(m.S[(size_t)(3 * pp + a) * nS + k]) * w[k];
      }
      for (int a = 0; a < 3; ++a) pos[a] += T(m.offset[p * 3 + a]) + (d1[a] - d2[a]);
    } else {
      for (int a = 0; a < 3; ++a) pos[a] += T(m.offset[p * 3 + a]);
    }
    cur = p;
  }
}

// Body-frame skeleton in T: A[j] (rotation root-frame -> joint j, A[0] = I), Pj[j] posed joint positions
// (root at origin), Jc[j] rest joints centred on the root (shape-dependent when use_shape).
template <typename T>
void body_skeleton(const KpCtx& c, T const* const* params, std::vector<T>& A, std::vector<T>& Pj,
                   std::vector<T>& Jc, std::vector<T>& Rl) {
  const Model& m = *c.m;
  const int nJ = m.nJ, nS = m.nS;
  A.assign(nJ * 9, T(0.0)); Pj.assign(nJ * 3, T(0.0)); Jc.assign(nJ * 3, T(0.0)); Rl.assign(nJ * 9, T(0.0));
  std::vector<T> Jb(nJ * 3);
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) {
      T x = T(m.J0[j * 3 + a]);
      if (c.use_shape && nS > 0) {
        const T* w = params[3 + (nJ - 1)];
        for (int k = 0; k < nS; ++k) x += T(m.S[(size_t)(3 * j + a) * nS + k]) * w[k];
      }
      Jb[j * 3 + a] = x;
    }
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) Jc[j * 3 + a] = Jb[j * 3 + a] - Jb[a];
  A[0] = A[4] = A[8] = T(1.0);
  Rl[0] = Rl[4] = Rl[8] = T(1.0);
  for (int j = 1; j < nJ; ++j) {
    const int p = m.parent[j];
    T R[9];
    AngleAxisToMatrix(params[3 + (j - 1)], R);
    for (int i = 0; i < 9; ++i) Rl[j * 9 + i] = R[i];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        T s = T(0.0);
        for (int k = 0; k < 3; ++k) s += A[p * 9 + r * 3 + k] * R[k * 3 + cc];
        A[j * 9 + r * 3 + cc] = s;
      }
    for (int r = 0; r < 3; ++r) {
      T s = Pj[p * 3 + r];
      for (int k = 0; k < 3; ++k) s += A[p * 9 + r * 3 + k] * (Jc[j * 3 + k] - Jc[p * 3 + k]);
      Pj[j * 3 + r] = s;
    }
  }
}

// One vertex of the SMPL forward in the body frame (root at origin, root rotation = identity).
template <typename T>
void vertex_body(const KpCtx& c, T const* const* params, int vid, const std::vector<T>& A,
                 const std::vector<T>& Pj, const std::vector<T>& Jc, const std::vector<T>& Rl,
                 T x[3]) {
  const Model& m = *c.m;
  const int nJ = m.nJ, nS = m.nS, P = m.P;
  T vp[3];
  for (int a = 0; a < 3; ++a) {
    T s = T(m.v_template[(size_t)vid * 3 + a] - m.J0[a]);
    if (c.use_shape && nS > 0) {
      const T* w = params[3 + (nJ - 1)];
      for (int k = 0; k < nS; ++k)
        s += T(m.shapedirs[((size_t)vid * 3 + a) * nS + k] - m.S[(size_t)a * nS + k]) * w[k];
    }
    if (c.pose_blend && P > 0) {
      const double* pd = &m.posedirs[((size_t)vid * 3 + a) * P];
      for (int j = 1; j < nJ; ++j)
        for (int e = 0; e < 9; ++e) {
          T f = Rl[j * 9 + e];
          if (e == 0 || e == 4 || e == 8) f = f - T(1.0);
          s += T(pd[9 * (j - 1) + e]) * f;
        }
    }
    vp[a] = s;
  }
  for (int a = 0; a < 3; ++a) x[a] = T(0.0);
  for (int j = 0; j < nJ; ++j) {
    const double w = m.weights[(size_t)vid * nJ + j];
    if (w == 0.0) continue;
    for (int r = 0; r < 3; ++r) {
      T s = Pj[j * 3 + r];
      for (int k = 0; k < 3; ++k) s += A[j * 9 + r * 3 + k] * (vp[k] - Jc[j * 3 + k]);
      x[r] += T(w) * s;
    }
  }
}

template <typename T>
bool kp_functor(const KpCtx& c, T const* const* params, T* residuals) {
  const Model& m = *c.m;
  const T* scale = params[0];
  const T* rootAA = params[1];
  const T* rootT = params[2];
  T pos[3];
  if (c.id < m.nJ) {
    fk_joint_body(c, params, pos);
  } else {
    std::vector<T> A, Pj, Jc, Rl;
    body_skeleton(c, params, A, Pj, Jc, Rl);
    if (c.id < m.nJ + m.nL) {
      vertex_body(c, params, m.landmark_vid[c.id - m.nJ], A, Pj, Jc, Rl, pos);
    } else {   // sparse regressor row over the posed vertices
      const int r = c.id - m.nJ - m.nL;
      for (int a = 0; a < 3; ++a) pos[a] = T(0.0);
      for (int e = m.kpreg_off[r]; e < m.kpreg_off[r + 1]; ++e) {
        T pv[3];
        vertex_body(c, params, m.kpreg_vid[e], A, Pj, Jc, Rl, pv);
        for (int a = 0; a < 3; ++a) pos[a] += T(m.kpreg_w[e]) * pv[a];
      }
    }
  }
  // include/Sim3BA.h:210-213
  T cam[3];
  for (int r = 0; r < 3; ++r)
    cam[r] = T(c.R0[r * 3 + 0]) * pos[0] + T(c.R0[r * 3 + 1]) * pos[1] + T(c.R0[r * 3 + 2]) * pos[2];
  // include/Sim3BA.h:216-219
  AngleAxisRotatePoint(rootAA, cam, cam);
  for (int r = 0; r < 3; ++r) cam[r] = (*scale) * cam[r] + rootT[r];
  // include/Sim3BA.h:222-225 (no guard on Z)
  const T u = T(c.fx) * cam[0] / cam[2] + T(c.cx);
  const T v = T(c.fy) * cam[1] / cam[2] + T(c.cy);
  residuals[0] = u - T(c.u_obs);
  residuals[1] = v - T(c.v_obs);
  return true;
}

// DynamicAutoDiffCostFunction<...,4>::Evaluate restated: ceil(ncols/4) passes, 4 seeded partials each.
void kp_autodiff(const KpCtx& c, const double* x /*ncols, packed*/, int ncols, double r[2], double* J) {
  const int nJ = c.m->nJ, nS = c.m->nS;
  const int nblocks = 3 + (nJ - 1) + ((ncols > 7 + 3 * (nJ - 1)) ? 1 : 0);
  std::vector<int> boff(nblocks + 1);
  boff[0] = 0; boff[1] = 1; boff[2] = 4; boff[3] = 7;
  for (int j = 1; j < nJ; ++j) boff[3 + j] = 7 + 3 * j;
  if (nblocks == 3 + nJ) boff[nblocks] = boff[nblocks - 1] + nS;
  typedef Jet<4> J4;
  std::vector<J4> xj(ncols);
  std::vector<const J4*> pp(nblocks);
  if (!J) {
    std::vector<const double*> pd(nblocks);
    for (int b = 0; b < nblocks; ++b) pd[b] = x + boff[b];
    kp_functor<double>(c, pd.data(), r);
    return;
  }
  for (int start = 0; start < ncols; start += 4) {
    for (int i = 0; i < ncols; ++i) {
      xj[i] = J4(x[i]);
      if (i >= start && i < start + 4) xj[i].v[i - start] = 1.0;
    }
    for (int b = 0; b < nblocks; ++b) pp[b] = xj.data() + boff[b];
    J4 rr[2];
    kp_functor<J4>(c, pp.data(), rr);
    r[0] = rr[0].a; r[1] = rr[1].a;
    for (int k = 0; k < 4 && start + k < ncols; ++k) {
      J[0 * ncols + start + k] = rr[0].v[k];
      J[1 * ncols + start + k] = rr[1].v[k];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Analytic path.  Plain doubles, explicit dR/da.
// ----------------------------------------------------------------------------------------------
inline void mat3mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) t[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  std::memcpy(C, t, sizeof(t));
}
inline void mat3vec(const double* A, const double* x, double* y) {
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3] * x[0] + A[r * 3 + 1] * x[1] + A[r * 3 + 2] * x[2];
  y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
inline void skew(const double* v, double* K) {
  K[0] = 0; K[1] = -v[2]; K[2] = v[1];
  K[3] = v[2]; K[4] = 0; K[5] = -v[0];
  K[6] = -v[1]; K[7] = v[0]; K[8] = 0;
}

// R(a) and dR/da_c (c = 0..2), both branches of AngleAxisRotatePoint.
// Rodrigues: R = cos I + sin [w]x + (1-cos) w w^T, w = a/theta.
void rodrigues_with_grad(const double a[3], double R[9], double dR[3][9]) {
  const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (th2 > std::numeric_limits<double>::epsilon()) {
    const double th = std::sqrt(th2), ct = std::cos(th), st = std::sin(th);
    const double omc = 2.0 * std::sin(0.5 * th) * std::sin(0.5 * th);  // 1 - cos, cancellation-free
    const double w[3] = {a[0] / th, a[1] / th, a[2] / th};
    double K[9];
    skew(w, K);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        R[r * 3 + c] = (r == c ? ct : 0.0) + st * K[r * 3 + c] + omc * w[r] * w[c];
    for (int k = 0; k < 3; ++k) {
      // d theta / d a_k = w_k ; d w / d a_k = (e_k - w w_k) / theta
      double dw[3];
      for (int i = 0; i < 3; ++i) dw[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) / th;
      double dK[9];
      skew(dw, dK);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          dR[k][r * 3 + c] = (r == c ? -st * w[k] : 0.0) + ct * w[k] * K[r * 3 + c] + st * dK[r * 3 + c] +
                             st * w[k] * w[r] * w[c] + omc * (dw[r] * w[c] + w[r] * dw[c]);
    }
  } else {
    double K[9];
    skew(a, K);
    for (int i = 0; i < 9; ++i) R[i] = K[i] + ((i % 4 == 0) ? 1.0 : 0.0);
    for (int k = 0; k < 3; ++k) {
      double e[3] = {0, 0, 0};
      e[k] = 1.0;
      skew(e, dR[k]);
    }
  }
}

struct FrameGeom {  // per-frame quantities shared by all keypoints of the frame
  std::vector<double> R, dR;    // [nJ][9], [nJ][3][9]  (index 0 = root angle-axis)
  std::vector<double> o;        // [nJ][3]  o_j(beta)  (chain offsets with the reference's shape terms)
  std::vector<double> A;        // [nJ][9]  A_j = prod of chain rotations incl. j  (A_0 = I)
  std::vector<double> Pj;       // [nJ][3]  posed joints, root at origin
  std::vector<double> Jc;       // [nJ][3]  rest joints centred at root (beta-dependent)
};

void frame_geom(const Model& m, const double* x, const double* beta, bool use_shape, FrameGeom& g) {
  const int nJ = m.nJ, nS = m.nS;
  g.R.assign(nJ * 9, 0); g.dR.assign(nJ * 27, 0); g.o.assign(nJ * 3, 0);
  g.A.assign(nJ * 9, 0); g.Pj.assign(nJ * 3, 0); g.Jc.assign(nJ * 3, 0);
  for (int j = 0; j < nJ; ++j) {
    const double* aa = (j == 0) ? (x + 1) : (x + 7 + 3 * (j - 1));
    double dR[3][9];
    rodrigues_with_grad(aa, &g.R[j * 9], dR);
    for (int k = 0; k < 3; ++k) std::memcpy(&g.dR[(j * 3 + k) * 9], dR[k], 9 * sizeof(double));
  }
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) {
      double Jb = m.J0[j * 3 + a] - m.J0[a];
      if (use_shape)
        for (int k = 0; k < nS; ++k) Jb += (m.S[(size_t)(3 * j + a) * nS + k] - m.S[(size_t)a * nS + k]) * beta[k];
      g.Jc[j * 3 + a] = Jb;
    }
  for (int j = 1; j < nJ; ++j) {
    const int p = m.parent[j];
    for (int a = 0; a < 3; ++a) {
      double s = m.offset[j * 3 + a];
      if (use_shape)
        for (int k = 0; k < nS; ++k)
          s += (m.S[(size_t)(3 * j + a) * nS + k] - (p >= 0 ? m.S[(size_t)(3 * p + a) * nS + k] : 0.0)) * beta[k];
      g.o[j * 3 + a] = s;
    }
  }
  g.A[0] = g.A[4] = g.A[8] = 1.0;
  for (int j = 1; j < nJ; ++j) {
    const int p = m.parent[j];
    mat3mul(&g.A[p * 9], &g.R[j * 9], &g.A[j * 9]);
    double t[3];
    mat3vec(&g.A[p * 9], &g.o[j * 3], t);
    for (int a = 0; a < 3; ++a) g.Pj[j * 3 + a] = g.Pj[p * 3 + a] + t[a];
  }
}

// residual + analytic Jacobian (2 x ncols, row-major) of one keypoint.
void kp_analytic(const KpCtx& c, const FrameGeom& g, const double* x, const double* beta, int ncols,
                 double r[2], double* J) {
  const Model& m = *c.m;
  const int nJ = m.nJ, nS = m.nS, P = m.P;
  const bool shape_cols = ncols > 7 + 3 * (nJ - 1);
  double q[3] = {0, 0, 0};
  std::vector<double> dq((size_t)3 * ncols, 0.0);  // d q / d column (body frame)
  auto DQ = [&](int row, int col) -> double& { return dq[(size_t)row * ncols + col]; };

  if (c.id < nJ) {
    const int jid = c.id;
    if (jid == 0 || m.parent[jid] < 0) {
      // include/Sim3BA.h:142-170 with no chain: q = offset[0] + S_0 beta (no parent term)
      for (int a = 0; a < 3; ++a) {
        q[a] = m.offset[jid * 3 + a];
        if (c.use_shape)
          for (int k = 0; k < nS; ++k) {
            q[a] += m.S[(size_t)(3 * jid + a) * nS + k] * beta[k];
            if (shape_cols) DQ(a, 7 + 3 * (nJ - 1) + k) = m.S[(size_t)(3 * jid + a) * nS + k];
          }
      }
    } else {
      for (int a = 0; a < 3; ++a) q[a] = g.Pj[jid * 3 + a];
      // pose columns: every proper ancestor k != root rotates (q - P_k)
      for (int k = m.parent[jid]; k > 0; k = m.parent[k]) {
        const int pk = m.parent[k];
        double rel[3] = {q[0] - g.Pj[k * 3], q[1] - g.Pj[k * 3 + 1], q[2] - g.Pj[k * 3 + 2]};
        double loc[3];  // local vector below joint k: R_k^T A_pk^T rel  (exact inverse for Rodrigues)
        {
          double Ak[9];
          mat3mul(&g.A[pk * 9], &g.R[k * 9], Ak);
          for (int i = 0; i < 3; ++i) loc[i] = Ak[i] * rel[0] + Ak[3 + i] * rel[1] + Ak[6 + i] * rel[2];
        }
        for (int cc = 0; cc < 3; ++cc) {
          double t[3];
          mat3vec(&g.dR[(k * 3 + cc) * 9], loc, t);
          mat3vec(&g.A[pk * 9], t, t);
          for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (k - 1) + cc) = t[a];
        }
      }
      if (c.use_shape && shape_cols) {
        // d q / d beta = sum over chain links A_par(c) (S_c - S_par(c))
        for (int cnode = jid; cnode > 0; cnode = m.parent[cnode]) {
          const int pc = m.parent[cnode];
          for (int k = 0; k < nS; ++k) {
            double d[3], t[3];
            for (int a = 0; a < 3; ++a)
              d[a] = m.S[(size_t)(3 * cnode + a) * nS + k] - m.S[(size_t)(3 * pc + a) * nS + k];
            mat3vec(&g.A[pc * 9], d, t);
            for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (nJ - 1) + k) += t[a];
          }
        }
      }
    }
  } else {
    // vertex keypoint through blend + LBS, body frame: one vertex (landmark) or a sparse regressor row over posed vertices
    const bool is_reg = c.id >= nJ + m.nL;
    const int e0 = is_reg ? m.kpreg_off[c.id - nJ - m.nL] : 0, e1 = is_reg ? m.kpreg_off[c.id - nJ - m.nL + 1] : 1;
    for (int ee = e0; ee < e1; ++ee) {
      const int vid = is_reg ? m.kpreg_vid[ee] : m.landmark_vid[c.id - nJ];
      const double coef = is_reg ? m.kpreg_w[ee] : 1.0;
      double vp[3];
      for (int a = 0; a < 3; ++a) {
        double s = m.v_template[(size_t)vid * 3 + a] - m.J0[a];
        if (c.use_shape)
          for (int k = 0; k < nS; ++k)
            s += (m.shapedirs[((size_t)vid * 3 + a) * nS + k] - m.S[(size_t)a * nS + k]) * beta[k];
        if (c.pose_blend && P > 0) {
          const double* pd = &m.posedirs[((size_t)vid * 3 + a) * P];
          for (int j = 1; j < nJ; ++j)
            for (int e = 0; e < 9; ++e)
              s += pd[9 * (j - 1) + e] * (g.R[j * 9 + e] - ((e % 4 == 0) ? 1.0 : 0.0));
        }
        vp[a] = s;
      }
      double Ablend[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int j = 0; j < nJ; ++j) {
        const double w = m.weights[(size_t)vid * nJ + j];
        if (w == 0.0) continue;
        double loc[3] = {vp[0] - g.Jc[j * 3], vp[1] - g.Jc[j * 3 + 1], vp[2] - g.Jc[j * 3 + 2]};
        double xj[3];
        mat3vec(&g.A[j * 9], loc, xj);
        for (int a = 0; a < 3; ++a) { xj[a] += g.Pj[j * 3 + a]; q[a] += coef * w * xj[a]; }
        for (int i = 0; i < 9; ++i) Ablend[i] += w * g.A[j * 9 + i];
        // pose columns: joint k in {j and its ancestors}, k != root, rotates (x_j - P_k)
        for (int k = j; k > 0; k = m.parent[k]) {
          const int pk = m.parent[k];
          double rel[3] = {xj[0] - g.Pj[k * 3], xj[1] - g.Pj[k * 3 + 1], xj[2] - g.Pj[k * 3 + 2]};
          double Ak[9], lk[3];
          mat3mul(&g.A[pk * 9], &g.R[k * 9], Ak);
          for (int i = 0; i < 3; ++i) lk[i] = Ak[i] * rel[0] + Ak[3 + i] * rel[1] + Ak[6 + i] * rel[2];
          for (int cc = 0; cc < 3; ++cc) {
            double t[3];
            mat3vec(&g.dR[(k * 3 + cc) * 9], lk, t);
            mat3vec(&g.A[pk * 9], t, t);
            for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (k - 1) + cc) += coef * w * t[a];
          }
        }
        if (c.use_shape && shape_cols) {
          // d/d beta of A_j (vp - Jc_j) + P_j
          for (int k = 0; k < nS; ++k) {
            double d[3], t[3];
            for (int a = 0; a < 3; ++a)
              d[a] = (m.shapedirs[((size_t)vid * 3 + a) * nS + k] - m.S[(size_t)a * nS + k]) -
                     (m.S[(size_t)(3 * j + a) * nS + k] - m.S[(size_t)a * nS + k]);
            mat3vec(&g.A[j * 9], d, t);
            for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (nJ - 1) + k) += coef * w * t[a];
            for (int cnode = j; cnode > 0; cnode = m.parent[cnode]) {
              const int pc = m.parent[cnode];
              for (int a = 0; a < 3; ++a)
                d[a] = m.S[(size_t)(3 * cnode + a) * nS + k] - m.S[(size_t)(3 * pc + a) * nS + k];
              mat3vec(&g.A[pc * 9], d, t);
              for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (nJ - 1) + k) += coef * w * t[a];
            }
          }
        }
      }
      if (c.pose_blend && P > 0) {
        // d vp / d a_k,c = posedirs[:, 9(k-1):9k] . vec(dR_k,c), carried by the blended rotation
        for (int k = 1; k < nJ; ++k)
          for (int cc = 0; cc < 3; ++cc) {
            double d[3], t[3];
            for (int a = 0; a < 3; ++a) {
              const double* pd = &m.posedirs[((size_t)vid * 3 + a) * P + 9 * (k - 1)];
              double s = 0;
              for (int e = 0; e < 9; ++e) s += pd[e] * g.dR[(k * 3 + cc) * 9 + e];
              d[a] = s;
            }
            mat3vec(Ablend, d, t);
            for (int a = 0; a < 3; ++a) DQ(a, 7 + 3 * (k - 1) + cc) += coef * t[a];
          }
      }
    }
  }

  // camera: y = R0 q ; z = R(a_root) y ; X = s z + t
  double y[3], z[3], X[3];
  mat3vec(c.R0, q, y);
  mat3vec(&g.R[0], y, z);
  const double s = x[0];
  for (int a = 0; a < 3; ++a) X[a] = s * z[a] + x[4 + a];
  r[0] = c.fx * X[0] / X[2] + c.cx - c.u_obs;
  r[1] = c.fy * X[1] / X[2] + c.cy - c.v_obs;
  if (!J) return;
  const double iz = 1.0 / X[2];
  const double dpi[6] = {c.fx * iz, 0.0, -c.fx * X[0] * iz * iz, 0.0, c.fy * iz, -c.fy * X[1] * iz * iz};
  double M[9];
  mat3mul(&g.R[0], c.R0, M);
  for (int i = 0; i < 9; ++i) M[i] *= s;
  std::vector<double> dX((size_t)3 * ncols, 0.0);
  for (int a = 0; a < 3; ++a) dX[(size_t)a * ncols + 0] = z[a];
  for (int cc = 0; cc < 3; ++cc) {
    double t[3];
    mat3vec(&g.dR[(0 * 3 + cc) * 9], y, t);
    for (int a = 0; a < 3; ++a) dX[(size_t)a * ncols + 1 + cc] = s * t[a];
    dX[(size_t)cc * ncols + 4 + cc] = 1.0;
  }
  for (int col = 7; col < ncols; ++col) {
    const double d[3] = {DQ(0, col), DQ(1, col), DQ(2, col)};
    if (d[0] == 0.0 && d[1] == 0.0 && d[2] == 0.0) continue;
    double t[3];
    mat3vec(M, d, t);
    for (int a = 0; a < 3; ++a) dX[(size_t)a * ncols + col] = t[a];
  }
  for (int col = 0; col < ncols; ++col)
    for (int rr = 0; rr < 2; ++rr)
      J[(size_t)rr * ncols + col] = dpi[rr * 3 + 0] * dX[col] + dpi[rr * 3 + 1] * dX[(size_t)ncols + col] +
                                    dpi[rr * 3 + 2] * dX[(size_t)2 * ncols + col];
}

// ----------------------------------------------------------------------------------------------
// GMM pose prior (sxyu/avatar GaussianMixture, recalled; SMPLify MaxMixturePrior algorithm)
// ----------------------------------------------------------------------------------------------
struct Gmm {
  int K = 0, D = 0;
  std::vector<double> weight, mean;        // [K], [K][D]
  std::vector<double> prec_cho;            // [K][D][D]  lower L, precision = L L^T
  std::vector<double> log_w;               // [K]  -log of the normalised mixture constant
  double resid_scale = std::sqrt(0.5);     // recalled upstream scaling; option (Q10)
};

bool cholesky_lower(std::vector<double>& A, int n) {  // in place, row-major; upper part zeroed
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (d <= 0.0) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
    for (int i = 0; i < j; ++i) A[(size_t)i * n + j] = 0.0;
  }
  return true;
}

}  // namespace

// ================================================================================================
// C interface (ctypes).  All matrices row-major doubles.
// ================================================================================================
extern "C" {

void* oracle_model_create(int V, int nJ, int nS, int P, const double* v_template, const double* shapedirs,
                          const double* posedirs, const double* j_regressor, const double* weights,
                          const int* parent, int nL, const int* landmark_vid) {
  Model* m = new Model();
  m->V = V; m->nJ = nJ; m->nS = nS; m->P = posedirs ? P : 0; m->nL = nL;
  m->v_template.assign(v_template, v_template + (size_t)V * 3);
  m->shapedirs.assign(shapedirs, shapedirs + (size_t)V * 3 * nS);
  if (posedirs) m->posedirs.assign(posedirs, posedirs + (size_t)V * 3 * P);
  m->j_regressor.assign(j_regressor, j_regressor + (size_t)nJ * V);
  m->weights.assign(weights, weights + (size_t)V * nJ);
  m->parent.assign(parent, parent + nJ);
  if (nL > 0) m->landmark_vid.assign(landmark_vid, landmark_vid + nL);
  derive(*m);
  return m;
}
void oracle_model_destroy(void* h) { delete static_cast<Model*>(h); }
// sparse keypoint regressors (CSR): keypoint id nJ + nL + r = sum_i weight_i posed(vertex_i)
void oracle_model_set_kp_regressors(void* h, int n, const int* offset, const int* vid, const double* weight) {
  Model* m = static_cast<Model*>(h);
  m->kpreg_off.assign(offset, offset + n + 1);
  m->kpreg_vid.assign(vid, vid + offset[n]);
  m->kpreg_w.assign(weight, weight + offset[n]);
}

void oracle_model_derived(void* h, double* J0, double* S, double* offset) {
  Model* m = static_cast<Model*>(h);
  if (J0) std::memcpy(J0, m->J0.data(), m->J0.size() * sizeof(double));
  if (S) std::memcpy(S, m->S.data(), m->S.size() * sizeof(double));
  if (offset) std::memcpy(offset, m->offset.data(), m->offset.size() * sizeof(double));
}

// One residual block.  x = packed [scale, rootAA, rootT, jointAA[1..nJ-1], (beta)], ncols = 76 or 86.
// mode 0 = analytic, 1 = stride-4 dual-number autodiff (reference-like).
void oracle_kp_block(void* h, int id, double u, double v, const double* intr, const double* R0,
                     int use_shape, int pose_blend, const double* x, int ncols, int mode, double* r,
                     double* J) {
  Model* m = static_cast<Model*>(h);
  KpCtx c{m, id, u, v, intr[0], intr[1], intr[2], intr[3], R0, use_shape != 0, pose_blend != 0};
  if (mode == 1) {
    kp_autodiff(c, x, ncols, r, J);
  } else {
    FrameGeom g;
    const double* beta = (ncols > 7 + 3 * (m->nJ - 1)) ? x + 7 + 3 * (m->nJ - 1) : nullptr;
    std::vector<double> zero(m->nS, 0.0);
    frame_geom(*m, x, beta ? beta : zero.data(), use_shape != 0 && beta, g);
    KpCtx c2 = c;
    c2.use_shape = use_shape != 0 && beta;
    kp_analytic(c2, g, x, beta ? beta : zero.data(), ncols, r, J);
  }
}

// Batched reprojection residuals/Jacobians for F frames.
//   params [F][76]; beta [nS] (beta_stride 0) or [F][nS] (beta_stride nS); kp_offset [F+1]
//   r [2 Ktot]; J [2 Ktot][ncols] (NULL -> residual only); ncols = 76 (no beta block) or 76 + nS
void oracle_evaluate_batch(void* h, int F, const int* kp_offset, const int* kp_id, const double* kp_uv,
                           const double* intr, const double* R0 /*[F][9]*/, int ncols, int use_shape,
                           int beta_stride, int pose_blend, const double* params, const double* beta,
                           int mode, int nthreads, double* r, double* J) {
  Model* m = static_cast<Model*>(h);
  const int npose = 7 + 3 * (m->nJ - 1);
  const bool has_beta = ncols > npose;   // beta block present (Q12: may be present yet unused)
  if (!has_beta) use_shape = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  if (mode == 1) {
    // reference-like: threads over residual blocks (Ceres evaluates blocks concurrently)
    const int Ktot = kp_offset[F];
    std::vector<int> frame_of(Ktot);
    for (int f = 0; f < F; ++f)
      for (int k = kp_offset[f]; k < kp_offset[f + 1]; ++k) frame_of[k] = f;
#pragma omp parallel for schedule(dynamic, 8)
    for (int k = 0; k < Ktot; ++k) {
      const int f = frame_of[k];
      std::vector<double> x(ncols);
      std::memcpy(x.data(), params + (size_t)f * npose, npose * sizeof(double));
      if (has_beta) std::memcpy(x.data() + npose, beta + (size_t)f * beta_stride, m->nS * sizeof(double));
      KpCtx c{m, kp_id[k], kp_uv[2 * k], kp_uv[2 * k + 1], intr[0], intr[1], intr[2], intr[3],
              R0 + (size_t)f * 9, use_shape != 0, pose_blend != 0};
      kp_autodiff(c, x.data(), ncols, r + 2 * (size_t)k, J ? J + (size_t)2 * k * ncols : nullptr);
    }
  } else {
#pragma omp parallel for schedule(dynamic, 4)
    for (int f = 0; f < F; ++f) {
      FrameGeom g;
      std::vector<double> zero(m->nS, 0.0);
      const double* b = use_shape ? beta + (size_t)f * beta_stride : zero.data();
      const double* x = params + (size_t)f * npose;
      frame_geom(*m, x, b, use_shape != 0, g);
      for (int k = kp_offset[f]; k < kp_offset[f + 1]; ++k) {
        KpCtx c{m, kp_id[k], kp_uv[2 * k], kp_uv[2 * k + 1], intr[0], intr[1], intr[2], intr[3],
                R0 + (size_t)f * 9, use_shape != 0, pose_blend != 0};
        kp_analytic(c, g, x, b, ncols, r + 2 * (size_t)k, J ? J + (size_t)2 * k * ncols : nullptr);
      }
    }
  }
}

// SMPL forward for one frame ("Avatar::update()" + Sim3).  x = 76 packed frame parameters.
// Outputs camera-frame joints [nJ][3] and cloud [V][3]:  X = s R(rootAA) R0 x_body + t.
static void forward_one(Model* m, const double* x, const double* beta, const double* R0, int use_shape,
                        int pose_blend, int nthreads, bool par_verts, double* joints, double* cloud) {
  const int nJ = m->nJ, nS = m->nS, V = m->V, P = m->P;
  FrameGeom g;
  std::vector<double> zero(nS, 0.0);
  const double* b = use_shape ? beta : zero.data();
  frame_geom(*m, x, b, use_shape != 0, g);
  // o_j(beta) = Jc_j - Jc_par(j), so the reference's chain joints ARE the SMPL posed joints (root at 0)
  const std::vector<double>& Pj = g.Pj;
  double M[9];
  mat3mul(&g.R[0], R0, M);
  const double s = x[0];
  for (int i = 0; i < 9; ++i) M[i] *= s;
  if (joints)
    for (int j = 0; j < nJ; ++j) {
      double t[3];
      mat3vec(M, &Pj[j * 3], t);
      for (int a = 0; a < 3; ++a) joints[j * 3 + a] = t[a] + x[4 + a];
    }
  if (!cloud) return;
#ifdef _OPENMP
  if (par_verts && nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static) if (par_verts)
  for (int v = 0; v < V; ++v) {
    double vp[3];
    for (int a = 0; a < 3; ++a) {
      double sacc = m->v_template[(size_t)v * 3 + a] - m->J0[a];
      if (use_shape)
        for (int k = 0; k < nS; ++k)
          sacc += (m->shapedirs[((size_t)v * 3 + a) * nS + k] - m->S[(size_t)a * nS + k]) * b[k];
      if (pose_blend && P > 0) {
        const double* pd = &m->posedirs[((size_t)v * 3 + a) * P];
        for (int j = 1; j < nJ; ++j)
          for (int e = 0; e < 9; ++e) sacc += pd[9 * (j - 1) + e] * (g.R[j * 9 + e] - ((e % 4 == 0) ? 1.0 : 0.0));
      }
      vp[a] = sacc;
    }
    double q[3] = {0, 0, 0};
    for (int j = 0; j < nJ; ++j) {
      const double w = m->weights[(size_t)v * nJ + j];
      if (w == 0.0) continue;
      double loc[3] = {vp[0] - g.Jc[j * 3], vp[1] - g.Jc[j * 3 + 1], vp[2] - g.Jc[j * 3 + 2]};
      double t[3];
      mat3vec(&g.A[j * 9], loc, t);
      for (int a = 0; a < 3; ++a) q[a] += w * (t[a] + Pj[j * 3 + a]);
    }
    double t[3];
    mat3vec(M, q, t);
    for (int a = 0; a < 3; ++a) cloud[(size_t)v * 3 + a] = t[a] + x[4 + a];
  }
}

void oracle_forward(void* h, const double* x, const double* beta, const double* R0, int use_shape,
                    int pose_blend, int nthreads, double* joints, double* cloud) {
  forward_one(static_cast<Model*>(h), x, beta, R0, use_shape, pose_blend, nthreads, true, joints, cloud);
}

// F frames, threads over frames (the CPU baseline of the batched forward).  beta_stride 0 = shared.
void oracle_forward_batch(void* h, int F, const double* params, const double* beta, int beta_stride,
                          const double* R0, int use_shape, int pose_blend, int nthreads, double* joints,
                          double* cloud) {
  Model* m = static_cast<Model*>(h);
  const int npose = 7 + 3 * (m->nJ - 1);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < F; ++f)
    forward_one(m, params + (size_t)f * npose, beta + (size_t)f * beta_stride, R0 + (size_t)f * 9, use_shape,
                pose_blend, 1, false, joints ? joints + (size_t)f * m->nJ * 3 : nullptr,
                cloud ? cloud + (size_t)f * m->V * 3 : nullptr);
}

// ---- GMM -------------------------------------------------------------------------------------
void* oracle_gmm_create(int K, int D, const double* weights, const double* means, const double* covs,
                        double resid_scale) {
  Gmm* g = new Gmm();
  g->K = K; g->D = D; g->resid_scale = resid_scale;
  g->weight.assign(weights, weights + K);
  g->mean.assign(means, means + (size_t)K * D);
  g->prec_cho.assign((size_t)K * D * D, 0.0);
  g->log_w.assign(K, 0.0);
  std::vector<double> half_logdet(K);
  for (int k = 0; k < K; ++k) {
    // precision = cov^{-1} via Cholesky of cov: cov = C C^T, prec = C^{-T} C^{-1}
    std::vector<double> C(covs + (size_t)k * D * D, covs + (size_t)(k + 1) * D * D);
    if (!cholesky_lower(C, D)) { delete g; return nullptr; }
    double ld = 0;
    for (int i = 0; i < D; ++i) ld += std::log(C[(size_t)i * D + i]);
    half_logdet[k] = ld;  // = 0.5 log det cov
    // Cinv (lower)
    std::vector<double> Ci((size_t)D * D, 0.0);
    for (int c = 0; c < D; ++c) {
      Ci[(size_t)c * D + c] = 1.0 / C[(size_t)c * D + c];
      for (int r = c + 1; r < D; ++r) {
        double s = 0;
        for (int t = c; t < r; ++t) s += C[(size_t)r * D + t] * Ci[(size_t)t * D + c];
        Ci[(size_t)r * D + c] = -s / C[(size_t)r * D + r];
      }
    }
    std::vector<double> Pm((size_t)D * D, 0.0);  // prec = Ci^T Ci
    for (int r = 0; r < D; ++r)
      for (int c = 0; c <= r; ++c) {
        double s = 0;
        for (int t = r; t < D; ++t) s += Ci[(size_t)t * D + r] * Ci[(size_t)t * D + c];
        Pm[(size_t)r * D + c] = s; Pm[(size_t)c * D + r] = s;
      }
    if (!cholesky_lower(Pm, D)) { delete g; return nullptr; }
    std::memcpy(&g->prec_cho[(size_t)k * D * D], Pm.data(), (size_t)D * D * sizeof(double));
  }
  // SMPLify MaxMixturePrior: w'_k = w_k / ((2 pi)^{D/2} sqrt(det cov_k) / min_j sqrt(det cov_j))
  const double min_hld = *std::min_element(half_logdet.begin(), half_logdet.end());
  for (int k = 0; k < K; ++k) {
    const double logw = std::log(g->weight[k]) - 0.5 * D * std::log(2.0 * M_PI) - (half_logdet[k] - min_hld);
    g->log_w[k] = -logw;
  }
  return g;
}
void oracle_gmm_destroy(void* h) { delete static_cast<Gmm*>(h); }
void oracle_gmm_get(void* h, double* prec_cho, double* neg_log_w) {
  Gmm* g = static_cast<Gmm*>(h);
  if (prec_cho) std::memcpy(prec_cho, g->prec_cho.data(), g->prec_cho.size() * sizeof(double));
  if (neg_log_w) std::memcpy(neg_log_w, g->log_w.data(), g->log_w.size() * sizeof(double));
}
// residual(x, &k): r[D+1] = [scale * L_k^T (x - mu_k) ; sqrt(-log w'_k)], k = argmin (|head|^2 - log w'_k)
int oracle_gmm_residual(void* h, const double* x, double* r) {
  Gmm* g = static_cast<Gmm*>(h);
  const int D = g->D;
  int best = 0;
  double bestv = std::numeric_limits<double>::infinity();
  std::vector<double> tmp(D), d(D);
  for (int k = 0; k < g->K; ++k) {
    for (int i = 0; i < D; ++i) d[i] = x[i] - g->mean[(size_t)k * D + i];
    const double* L = &g->prec_cho[(size_t)k * D * D];
    double sq = 0;
    for (int c = 0; c < D; ++c) {
      double s = 0;
      for (int rr = c; rr < D; ++rr) s += L[(size_t)rr * D + c] * d[rr];
      tmp[c] = g->resid_scale * s;
      sq += tmp[c] * tmp[c];
    }
    const double val = sq + g->log_w[k];
    if (val < bestv) {
      bestv = val; best = k;
      for (int c = 0; c < D; ++c) r[c] = tmp[c];
      r[D] = std::sqrt(g->log_w[k]);
    }
  }
  return best;
}

// PosePriorAAAnalytic::Evaluate (include/Sim3BA.h:263-315).  x = 69 stacked joint angle-axis.
// r [nRes]; J dense row-major [nRes][D] assembled from the reference's 23 blocks of nRes x 3.
int oracle_pose_prior(void* gmm, double beta_pose, int D, const double* x, double* r, double* J) {
  Gmm* g = static_cast<Gmm*>(gmm);
  const bool use_gmm = g && g->K > 0;
  const int nRes = use_gmm ? D + 1 : D;
  int comp = 0;
  if (use_gmm) {
    comp = oracle_gmm_residual(g, x, r);                       // :280
    for (int i = 0; i < nRes; ++i) r[i] *= beta_pose;
  } else {
    for (int i = 0; i < D; ++i) r[i] = x[i] * beta_pose;       // :283
  }
  if (J) {
    std::fill(J, J + (size_t)nRes * D, 0.0);
    if (use_gmm) {
      // :298-299  block j (nRes x 3): top D rows = L.middleRows(3j,3)^T * betaPose; last row zero
      const double* L = &g->prec_cho[(size_t)comp * D * D];
      for (int row = 0; row < D; ++row)
        for (int col = 0; col < D; ++col) J[(size_t)row * D + col] = L[(size_t)col * D + row] * beta_pose;
    } else {
      for (int i = 0; i < D; ++i) J[(size_t)i * D + i] = beta_pose;  // :304-310
    }
  }
  return comp;
}

// The prior blocks of F frames, threads over blocks as Ceres evaluates them: PosePriorAAAnalytic per frame
// (include/Sim3BA.h:263-315; r_pose [F][nRes], J_pose [F][nRes][D] or NULL, comp [F] or NULL) and
// ShapePriorL2Analytic per frame (:331-343; r_shape [F][nS], beta [F][nS]; skipped when beta_shape <= 0).
// The CPU baseline of bench.py times this beside oracle_evaluate_batch so that both sides evaluate the same blocks.
void oracle_priors_batch(void* gmm, double beta_pose, int D, int F, const double* params /*[F][7 + D]*/,
                         double beta_shape, int nS, const double* beta /*[F][nS]*/, int want_jac, int nthreads,
                         double* r_pose, double* J_pose, int* comp, double* r_shape) {
  Gmm* g = static_cast<Gmm*>(gmm);
  const int nRes = (g && g->K > 0) ? D + 1 : D;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < F; ++f) {
    if (beta_pose > 0.0) {
      const int k = oracle_pose_prior(gmm, beta_pose, D, params + (size_t)f * (7 + D) + 7, r_pose + (size_t)f * nRes,
                                      (want_jac && J_pose) ? J_pose + (size_t)f * nRes * D : nullptr);
      if (comp) comp[f] = k;
    }
    if (beta_shape > 0.0 && r_shape)
      for (int i = 0; i < nS; ++i) r_shape[(size_t)f * nS + i] = beta_shape * beta[(size_t)f * nS + i];   // :336
  }
}

// ceres::HuberLoss(delta) on s = |r|^2: rho = [rho, rho', rho'']
void oracle_huber(double delta, double s, double* rho) {
  const double b = delta * delta;
  if (s > b) {
    const double rt = std::sqrt(s);
    rho[0] = 2.0 * delta * rt - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), delta / rt);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// include/Utils.h:102-115 — note: uses update() joints, i.e. no Sim3 scale (quirk Q5)
double oracle_mean_pixel_error(int K, const int* jid, const double* uv, const double* joints /*[nJ][3]*/,
                               const double* intr) {
  if (K == 0) return 0.0;
  double sum = 0;
  for (int k = 0; k < K; ++k) {
    const double* Jp = joints + 3 * jid[k];
    const double u = intr[0] * Jp[0] / Jp[2] + intr[2];
    const double v = intr[1] * Jp[1] / Jp[2] + intr[3];
    sum += std::hypot(u - uv[2 * k], v - uv[2 * k + 1]);
  }
  return sum / K;
}

void oracle_rodrigues(const double* aa, double* R, double* dR /*[3][9]*/) {
  double d[3][9];
  rodrigues_with_grad(aa, R, d);
  if (dR) std::memcpy(dR, d, sizeof(d));
}

int oracle_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
