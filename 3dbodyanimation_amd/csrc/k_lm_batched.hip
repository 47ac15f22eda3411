// k_lm_batched.hip — device-resident Levenberg-Marquardt for batches of INDEPENDENT frames
// (3dba_single fits every frame on its own: src/main_single_frame.cpp:192-246; BASELINE configs[1], [2]).
//
// The reference hands each frame to ceres::Solve (LM + DENSE_QR on a 86-column problem,
// include/Sim3BA.h:472-479,641-647).  Here the whole LM state of every frame lives in HBM and one
// workgroup per frame does what DENSE_QR did, per iteration, without a host round trip:
//   k_lm_step    robustified Gram matrix  H = Jhat^T Jhat, g = Jhat^T rhat  on the f64 matrix cores
//                (v_mfma_f64_16x16x4_f64, same register as A and B operand, Jhat staged in LDS), prior blocks
//                added, Jacobi scaling, LM damping, Cholesky with the right-hand side carried as an extra row
//                (forward substitution for free), backward substitution, step, model cost change, candidate
//                point projected on the scale bounds
//   k_lm_accept  cost at the candidate, step quality rho, accept/reject, trust-region radius update,
//                Ceres' termination tests
// Algorithm = host_solver.cpp's (Ceres 1.14 defaults); the host only launches
// {sweep, step, residual sweep, accept} per iteration and polls the active-frame counter now and then.
#include "bodyfit_device.h"
#include "dense_inl.h"

namespace bodyfit {
#ifdef BODYFIT_STAMPS
__device__ unsigned long long* g_lm_dbg = nullptr;   // diagnostic builds only: per-frame s_memtime stamps of k_lm_step
#endif
namespace {

typedef __attribute__((ext_vector_type(4))) double d4;
constexpr int kN = 86;             // max unknowns per frame (76 + 10)
constexpr int kLd = 88;            // LDS leading dimension of the (n+1) x (n+1) system
constexpr int kRowsMax = 64;       // reprojection rows per frame handled on the device (32 keypoints)
constexpr int kJLd = 96;           // Jhat leading dimension: 6 column tiles of 16 (86 columns + rhat + pad)
constexpr int kMRows = 112;        // panel layout of the damped system: 96 padded unknowns + one tile row for the rhs
constexpr int kMLd = 98;
static_assert(kRowsMax * kJLd + 69 * 69 <= kMRows * kMLd, "Jhat and the GMM precision matrix share the damped system's region");

#ifdef BODYFIT_STAMPS
#define LSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (g_lm_dbg && threadIdx.x == 0) {                                                             \
      unsigned long long t_;                                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      g_lm_dbg[(size_t)blockIdx.x * 16 + (i)] = t_;                                                 \
    }                                                                                               \
  } while (0)
#define LTIME(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define LACC(i, dt) do { if (g_lm_dbg && threadIdx.x == 0) g_lm_dbg[(size_t)blockIdx.x * 16 + (i)] += (dt); } while (0)
#else
#define LSTAMP(i)
#define LTIME(var)
#define LACC(i, dt)
#endif

__device__ inline double huber_rho(double delta, double s, double* rho1) {
  const double b = delta * delta;
  if (delta > 0.0 && s > b) {
    const double rt = sqrt(s);
    *rho1 = delta / rt;
    return 2.0 * delta * rt - b;
  }
  *rho1 = 1.0;
  return s;
}

__device__ inline double block_sum(double v, double* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// cost of one frame from a residual vector: 1/2 sum rho(|r_kp|^2) + 1/2 |prior rows|^2 + 1/2 |shape rows|^2
// (256 threads: k_lm_init / k_lm_accept; the 512-thread form is frame_cost8 below)
__device__ double frame_cost(const LmProblem& P, int f, const double* __restrict__ r, double* red, int tid) {
  double acc = 0.0;
  for (int k = P.kp_offset[f] + tid; k < P.kp_offset[f + 1]; k += 256) {
    const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
    double r1d;
    acc += 0.5 * huber_rho(P.huber, r0 * r0 + r1 * r1, &r1d);
  }
  for (int i = tid; i < P.prior_rows; i += 256) {
    const double v = r[P.row_prior + (size_t)f * P.prior_rows + i];
    acc += 0.5 * v * v;
  }
  if (P.shape_rows_per_frame > 0)
    for (int i = tid; i < P.shape_rows_per_frame; i += 256) {
      const double v = r[P.row_shape + (size_t)f * P.shape_rows_per_frame + i];
      acc += 0.5 * v * v;
    }
  return block_sum(acc, red, tid);
}

__global__ __launch_bounds__(256) void k_lm_init(LmProblem P, LmState S, const double* __restrict__ r) {
  __shared__ double red[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const double c = frame_cost(P, f, r, red, tid);
  if (tid == 0) {
    S.cost[f] = c;
    S.initial_cost[f] = c;
    S.radius[f] = 1e4;
    S.dec[f] = 2.0;
    int fl = kLmActive;
    if (!(c == c) || c > 1e300) fl = (2 << kLmTermShift);   // non-finite initial cost: failure
    S.flags[f] = fl;
    S.iters[f] = 0; S.n_ok[f] = 0; S.n_bad[f] = 0;
    if (fl & kLmActive) atomicAdd(S.active_count, 1);
  }
}

// f64 value of lane `src` (wave-uniform lane id): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int kStepThreads = 512, kStepWaves = 8;
__device__ inline double block_sum8(double v, double* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// k_lm_step runs 512 threads = two waves per SIMD: its serial parts (diagonal blocks, back substitution) belong to wave 0,
// every other part is spread over all eight waves so that LDS and L2 latency of one wave is covered by its neighbour.
// Judge the candidate of frame f (cost at the candidate `new_cost` known to every thread): step quality, accept / reject,
// trust-region radius, Ceres' termination tests.  Accepted: x <- x_new, and the frame's prior rows (all rows when
// `all_rows`: the speculative iteration keeps the candidate's reprojection rows too) and GMM component become the current
// ones.  Returns the frame's new flags in *flags_io; true when the step was accepted.
struct TrustState { double cost, model, radius, dec; };
// (loaded by every thread BEFORE the barriers of the cost reduction: thread 0 rewrites these words after it)
__device__ __forceinline__ TrustState load_trust(const LmState& S, int f) {
  return TrustState{S.cost[f], S.model[f], S.radius[f], S.dec[f]};
}
__device__ __forceinline__ bool judge_candidate(const LmProblem& P, const LmState& S, int f, int tid, const TrustState& T,
                                                double new_cost,
                                                const double* __restrict__ r_new, double* __restrict__ r_cur,
                                                const int* __restrict__ comp_new, int* __restrict__ comp_cur, bool all_rows,
                                                int* flags_io) {
  const int flags = *flags_io;
  const int npose = kFrameParams, nb = P.ncols - npose;
  const double cost = T.cost, model = T.model;
  const double change = cost - new_cost;
  const double rho = change / model;
  const bool accept = (new_cost == new_cost) && new_cost < 1e300 && model > 0.0 && rho > 1e-3;
  int fl = flags & ~kLmHasCand;
  if (accept) {
    if (tid < npose) S.x[(size_t)f * npose + tid] = S.x_new[(size_t)f * npose + tid];
    else if (tid - npose < nb) S.beta[(size_t)f * nb + tid - npose] = S.beta_new[(size_t)f * nb + tid - npose];
    // the prior rows (and the GMM component) at the accepted point were computed by the residual sweep: they become
    // the current ones here, so the Jacobian sweep that follows launches no prior workgroups
    if (r_cur) {
      if (tid < P.prior_rows) {
        const size_t o = P.row_prior + (size_t)f * P.prior_rows + tid;
        r_cur[o] = r_new[o];
      } else if (tid >= 128 && tid - 128 < P.shape_rows_per_frame) {
        const size_t o = P.row_shape + (size_t)f * P.shape_rows_per_frame + tid - 128;
        r_cur[o] = r_new[o];
      }
      if (tid == 255 && comp_cur && comp_new) comp_cur[f] = comp_new[f];
      if (all_rows && tid >= 256) {
        const int k0 = P.kp_offset[f], nr = 2 * (P.kp_offset[f + 1] - k0);
        for (int i = tid - 256; i < nr; i += 256) r_cur[2 * (size_t)k0 + i] = r_new[2 * (size_t)k0 + i];
      }
    }
    if (fabs(change) < 1e-6 * cost) fl &= ~(kLmActive | kLmTermMask);   // function tolerance
  } else {
    if (T.radius / T.dec < 1e-32) fl = (fl & ~(kLmActive | kLmTermMask)) | (2 << kLmTermShift);
  }
  if (tid == 0) {
    // (straight-line: a branch that picks between S.n_ok and S.n_bad becomes a pointer table in scratch memory)
    const int it0 = S.iters[f], ok0 = S.n_ok[f], bad0 = S.n_bad[f];
    const double t = 2.0 * rho - 1.0;
    const double rad_ok = fmin(1e16, T.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
    S.iters[f] = it0 + 1;
    S.n_ok[f] = ok0 + (accept ? 1 : 0);
    S.n_bad[f] = bad0 + (accept ? 0 : 1);
    S.cost[f] = accept ? new_cost : T.cost;
    S.radius[f] = accept ? rad_ok : T.radius / T.dec;
    S.dec[f] = accept ? 2.0 : T.dec * 2.0;
    if ((flags & kLmActive) && !(fl & kLmActive)) atomicSub(S.active_count, 1);
    S.flags[f] = fl;
  }
  *flags_io = fl;
  return accept;
}

// 512-thread form of frame_cost: the first four waves carry exactly frame_cost's terms and the sum is formed in the same
// order, so the cost (and with it every decision of the trust region) is bit-identical in both forms of the iteration
__device__ double frame_cost8(const LmProblem& P, int f, const double* __restrict__ r, double* red, int tid) {
  double acc = 0.0;
  if (tid < 256) {
    for (int k = P.kp_offset[f] + tid; k < P.kp_offset[f + 1]; k += 256) {
      const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
      double r1d;
      acc += 0.5 * huber_rho(P.huber, r0 * r0 + r1 * r1, &r1d);
    }
    for (int i = tid; i < P.prior_rows; i += 256) {
      const double v = r[P.row_prior + (size_t)f * P.prior_rows + i];
      acc += 0.5 * v * v;
    }
    if (P.shape_rows_per_frame > 0)
      for (int i = tid; i < P.shape_rows_per_frame; i += 256) {
        const double v = r[P.row_shape + (size_t)f * P.shape_rows_per_frame + i];
        acc += 0.5 * v * v;
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Speculative iteration (cand.r != nullptr): the sweep before this launch evaluated residuals AND Jacobian at the
// candidate of the previous step into cand.{r, J, comp}.  The kernel first judges that candidate (what k_lm_accept does as a
// launch of its own in the plain iteration); an accepted frame builds its next system straight from the candidate's
// buffers and copies them into the current ones on the way (fire-and-forget stores under the Gram product), a rejected
// frame rebuilds from the current ones.  One launch and one sweep less per iteration.
struct LmCandidate { const double* r; const double* J; const int* comp; };
__global__ __launch_bounds__(kStepThreads) void k_lm_step(LmProblem P, LmState S, double* r_cur, double* J_cur, int* comp_cur,
                                                  LmCandidate cand,
                                                  const unsigned char* __restrict__ constant, int first_iter) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* M = sm;                          // kMRows x kMLd : damped scaled system (+ rhs row), factored in place
  double* Jh = sm;                         // kRowsMax x kJLd : robustified [J | r], lives in M's region until H is built
  double* H0 = sm + kMRows * kMLd;         // (n+1) x kLd : undamped unscaled H (row n = gradient)
  double* vec = H0 + (kN + 1) * kLd;       // g[88], scale[88], ds[96+], d[88], red[8]
  double* g = vec;
  double* sc = vec + 88;
  double* ds = vec + 176;                  // 112 entries
  double* dd = vec + 288;
  double* red = vec + 376;
  double* invd = vec + 392;                // 112 entries: 1 / L_jj of the factor (red holds 8 wave partials + a status word)
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = P.ncols, npose = kFrameParams, nb = n - npose;
  const int npad = (n + 15) & ~15, NB = npad >> 4;     // 80 / 96 unknowns padded to whole 16-column panels
  int flags = S.flags[f];
  bool use_cand = false;
  if (cand.r && (flags & kLmHasCand)) {
    const TrustState T = load_trust(S, f);
    const double new_cost = frame_cost8(P, f, cand.r, red, tid);
    use_cand = judge_candidate(P, S, f, tid, T, new_cost, cand.r, r_cur, cand.comp, comp_cur, true, &flags);
    __syncthreads();   // x, beta, radius of the new point are read by other threads below (same CU: L1 is shared)
  }
  // (bit arithmetic instead of ?: — hipcc turns a select between two kernel-argument pointers into a two-entry table
  //  in scratch memory with an indexed load on the kernel's critical path)
  auto pick = [&](const void* cur, const void* cnd) {
    const unsigned long long a = (unsigned long long)cur, b = (unsigned long long)cnd;
    return a ^ ((a ^ b) & (0ull - (unsigned long long)use_cand));
  };
  const double* __restrict__ r = reinterpret_cast<const double*>(pick(r_cur, cand.r));
  const double* __restrict__ J = reinterpret_cast<const double*>(pick(J_cur, cand.J));
  const int* __restrict__ comp = reinterpret_cast<const int*>(pick(comp_cur, cand.comp));
  // frames that leave without a candidate still hand the residual sweep (its prior workgroups read every frame) a
  // well-defined point: x_new = x
  auto no_candidate = [&]() {
    if (tid < npose) S.x_new[(size_t)f * npose + tid] = S.x[(size_t)f * npose + tid];
    else if (tid < n) S.beta_new[(size_t)f * nb + tid - npose] = S.beta[(size_t)f * nb + tid - npose];
  };
  if (!(flags & kLmActive)) {
    if (tid == 0) S.flags[f] = flags & ~kLmHasCand;
    no_candidate();
    return;
  }
  const int k0 = P.kp_offset[f], nrows = 2 * (P.kp_offset[f + 1] - k0);
  LSTAMP(0);

  // ---- every global operand of the step is requested here, in one go (one L2 round trip instead of five) ---------------
  //   J (11 values per thread), the residual rows, the selected GMM component's precision matrix (10 per thread),
  //   x - mu, the prior / shape residual rows, the constant mask, the Jacobi scaling of the first iterate.
  //   Unconditional loads from clamped addresses, masked afterwards (a predicated load is a branch with its own wait).
  const int nrows4 = (nrows + 3) & ~3;             // Jhat is zero-padded to whole k-steps of the matrix cores
  const int D = npose - 7;
  const double bp = P.beta_pose, bs = P.beta_shape;
  const bool has_prior = P.prior_rows > 0, has_gmm = has_prior && P.prec != nullptr;
  const bool has_shape = P.shape_rows_per_frame > 0;
  double* Pl = sm + kRowsMax * kJLd;       // [D][D] precision matrix of the frame's component, beside Jhat in M's region
  double* gp = invd;                       // [88] prior part of the gradient (invd is free until the factorisation)
  double* dx = dd;                         // [69] x - mu (dd is free until the step)
  double* cf = vec + 504;                  // [88] 1.0 = parameter held constant
  double jv[11];
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
    const bool on = row < nrows && c < n;
    jv[u] = J[on ? (size_t)(2 * k0 + row) * n + c : (size_t)0];
    if (!on) jv[u] = 0.0;
  }
  double r_own, r_other;     // this row's residual and the other coordinate of its keypoint
  {
    const int row = min(tid, max(nrows, 1) - 1);
    r_own = r[2 * (size_t)k0 + row]; r_other = r[2 * (size_t)k0 + (row ^ 1)];
  }
  double pv[10];
  double dxv = 0.0, gpv = 0.0, scv = 1.0, cfv = 0.0;
  if (has_gmm) {
    const int kc = comp[f];
    const double* Pm = P.prec + (size_t)kc * D * D;       // L L^T of the selected component
#pragma unroll
    for (int u = 0; u < 10; ++u) pv[u] = Pm[min(tid + u * kStepThreads, D * D - 1)];   // 69 * 69 = 4761 <= 10 * 512
    const int t = min(tid, D - 1);
    dxv = S.x[(size_t)f * npose + 7 + t] - P.gmm_mean[(size_t)kc * D + t];
  } else if (has_prior) {               // L2 pose prior: J^T r = beta_p r_prior on the 69 joint columns
    gpv = bp * r[P.row_prior + (size_t)f * P.prior_rows + min(max(tid - 7, 0), P.prior_rows - 1)];
    if (tid < 7 || tid >= npose) gpv = 0.0;
  }
  if (has_shape && tid >= npose && tid < n)                  // (ten threads: a branch of its own is cheap here)
    gpv = bs * r[P.row_shape + (size_t)f * P.shape_rows_per_frame + tid - npose];
  if (constant) cfv = constant[min(tid, npose - 1)] ? 1.0 : 0.0;
  if (!first_iter) scv = S.scale[(size_t)f * kN + min(tid, n - 1)];
  if (use_cand) {     // the accepted candidate's Jacobian becomes the current one (fire-and-forget)
#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
      if (row < nrows && c < n) J_cur[(size_t)(2 * k0 + row) * n + c] = jv[u];
    }
  }
  // ---- Jhat = sqrt(rho') [J | r], zero padded to 6 column tiles of 16 and a multiple of 4 rows --------------------------------
  if (tid < kRowsMax) {   // per-row robust weight sqrt(rho') and weighted residual
    double sw = 0.0, rr = 0.0;
    if (tid < nrows) {
      double rho1;
      huber_rho(P.huber, (tid & 1) ? r_other * r_other + r_own * r_own : r_own * r_own + r_other * r_other, &rho1);
      sw = sqrt(rho1);
      rr = sw * r_own;
    }
    ds[tid] = sw;                 // ds is free until the solve
    Jh[tid * kJLd + n] = rr;      // column n = rhat
  }
  if (tid < 88) {
    cf[tid] = (tid < npose) ? cfv : 0.0;
    gp[tid] = gpv;
    if (tid < D) dx[tid] = dxv;
  }
  if (has_gmm) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int e = tid + u * kStepThreads;
      if (e < D * D) Pl[e] = pv[u];
    }
  }
  __syncthreads();
  LSTAMP(11);
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
    if (row < nrows4 && c != n) Jh[row * kJLd + c] = ds[row] * jv[u];   // (jv is zero outside J; ds is zero past nrows)
  }
  for (int i = tid; i < nrows4 * (kJLd - 88); i += kStepThreads) Jh[(i / (kJLd - 88)) * kJLd + 88 + i % (kJLd - 88)] = 0.0;
  __syncthreads();
  LSTAMP(1);
  // ---- Gram matrix on the f64 matrix cores: 21 lower tile pairs dealt to the 8 waves; every tile leaves with its prior
  //      terms added and is stored in both triangles (the model cost change needs H d with the full matrix) ---------
  {
    const int m = lane & 15, kk = lane >> 4;
    const double bp2 = bp * bp, bs2 = bs * bs;
    const int nsteps = nrows4 / 4;
    // The wave's (up to) three tiles t = wave, wave + 8, wave + 16 run their k loops INTERLEAVED: three independent
    // accumulator chains keep the matrix pipe fed, and the three epilogues (prior terms from LDS, stores in both triangles)
    // are issued together.  The k loop is software-pipelined by hand, two k-steps deep per tile: operands of step s + 2 are
    // requested right after the products of step s (left to itself the compiler reads, waits, multiplies, and every product
    // pays a full LDS latency); the uniform branches keep the order.
    int tis[3], tjs[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = min(wave + 8 * u, 20);
      int ti = 0, rem = t;
      while (rem > ti) { rem -= ti + 1; ++ti; }       // t -> (ti, tj) of the lower triangle, row-major
      tis[u] = ti; tjs[u] = rem;
    }
    const bool third = wave + 16 < 21;                 // waves 0-4 carry three tiles, waves 5-7 two
    d4 acc[3];
    const double* pa[3];
    const double* pb[3];
    double av[3][2], bv[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      acc[u] = d4{0.0, 0.0, 0.0, 0.0};
      pa[u] = Jh + kk * kJLd + 16 * tis[u] + m;
      pb[u] = Jh + kk * kJLd + 16 * tjs[u] + m;
#pragma unroll
      for (int w = 0; w < 2; ++w) { av[u][w] = pa[u][4 * w * kJLd]; bv[u][w] = pb[u][4 * w * kJLd]; }
    }
    for (int s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const bool on = s0 + w < nsteps;
        const int sn = min(s0 + 2 + w, kRowsMax / 4 - 1);     // (past the last step: a row nothing multiplies)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (on && (u < 2 || third)) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][w], bv[u][w], acc[u], 0, 0, 0);
          av[u][w] = pa[u][4 * sn * kJLd]; bv[u][w] = pb[u][4 * sn * kJLd];
        }
      }
    }
    // D: column = lane & 15 (B side, tile tj), row = (lane >> 4) + 4 q (A side, tile ti).  Branch-free up to the
    // stores: the prior terms are read from clamped addresses and masked (row n, the rhat column, is the gradient
    // and takes none; its mirror image lands in column n of H0, which nothing reads)
    double pl[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int j = 16 * tjs[u] + m;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * tis[u] + kk + 4 * q;
        pl[u][q] = has_gmm ? Pl[min(max(i - 7, 0), 68) * 69 + min(max(j - 7, 0), 68)] : ((i == j) ? 1.0 : 0.0);
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (u == 2 && !third) continue;
      const int j = 16 * tjs[u] + m;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * tis[u] + kk + 4 * q;
        double v = acc[u][q];
        if (has_prior && j >= 7 && i < npose) v += bp2 * pl[u][q];
        if (has_shape && i == j && i >= npose && i < n) v += bs2;
        if (i <= n && j < n && j <= i) {
          H0[i * kLd + j] = v;
          H0[j * kLd + i] = v;
        }
      }
    }
    // prior part of the gradient for the GMM: J^T r = beta_p^2 s Prec (x - mu); Prec is symmetric, so thread j reads
    // column j (consecutive lanes, consecutive words).  Waves 6 and 7 carry one Gram tile less than the others.
    if (has_gmm && tid >= 384 + 7 && tid < 384 + npose) {
      const int j = tid - 384 - 7;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      static_assert(kFrameParams - 7 == 69, "three batches of 23");
#pragma unroll
      for (int kb = 0; kb < 69; kb += 23) {       // 23 elements per batch, their 46 LDS reads in flight together
        double pw[23], dw[23];
#pragma unroll
        for (int u = 0; u < 23; ++u) { pw[u] = Pl[(kb + u) * 69 + j]; dw[u] = dx[kb + u]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u + 2 < 23; u += 3) { a0 += pw[u] * dw[u]; a1 += pw[u + 1] * dw[u + 1]; a2 += pw[u + 2] * dw[u + 2]; }
        a0 += pw[21] * dw[21]; a1 += pw[22] * dw[22];
        __builtin_amdgcn_sched_barrier(0);
      }
      gp[7 + j] = bp2 * P.gmm_scale * ((a0 + a1) + a2);
    }
  }
  LSTAMP(13);
  __syncthreads();
  LSTAMP(2);
  LSTAMP(3);
  // ---- gradient, Jacobi scaling (fixed at the first iterate) ----------------------------------------------------
  double gm = 0.0;
  if (tid < n) {
    const double gi0 = H0[n * kLd + tid] + gp[tid];
    g[tid] = gi0;
    const double s0 = first_iter ? 1.0 / (1.0 + sqrt(H0[tid * kLd + tid])) : scv;
    if (first_iter) S.scale[(size_t)f * kN + tid] = s0;
    sc[tid] = s0;
    // gradient tolerance (projected on the scale bounds), Ceres gradient_tolerance = 1e-10
    if (cf[tid] == 0.0) {
      double gi = gi0;
      if (tid == 0) {
        const double x0 = S.x[(size_t)f * npose];
        gi = x0 - fmin(fmax(x0 - gi, P.scale_lo), P.scale_hi);
      }
      gm = fabs(gi);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
  if (lane == 0) red[wave] = gm;
  __syncthreads();
  gm = fmax(red[0], red[1]);        // (n <= 86: waves 0 and 1 hold every entry)
  LSTAMP(12);
  if (gm <= 1e-10) {
    if (tid == 0) {
      S.flags[f] = (flags & ~(kLmActive | kLmHasCand | kLmTermMask));   // termination 0: convergence
      atomicSub(S.active_count, 1);
    }
    no_candidate();
    return;
  }
  // ---- scaled, damped system in panel layout: unknowns padded with identity to npad, rhs = row npad -------------
  const double radius = S.radius[f];
  const double inv_radius = 1.0 / radius;
  {
    // 112 x 96 entries, 21 per thread: columns j = (tid & 31) + 32 jj, rows i = (tid >> 5) + 16 ii.  Straight-line code:
    // the per-column and per-row factors first, then the 21 matrix reads in flight together, then the 21 stores.
    double scj[3], cfj[3], gj[3], sci[7], cfi[7];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = min((tid & 31) + 32 * jj, n - 1);
      scj[jj] = sc[j]; cfj[jj] = cf[j]; gj[jj] = g[j];
    }
#pragma unroll
    for (int ii = 0; ii < 7; ++ii) {
      const int i = min((tid >> 5) + 16 * ii, n - 1);
      sci[ii] = sc[i]; cfi[ii] = cf[i];
    }
    double hv[7][3];
#pragma unroll
    for (int ii = 0; ii < 7; ++ii)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int i = min((tid >> 5) + 16 * ii, n - 1), j = min((tid & 31) + 32 * jj, n - 1);
        hv[ii][jj] = H0[i * kLd + j];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ii = 0; ii < 7; ++ii) {
      const int i = (tid >> 5) + 16 * ii;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int j = (tid & 31) + 32 * jj;
        double v = 0.0;
        if (i < n && j <= i) {
          if (cfi[ii] != 0.0 || cfj[jj] != 0.0) v = (i == j) ? 1.0 : 0.0;
          else {
            v = hv[ii][jj] * sci[ii] * scj[jj];
            if (i == j) v += fmin(fmax(v, 1e-6), 1e32) * inv_radius;
          }
        } else if (i < npad) {
          v = (i == j) ? 1.0 : 0.0;                                  // identity padding keeps the system SPD
        } else if (i == npad && j < n) {
          v = (cfj[jj] != 0.0) ? 0.0 : -gj[jj] * scj[jj];            // rhs = -S g
        }
        if (i < npad + 16 && j < npad) M[i * kMLd + j] = v;
      }
    }
  }
  if (tid == 0) red[8] = 1.0;   // factorisation status
  __syncthreads();

  LSTAMP(4);
  if (tid >= n && tid < npad) invd[tid] = 1.0;       // padded unknowns: unit pivots
  // ---- blocked right-looking Cholesky, 16-column panels; the rhs row rides along as one more row below ----------
  // (a) diagonal block p in registers of wave 0: lane r holds row r; column values travel by v_readlane (dense_inl.h:
  //     the next pivot's reciprocal square root runs under the current pivot's updates).
  //     lanes 0-15: rows of the block; lanes 16-31: rows of the identity, which come out as L_pp^-T (kept in the unused
  //     strict upper triangle of the block, its diagonal 1 / L_cc in invd): the panel solve and the back substitution
  //     become products on the matrix cores
  auto diag_block = [&](int p) {
    // the block and the identity below it spread over all 64 lanes of wave 0 in the layout of an f64 16 x 16 accumulator
    // (dense_inl.h diag_factor16_acc): lane (m, kk), register q <-> row kk + 4 q, column m
    const int c0 = 16 * p;
    const int m = lane & 15, kk = lane >> 4;
    double a[4], b[4], invc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // (the FULL symmetric block, mirrored from its lower triangle: the strict upper triangle of a
      const int r = kk + 4 * q, lo = max(r, m), hi = min(r, m);   //  factored tile holds L^-T, and only the lower one is kept up to date)
      a[q] = M[(c0 + lo) * kMLd + c0 + hi];
      b[q] = (r == m) ? 1.0 : 0.0;
    }
    const bool okp = diag_factor16_acc(a, b, lane, invc, min(16, n - c0));
    // lower triangle: rows of L; strict upper triangle: rows of L_pp^-T; its diagonal 1 / L_cc in invd
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = kk + 4 * q;
      M[(c0 + r) * kMLd + c0 + m] = (r >= m) ? a[q] : b[q];
    }
    if (kk == 0) invd[c0 + m] = invc;
    if (lane == 0 && !okp) red[8] = 0.0;
  };
  if (wave == 0) diag_block(0);
  __syncthreads();
  for (int p = 0; p < NB; ++p) {
    const int c0 = 16 * p;
    unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0;
    LTIME(ta0);
    LTIME(ta1);
    if (red[8] == 0.0) break;
    // (b) panel solve  X = A_below L_pp^-T  on the matrix cores: 16-row tiles below the diagonal block incl. the rhs tile
    {
      const int m = lane & 15, kk = lane >> 4;
      for (int I = p + 1 + wave; I <= NB; I += kStepWaves) {
        double a4[4], b4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int k = 4 * s4 + kk;   // B[k][j = m] = (L^-T)[k][m]
          a4[s4] = M[(16 * I + m) * kMLd + c0 + k];
          b4[s4] = (k < m) ? M[(c0 + k) * kMLd + c0 + m] : (k == m ? invd[c0 + k] : 0.0);
        }
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], b4[s4], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) M[(16 * I + kk + 4 * q) * kMLd + c0 + m] = acc[q];
      }
    }
    __syncthreads();
    LTIME(ta2);
    // (c) trailing update  A[I][Kc] -= X_I X_Kc^T  on the f64 matrix cores (tiles at and below the diagonal).
    //     Look-ahead: wave 0 updates the next diagonal tile first and factors it at once (the long serial part of a
    //     panel) while the other seven waves update the rest of the trailing matrix.
    {
      const int m = lane & 15, kk = lane >> 4;
      auto tile_update = [&](int I, int Kc) {
        d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = M[(16 * I + kk + 4 * q) * kMLd + 16 * Kc + m];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double av = -M[(16 * I + m) * kMLd + c0 + 4 * s4 + kk];
          const double bv = M[(16 * Kc + m) * kMLd + c0 + 4 * s4 + kk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) M[(16 * I + kk + 4 * q) * kMLd + 16 * Kc + m] = acc[q];
      };
      if (wave == 0) {
        if (p + 1 < NB) {
          tile_update(p + 1, p + 1);
          diag_block(p + 1);
        }
      } else {
        int t = 0;
        for (int I = p + 1; I <= NB; ++I)
          for (int Kc = p + 1; Kc <= I && Kc < NB; ++Kc) {
            if (I == p + 1 && Kc == p + 1) continue;                 // wave 0's
            if (t++ % (kStepWaves - 1) != wave - 1) continue;
            tile_update(I, Kc);
          }
      }
    }
    __syncthreads();
    LTIME(ta3);
    LACC(10, ta3 - ta2);
    (void)ta0; (void)ta1; (void)ta2; (void)ta3;
  }
  if (red[8] == 0.0) {
    if (tid == 0) {
      const double dec = S.dec[f];
      const double rad = radius / dec;
      S.radius[f] = rad;
      S.dec[f] = dec * 2.0;
      S.n_bad[f] += 1;
      S.iters[f] += 1;
      int fl = flags & ~kLmHasCand;
      if (rad < 1e-32) { fl = (fl & ~(kLmActive | kLmTermMask)) | (2 << kLmTermShift); atomicSub(S.active_count, 1); }
      S.flags[f] = fl;
    }
    no_candidate();
    return;
  }
  LSTAMP(5);
  // ---- backward substitution ds = L^-T y, wave 0, on the vector pipe: ONE right-hand side makes a 16 x 16 x 4 f64 matrix
  //      instruction (64 cycles on gfx950, fifteen of its sixteen rows idle) four times dearer than the four FMAs per
  //      lane it replaces.  Row form  X L = Z  (Z = y, the rhs row):  X_p = (Z_p - sum_{q > p} X_q L[q][p]) L_pp^-1, panel by
  //      panel from the last, right-looking: as soon as X_p is known every Z_q, q < p, takes its update.
  //      lane = (kk = lane >> 4, m = lane & 15): partial sums over k = 4 s + kk, reduced across kk by two shuffles. ----------
  if (wave == 0) {
    const int m = lane & 15, kk = lane >> 4;
    const double* Zr = M + npad * kMLd;          // row 0 of the rhs tile = y
    double z[6];                                  // Z_p[m], replicated over kk
#pragma unroll
    for (int q = 0; q < 6; ++q) z[q] = Zr[min(16 * q, npad - 16) + m];
    for (int p = NB - 1; p >= 0; --p) {
      // operands of this panel: L_pp^-1 (strict upper triangle of the diagonal tile holds L_pp^-T, invd its diagonal) and
      // the tiles L[p][q], q < p — none depends on the chain, all are requested before the first product
      double li[4], lt[5][4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 4 * s4 + kk;
        const double up = M[(16 * p + min(m, k)) * kMLd + 16 * p + max(m, k)];
        const double dg = invd[16 * p + k];
        li[s4] = (m < k) ? up : (m == k ? dg : 0.0);
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int qc = min(q, max(p - 1, 0));
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) lt[q][s4] = M[(16 * p + 4 * s4 + kk) * kMLd + 16 * qc + m];
      }
      double zp = z[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) zp = (q == p) ? z[q] : zp;
      // X_p[m] = sum_k Z_p[k] (L_pp^-1)[k][m]
      double xp = 0.0;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) xp += __shfl(zp, 4 * s4 + kk, 64) * li[s4];   // Z_p[4 s + kk] lives in lane 4 s + kk
      xp += __shfl_xor(xp, 16, 64);
      xp += __shfl_xor(xp, 32, 64);
      if (kk == 0) ds[16 * p + m] = xp;
      // Z_q[m] -= sum_k X_p[k] L[p][q][k][m]
      double xk[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) xk[s4] = __shfl(xp, 4 * s4 + kk, 64);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        if (q < p) {      // (uniform)
          double a = 0.0;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) a += xk[s4] * lt[q][s4];
          a += __shfl_xor(a, 16, 64);
          a += __shfl_xor(a, 32, 64);
          z[q] -= a;
        }
      }
    }
  }
  __syncthreads();
  LSTAMP(6);
  // ---- step, projection on the scale bounds, model change -dg - 1/2 d H d with the undamped H ----------------------
  const double* xf = S.x + (size_t)f * npose;
  if (tid < n) {
    double di = ds[tid] * sc[tid];
    if (tid == 0) {
      const double s_new = fmin(fmax(xf[0] + di, P.scale_lo), P.scale_hi);
      di = s_new - xf[0];
    }
    dd[tid] = di;
  }
  __syncthreads();
  double part = 0.0, dn = 0.0, xn = 0.0;
  if (tid < n) {
    // (H0 is stored in both triangles: thread i walks COLUMN i, consecutive lanes read consecutive words)
    double h0 = 0.0, h1 = 0.0;
    double h2 = 0.0, h3 = 0.0;
    const int nfull = n & ~15;
#pragma unroll 1
    for (int jb = 0; jb < nfull; jb += 16) {     // 16 columns per batch, their 32 LDS reads in flight together
      const double* hp = H0 + jb * kLd + tid;
      const double* dp = dd + jb;
      double hw[16], dw[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { hw[u] = hp[u * kLd]; dw[u] = dp[u]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        h0 += hw[u] * dw[u]; h1 += hw[u + 1] * dw[u + 1]; h2 += hw[u + 2] * dw[u + 2]; h3 += hw[u + 3] * dw[u + 3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = nfull; j < n; ++j) h0 += H0[j * kLd + tid] * dd[j];
    h0 += h2; h1 += h3;
    const double hd = h0 + h1;
    part = -dd[tid] * g[tid] - 0.5 * dd[tid] * hd;
    dn = dd[tid] * dd[tid];
    const double xv = (tid < npose) ? xf[tid] : S.beta[(size_t)f * nb + tid - npose];
    xn = xv * xv;
  }
  // three sums in one pass (n <= 86: waves 0 and 1 hold every term)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    part += __shfl_xor(part, off, 64);
    dn += __shfl_xor(dn, off, 64);
    xn += __shfl_xor(xn, off, 64);
  }
  if (lane == 0 && wave < 2) { red[wave] = part; red[2 + wave] = dn; red[4 + wave] = xn; }
  __syncthreads();
  const double model = red[0] + red[1];
  const double dnorm = sqrt(red[2] + red[3]);
  const double xnorm = sqrt(red[4] + red[5]);
  if (dnorm <= 1e-8 * (xnorm + 1e-8)) {      // Ceres parameter_tolerance
    if (tid == 0) {
      S.flags[f] = (flags & ~(kLmActive | kLmHasCand | kLmTermMask));
      atomicSub(S.active_count, 1);
    }
    no_candidate();
    return;
  }
  if (tid < npose) S.x_new[(size_t)f * npose + tid] = xf[tid] + dd[tid];
  else if (tid < n) S.beta_new[(size_t)f * nb + tid - npose] = S.beta[(size_t)f * nb + tid - npose] + dd[tid];
  if (tid == 0) {
    S.model[f] = model;
    S.flags[f] = flags | kLmHasCand;
  }
  LSTAMP(7);
}

// Reprojection part of one frame's normal equations for the window solver (host_solver.cpp): the robustified Gram
// matrix of [J_f | r_f] — A_f (76 x 76), B_f (76 x nb), this frame's share of C (nb x nb), and the gradient in row n —
// on the f64 matrix cores, lower triangle of an (n + 1) x kLd panel per frame.  The host adds the prior / temporal
// blocks (constant Jacobians) and runs the block-tridiagonal factorisation; it no longer needs J itself
// ("normal equations built on device").
// sel != nullptr (window LM, launch_frame_normal_sel): *sel = 0: (r_io, J); 1: (r_alt, J_alt) — the accepted candidate, whose
// residual rows this launch also copies into r_io (its own frame's reprojection rows and a slice of the remaining rows per
// workgroup); 2: nothing to do.
__global__ __launch_bounds__(kStepThreads) void k_frame_normal(int F, int n, const int* __restrict__ kp_offset, double huber,
                                                       double* __restrict__ r_io, const double* __restrict__ J_in,
                                                       const double* __restrict__ r_alt, const double* __restrict__ J_alt,
                                                       const double* __restrict__ sel, int total_rows,
                                                       double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Jh = sm;                         // kRowsMax x kJLd : robustified [J | r]
  double* ds = sm + kRowsMax * kJLd;       // per-row sqrt(rho')
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int which = sel ? (int)sel[0] : 0;   // (uniform)
  if (which == 2) return;
  const double* __restrict__ r = (which == 1) ? r_alt : r_io;
  const double* __restrict__ J = (which == 1) ? J_alt : J_in;
  if (which == 1) {
    // the candidate's rows become the current ones: this frame's reprojection rows (read again below from r_alt, not from
    // here) and the f-th slice of the prior / shape / temporal rows
    const int kA = 2 * kp_offset[0], kB = 2 * kp_offset[F];          // reprojection rows [kA, kB)
    const int a0 = 2 * kp_offset[f], a1 = 2 * kp_offset[f + 1];
    for (int i = a0 + tid; i < a1; i += kStepThreads) r_io[i] = r_alt[i];
    const int rest = total_rows - (kB - kA), chunk = (rest + F - 1) / F;
    const int b0 = kB + f * chunk, b1 = min(total_rows, b0 + chunk);
    for (int i = b0 + tid; i < b1; i += kStepThreads) r_io[i] = r_alt[i];
  }
  const int k0 = kp_offset[f], nrows = 2 * (kp_offset[f + 1] - k0);
  const int nrows4 = (nrows + 3) & ~3;
  // J and the residual rows in one round trip (unconditional loads from clamped addresses, masked afterwards)
  double jv[11];
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
    const bool on = row < nrows && c < n;
    jv[u] = J[on ? (size_t)(2 * k0 + row) * n + c : (size_t)0];
    if (!on) jv[u] = 0.0;
  }
  double r_own, r_other;
  {
    const int row = min(tid, max(nrows, 1) - 1);
    r_own = r[2 * (size_t)k0 + row]; r_other = r[2 * (size_t)k0 + (row ^ 1)];
  }
  if (tid < kRowsMax) {
    double sw = 0.0, rr = 0.0;
    if (tid < nrows) {
      double rho1;
      huber_rho(huber, (tid & 1) ? r_other * r_other + r_own * r_own : r_own * r_own + r_other * r_other, &rho1);
      sw = sqrt(rho1);
      rr = sw * r_own;
    }
    ds[tid] = sw;
    Jh[tid * kJLd + n] = rr;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
    if (row < nrows4 && c != n) Jh[row * kJLd + c] = ds[row] * jv[u];
  }
  for (int i = tid; i < nrows4 * (kJLd - 88); i += kStepThreads) Jh[(i / (kJLd - 88)) * kJLd + 88 + i % (kJLd - 88)] = 0.0;
  __syncthreads();
  double* H = out + (size_t)f * (kN + 1) * kLd;
  const int m = lane & 15, kk = lane >> 4;
  const int nsteps = nrows4 / 4;
  int pair = 0;
  for (int ti = 0; ti < 6; ++ti)
    for (int tj = 0; tj <= ti; ++tj, ++pair) {
      if (pair % kStepWaves != wave) continue;
      // k loop software-pipelined four steps deep (see k_lm_step)
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* pa = Jh + kk * kJLd + 16 * ti + m;
      const double* pb = Jh + kk * kJLd + 16 * tj + m;
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = pa[4 * u * kJLd]; bv[u] = pb[4 * u * kJLd]; }
      for (int s0 = 0; s0 < nsteps; s0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (s0 + u < nsteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
          const int sn = min(s0 + 4 + u, kRowsMax / 4 - 1);
          av[u] = pa[4 * sn * kJLd]; bv[u] = pb[4 * sn * kJLd];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * ti + kk + 4 * q, j = 16 * tj + m;
        if (i <= n && j < n && j <= i) H[i * kLd + j] = acc[q];
      }
    }
}

__global__ __launch_bounds__(256) void k_lm_accept(LmProblem P, LmState S, const double* __restrict__ r_new,
                                                    double* __restrict__ r_cur, const int* __restrict__ comp_new,
                                                    int* __restrict__ comp_cur) {
  __shared__ double red[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  int flags = S.flags[f];
  if (!(flags & kLmHasCand)) return;
  const TrustState T = load_trust(S, f);
  const double new_cost = frame_cost(P, f, r_new, red, tid);
  judge_candidate(P, S, f, tid, T, new_cost, r_new, r_cur, comp_new, comp_cur, false, &flags);
}

}  // namespace

size_t lm_step_lds_bytes() { return (size_t)(kMRows * kMLd + (kN + 1) * kLd + 640) * sizeof(double); }

void launch_lm_init(const LmProblem& P, const LmState& S, const double* d_r, hipStream_t s) {
  BODYFIT_LAUNCH(k_lm_init, dim3(P.F), dim3(256), 0, s, P, S, d_r);
}
void launch_lm_step(const LmProblem& P, const LmState& S, double* d_r, double* d_J, int* d_comp, const double* d_r_cand,
                    const double* d_J_cand, const int* d_comp_cand, const unsigned char* d_constant, int first_iter,
                    hipStream_t s) {
  static DeviceOnce attr;
  const size_t lds = lm_step_lds_bytes();
  attr.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  BODYFIT_LAUNCH(k_lm_step, dim3(P.F), dim3(kStepThreads), lds, s, P, S, d_r, d_J, d_comp,
                     LmCandidate{d_r_cand, d_J_cand, d_comp_cand}, d_constant, first_iter);
}
void launch_lm_accept(const LmProblem& P, const LmState& S, const double* d_r_new, double* d_r_cur, const int* d_comp_new,
                      int* d_comp_cur, hipStream_t s) {
  BODYFIT_LAUNCH(k_lm_accept, dim3(P.F), dim3(256), 0, s, P, S, d_r_new, d_r_cur, d_comp_new, d_comp_cur);
}

void launch_frame_normal(int F, int n, const int* d_kp_offset, double huber, const double* d_r, const double* d_J,
                         double* d_out, hipStream_t s) {
  if (F <= 0) return;
  const size_t lds = (size_t)(kRowsMax * kJLd + kRowsMax) * sizeof(double);
  static DeviceOnce attr_set;
  attr_set.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_normal), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  BODYFIT_LAUNCH(k_frame_normal, dim3(F), dim3(kStepThreads), lds, s, F, n, d_kp_offset, huber, const_cast<double*>(d_r), d_J,
                 (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0, d_out);
}
void launch_frame_normal_sel(int F, int n, const int* d_kp_offset, double huber, double* d_r, const double* d_J,
                             const double* d_r_alt, const double* d_J_alt, const double* d_sel, int total_rows, double* d_out,
                             hipStream_t s) {
  if (F <= 0) return;
  const size_t lds = (size_t)(kRowsMax * kJLd + kRowsMax) * sizeof(double);
  static DeviceOnce attr_set;
  attr_set.run(current_device(), [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_normal), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  BODYFIT_LAUNCH(k_frame_normal, dim3(F), dim3(kStepThreads), lds, s, F, n, d_kp_offset, huber, d_r, d_J, d_r_alt, d_J_alt, d_sel,
                 total_rows, d_out);
}

}  // namespace bodyfit

#ifdef BODYFIT_STAMPS
extern "C" int bodyfit_debug_set_lm_stamp_buffer(unsigned long long* d_buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(bodyfit::g_lm_dbg), &d_buf, sizeof(d_buf));
}
#endif
