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
static_assert(kRowsMax * kJLd <= kMRows * kMLd, "Jhat must fit in the region it shares with the damped system");

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
    S.iters[f] += 1;
    if (accept) {
      S.cost[f] = new_cost;
      const double t = 2.0 * rho - 1.0;
      S.radius[f] = fmin(1e16, T.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      S.dec[f] = 2.0;
      S.n_ok[f] += 1;
    } else {
      S.radius[f] = T.radius / T.dec;
      S.dec[f] = T.dec * 2.0;
      S.n_bad[f] += 1;
    }
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
  const double* __restrict__ r = use_cand ? cand.r : r_cur;
  const double* __restrict__ J = use_cand ? cand.J : J_cur;
  const int* __restrict__ comp = use_cand ? cand.comp : comp_cur;
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

  // ---- Jhat = sqrt(rho') [J | r], zero padded to 6 column tiles of 16 and a multiple of 4 rows -------------
  const int nrows4 = (nrows + 3) & ~3;
  if (tid < kRowsMax) {   // per-row robust weight sqrt(rho') and weighted residual (one round trip)
    double sw = 0.0, rr = 0.0;
    if (tid < nrows) {
      const int k = k0 + (tid >> 1);
      const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
      double rho1;
      huber_rho(P.huber, r0 * r0 + r1 * r1, &rho1);
      sw = sqrt(rho1);
      rr = sw * ((tid & 1) ? r1 : r0);
    }
    ds[tid] = sw;                 // ds is free until the solve
    Jh[tid * kJLd + n] = rr;      // column n = rhat
  }
  __syncthreads();
  {
    // fixed trip count (kRowsMax * 88 / 512 = 11 predicated passes): all loads of J in flight together
    double jv[11];
#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
      jv[u] = (row < nrows && c < n) ? J[(size_t)(2 * k0 + row) * n + c] : 0.0;
    }
    if (use_cand) {
#pragma unroll
      for (int u = 0; u < 11; ++u) {
        const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
        if (row < nrows && c < n) J_cur[(size_t)(2 * k0 + row) * n + c] = jv[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
      if (row < nrows4 && c != n) Jh[row * kJLd + c] = (row < nrows && c < n) ? ds[row] * jv[u] : 0.0;
    }
    for (int i = tid; i < nrows4 * (kJLd - 88); i += kStepThreads) Jh[(i / (kJLd - 88)) * kJLd + 88 + i % (kJLd - 88)] = 0.0;
  }
  __syncthreads();
  LSTAMP(1);
  // ---- Gram matrix on the f64 matrix cores: 21 lower tile pairs dealt to the 4 waves -----------------------
  {
    const int m = lane & 15, kk = lane >> 4;
    int pair = 0;
    for (int ti = 0; ti < 6; ++ti)
      for (int tj = 0; tj <= ti; ++tj, ++pair) {
        if (pair % kStepWaves != wave) continue;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int nsteps = nrows4 / 4;
        for (int s0 = 0; s0 < nsteps; s0 += 8) {      // eight k-steps per batch: their 16 LDS reads are in flight together
          double av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool on = s0 + u < nsteps;
            const int row = on ? 4 * (s0 + u) + kk : 0;
            av[u] = on ? Jh[row * kJLd + 16 * ti + m] : 0.0;
            bv[u] = on ? Jh[row * kJLd + 16 * tj + m] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        // D: column = lane & 15 (B side, tile tj), row = (lane >> 4) + 4 q (A side, tile ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 16 * ti + kk + 4 * q, j = 16 * tj + m;
          if (i <= n && j < n && j <= i) H0[i * kLd + j] = acc[q];   // row n (the rhat column) is the gradient
        }
      }
  }
  __syncthreads();
  LSTAMP(2);
  // ---- priors: pose prior on the 69 joint columns, shape prior on beta ---------------------------------------
  const int D = npose - 7;
  if (P.prior_rows > 0) {
    const double* rp = r + P.row_prior + (size_t)f * P.prior_rows;
    const double bp = P.beta_pose;
    if (P.prec) {
      const int kc = comp[f];
      const double* Pm = P.prec + (size_t)kc * D * D;       // L L^T of the selected component
      {
        double pv[10];   // 69 * 69 = 4761 <= 10 * 512
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int e = tid + u * kStepThreads;
          pv[u] = (e < D * D) ? Pm[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int e = tid + u * kStepThreads, i = e / D, j = e % D;
          if (e < D * D && j <= i) H0[(7 + i) * kLd + 7 + j] += bp * bp * pv[u];
        }
      }
      if (tid < D) {
        // J^T r = beta_p L r[0:69] with r[0:69] = beta_p s L^T (x - mu)  ->  beta_p^2 s Prec (x - mu):
        // one row of the precision matrix per thread, 69 independent loads
        const double* xq = S.x + (size_t)f * npose + 7;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 69; ++k) a += (k < D) ? Pm[(size_t)tid * D + k] * (xq[k] - P.gmm_mean[(size_t)kc * D + k]) : 0.0;
        H0[n * kLd + 7 + tid] += bp * bp * P.gmm_scale * a;
      }
    } else if (tid < D) {
      H0[(7 + tid) * kLd + 7 + tid] += bp * bp;
      H0[n * kLd + 7 + tid] += bp * rp[tid];
    }
  }
  if (P.shape_rows_per_frame > 0 && tid >= 128 && tid - 128 < nb) {
    const int i = tid - 128;
    H0[(npose + i) * kLd + npose + i] += P.beta_shape * P.beta_shape;
    H0[n * kLd + npose + i] += P.beta_shape * r[P.row_shape + (size_t)f * P.shape_rows_per_frame + i];
  }
  __syncthreads();
  LSTAMP(3);
  // ---- gradient, Jacobi scaling (fixed at the first iterate) ----------------------------------------------------
  if (tid < n) {
    g[tid] = H0[n * kLd + tid];
    const double s0 = first_iter ? 1.0 / (1.0 + sqrt(H0[tid * kLd + tid])) : S.scale[(size_t)f * kN + tid];
    if (first_iter) S.scale[(size_t)f * kN + tid] = s0;
    sc[tid] = s0;
  }
  // gradient tolerance (projected on the scale bounds), Ceres gradient_tolerance = 1e-10
  double gm = 0.0;
  if (tid < n && !(tid < npose && constant && constant[tid])) {
    double gi = H0[n * kLd + tid];
    if (tid == 0) {
      const double s0 = S.x[(size_t)f * npose];
      gi = s0 - fmin(fmax(s0 - gi, P.scale_lo), P.scale_hi);
    }
    gm = fabs(gi);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
  if (lane == 0) red[wave] = gm;
  __syncthreads();
  gm = fmax(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), fmax(fmax(red[4], red[5]), fmax(red[6], red[7])));
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
  for (int ii = 0; ii < (kMRows + 15) / 16; ++ii) {         // 32 columns x 16 rows per pass, no integer division
    const int i = (tid >> 5) + 16 * ii;
    if (i >= npad + 16) continue;
    const bool ci = i < npose && constant && constant[i];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = (tid & 31) + 32 * jj;
      if (j >= npad) continue;
      double v = 0.0;
      if (i < n && j <= i) {
        const bool cj = j < npose && constant && constant[j];
        if (ci || cj) v = (i == j) ? 1.0 : 0.0;
        else {
          v = H0[i * kLd + j] * sc[i] * sc[j];
          if (i == j) v += fmin(fmax(v, 1e-6), 1e32) * inv_radius;
        }
      } else if (i < npad) {
        v = (i == j) ? 1.0 : 0.0;                                  // identity padding keeps the system SPD
      } else if (i == npad && j < n) {
        v = (j < npose && constant && constant[j]) ? 0.0 : -g[j] * sc[j];   // rhs = -S g
      }
      M[i * kMLd + j] = v;
    }
  }
  if (tid == 0) red[8] = 1.0;   // factorisation status
  __syncthreads();
  // H0 becomes the full symmetric undamped matrix (model cost change needs H d)
  for (int e = tid; e < n * n; e += kStepThreads) {
    const int i = e / n, j = e % n;
    if (j > i) H0[i * kLd + j] = H0[j * kLd + i];
  }

  LSTAMP(4);
  if (tid >= n && tid < npad) invd[tid] = 1.0;       // padded unknowns: unit pivots
  // ---- blocked right-looking Cholesky, 16-column panels; the rhs row rides along as one more row below ----------
  // (a) diagonal block p in registers of wave 0: lane r holds row r; column values travel by v_readlane (dense_inl.h:
  //     the next pivot's reciprocal square root runs under the current pivot's updates).
  //     lanes 0-15: rows of the block; lanes 16-31: rows of the identity, which come out as L_pp^-T (kept in the unused
  //     strict upper triangle of the block, its diagonal 1 / L_cc in invd): the panel solve and the back substitution
  //     become products on the matrix cores
  auto diag_block = [&](int p) {
    const int c0 = 16 * p;
    const int rr = lane & 15, grp = lane >> 4;
    double a[16], iv[16];
    const double* src = M + (c0 + rr) * kMLd + c0;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = src[k];                     // (unconditional: a predicated read is a branch each)
    if (grp == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = (k == rr) ? 1.0 : 0.0;
    }
    const bool okp = diag_factor16(a, rr, grp != 1, iv, min(16, n - c0));
    // lower triangle (block rows) and strict upper triangle (identity rows) in one pass of unconditional stores per column
    if (grp < 2) {
      double* dst = M + (c0 + rr) * kMLd + c0;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if ((grp == 0) == (k <= rr)) dst[k] = a[k];
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) invd[c0 + k] = iv[k];
      if (!okp) red[8] = 0.0;
    }
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
    LACC(8, ta1 - ta0); LACC(9, ta2 - ta1); LACC(10, ta3 - ta2);
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
  // ---- backward substitution ds = L^-T y on the matrix cores, wave 0: in row form  X L = Z  with the rhs tile (rows npad ..,
  //      row 0 = y) as Z, panel by panel from the last:  X_p = (Z_p - sum_{q > p} X_q L[q][p]) L_pp^-1  -----------------------
  if (wave == 0) {
    const int m = lane & 15, kk = lane >> 4;
    double* Z = M + npad * kMLd;
    for (int p = NB - 1; p >= 0; --p) {
      d4 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = Z[(kk + 4 * q) * kMLd + 16 * p + m];
      for (int qp = p + 1; qp < NB; ++qp) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Z[m * kMLd + 16 * qp + 4 * s4 + kk], M[(16 * qp + 4 * s4 + kk) * kMLd + 16 * p + m],
                                                     acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) Z[(kk + 4 * q) * kMLd + 16 * p + m] = acc[q];
      d4 xo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 4 * s4 + kk;   // B[k][n = m] = (L_pp^-1)[k][m] = (L_pp^-T)[m][k]
        const double li = (m < k) ? M[(16 * p + m) * kMLd + 16 * p + k] : (m == k ? invd[16 * p + k] : 0.0);
        xo = __builtin_amdgcn_mfma_f64_16x16x4f64(Z[m * kMLd + 16 * p + k], li, xo, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) Z[(kk + 4 * q) * kMLd + 16 * p + m] = xo[q];
    }
    ds[lane] = Z[lane];
    if (lane + 64 < npad) ds[lane + 64] = Z[lane + 64];
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
    double hd = 0.0;
    for (int j = 0; j < n; ++j) hd += H0[tid * kLd + j] * dd[j];
    part = -dd[tid] * g[tid] - 0.5 * dd[tid] * hd;
    dn = dd[tid] * dd[tid];
    const double xv = (tid < npose) ? xf[tid] : S.beta[(size_t)f * nb + tid - npose];
    xn = xv * xv;
  }
  const double model = block_sum8(part, red, tid);
  const double dnorm = sqrt(block_sum8(dn, red, tid));
  const double xnorm = sqrt(block_sum8(xn, red, tid));
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
__global__ __launch_bounds__(kStepThreads) void k_frame_normal(int F, int n, const int* __restrict__ kp_offset, double huber,
                                                       const double* __restrict__ r, const double* __restrict__ J,
                                                       double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Jh = sm;                         // kRowsMax x kJLd : robustified [J | r]
  double* ds = sm + kRowsMax * kJLd;       // per-row sqrt(rho')
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = kp_offset[f], nrows = 2 * (kp_offset[f + 1] - k0);
  const int nrows4 = (nrows + 3) & ~3;
  if (tid < kRowsMax) {
    double sw = 0.0, rr = 0.0;
    if (tid < nrows) {
      const int k = k0 + (tid >> 1);
      const double r0 = r[2 * (size_t)k], r1 = r[2 * (size_t)k + 1];
      double rho1;
      huber_rho(huber, r0 * r0 + r1 * r1, &rho1);
      sw = sqrt(rho1);
      rr = sw * ((tid & 1) ? r1 : r0);
    }
    ds[tid] = sw;
    Jh[tid * kJLd + n] = rr;
  }
  __syncthreads();
  {
    double jv[11];
#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
      jv[u] = (row < nrows && c < n) ? J[(size_t)(2 * k0 + row) * n + c] : 0.0;
    }

#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int i = tid + u * kStepThreads, row = i / 88, c = i % 88;
      if (row < nrows4 && c != n) Jh[row * kJLd + c] = (row < nrows && c < n) ? ds[row] * jv[u] : 0.0;
    }
    for (int i = tid; i < nrows4 * (kJLd - 88); i += kStepThreads) Jh[(i / (kJLd - 88)) * kJLd + 88 + i % (kJLd - 88)] = 0.0;
  }
  __syncthreads();
  double* H = out + (size_t)f * (kN + 1) * kLd;
  const int m = lane & 15, kk = lane >> 4;
  int pair = 0;
  for (int ti = 0; ti < 6; ++ti)
    for (int tj = 0; tj <= ti; ++tj, ++pair) {
      if (pair % kStepWaves != wave) continue;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int s = 0; s < nrows4 / 4; ++s) {
        const double a = Jh[(4 * s + kk) * kJLd + 16 * ti + m];
        const double b = Jh[(4 * s + kk) * kJLd + 16 * tj + m];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
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

size_t lm_step_lds_bytes() { return (size_t)(kMRows * kMLd + (kN + 1) * kLd + 512) * sizeof(double); }

void launch_lm_init(const LmProblem& P, const LmState& S, const double* d_r, hipStream_t s) {
  hipLaunchKernelGGL(k_lm_init, dim3(P.F), dim3(256), 0, s, P, S, d_r);
}
void launch_lm_step(const LmProblem& P, const LmState& S, double* d_r, double* d_J, int* d_comp, const double* d_r_cand,
                    const double* d_J_cand, const int* d_comp_cand, const unsigned char* d_constant, int first_iter,
                    hipStream_t s) {
  static bool attr = false;
  const size_t lds = lm_step_lds_bytes();
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(k_lm_step, dim3(P.F), dim3(kStepThreads), lds, s, P, S, d_r, d_J, d_comp,
                     LmCandidate{d_r_cand, d_J_cand, d_comp_cand}, d_constant, first_iter);
}
void launch_lm_accept(const LmProblem& P, const LmState& S, const double* d_r_new, double* d_r_cur, const int* d_comp_new,
                      int* d_comp_cur, hipStream_t s) {
  hipLaunchKernelGGL(k_lm_accept, dim3(P.F), dim3(256), 0, s, P, S, d_r_new, d_r_cur, d_comp_new, d_comp_cur);
}

void launch_frame_normal(int F, int n, const int* d_kp_offset, double huber, const double* d_r, const double* d_J,
                         double* d_out, hipStream_t s) {
  if (F <= 0) return;
  const size_t lds = (size_t)(kRowsMax * kJLd + kRowsMax) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_normal), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_frame_normal, dim3(F), dim3(kStepThreads), lds, s, F, n, d_kp_offset, huber, d_r, d_J, d_out);
}

}  // namespace bodyfit

#ifdef BODYFIT_STAMPS
extern "C" int bodyfit_debug_set_lm_stamp_buffer(unsigned long long* d_buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(bodyfit::g_lm_dbg), &d_buf, sizeof(d_buf));
}
#endif
