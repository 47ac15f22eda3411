// bodyfit_api.hip — host side of the C ABI in include/bodyfit.h: model upload (operand packing for the
// MFMA kernel), GMM precompute, problem buffers, evaluation sweeps, Ceres-style per-block access.
#include "../../include/bodyfit.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bodyfit_device.h"
#include "collectives.h"
#include "solver_view.h"

using namespace bodyfit;

namespace bodyfit {
std::atomic<long> g_launch_count{0};
}

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(BODYFIT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

// Device blocks of destroyed problems, kept for the next problem of the same shape.  The reference's staged drivers
// (src/main_multi_frame.cpp:109-217: anchors, then one solve per sliding window, an update() after each) create and destroy
// two problems per stage; hipFree synchronises the device and unmaps — 0.6 ms per problem, 12 of the 149 ms of a staged
// 128-frame run (tools/probes/run_multi_breakdown.py).  A block is reused only for a request of exactly its size on its device;
// contents are unspecified, as hipMalloc's are (problem creation clears what it needs cleared).  Bounded: beyond kLimit bytes
// per device a block is freed at once; bodyfit_model_destroy empties its device's list.  Nothing is freed at process exit (the
// runtime may be gone by then).
class BlockPool {
 public:
  static BlockPool& get() { static BlockPool* p = new BlockPool; return *p; }
  void* take(int dev, size_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    auto& f = free_[dev];
    auto it = f.find(bytes);
    if (it == f.end()) return nullptr;
    void* p = it->second;
    f.erase(it);
    held_[dev] -= bytes;
    return p;
  }
  void give(int dev, size_t bytes, void* p) {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (held_[dev] + bytes <= kLimit) {
        free_[dev].emplace(bytes, p);
        held_[dev] += bytes;
        return;
      }
    }
    (void)hipFree(p);
  }
  void trim(int dev) {
    std::multimap<size_t, void*> drop;
    {
      std::lock_guard<std::mutex> g(mu_);
      drop.swap(free_[dev]);
      held_[dev] = 0;
    }
    for (auto& kv : drop) (void)hipFree(kv.second);
  }
 private:
  static constexpr size_t kLimit = (size_t)1 << 30;
  std::mutex mu_;
  std::map<int, std::multimap<size_t, void*>> free_;
  std::map<int, size_t> held_;
};

struct Allocs {
  struct Block { void* p; size_t bytes; int dev; };
  std::vector<Block> blocks;
  bool pooled = false;   // problems: blocks go back to the BlockPool (the owner has synchronised the device first)
  ~Allocs() {
    for (const Block& b : blocks) {
      if (pooled) BlockPool::get().give(b.dev, b.bytes, b.p);
      else (void)hipFree(b.p);
    }
  }
  template <typename T>
  hipError_t alloc(T** out, size_t n) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    int dev = 0;
    (void)hipGetDevice(&dev);
    void* p = pooled ? BlockPool::get().take(dev, bytes) : nullptr;
    hipError_t e = hipSuccess;
    if (!p) e = hipMalloc(&p, bytes);
    if (!p && e != hipSuccess) {
      // out of memory while the pool sits on up to 1 GiB of blocks of other sizes: give them back and try once more
      (void)hipGetLastError();
      BlockPool::get().trim(dev);
      e = hipMalloc(&p, bytes);
    }
    if (e == hipSuccess) {
      blocks.push_back(Block{p, bytes, dev});
      *out = static_cast<T*>(p);
    }
    return e;
  }
  template <typename T>
  hipError_t upload(const T** out, const std::vector<T>& h) {
    T* p = nullptr;
    hipError_t e = alloc(&p, h.size());
    if (e != hipSuccess) return e;
    if (!h.empty()) e = hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    *out = p;
    return e;
  }
};

// a page-locked host buffer (hipHostMalloc): the host side of every copy of the Ceres-kept path.  From pageable memory the
// runtime stages a copy through its own bounce buffers in chunks, synchronously: 9 MB of Jacobian per 256-frame sweep came
// back at ~17 GB/s and the sweep's parameters went up behind a stall; page-locked, both are single DMA transfers that are
// asynchronous on the problem's copy stream.
template <typename T>
struct Pinned {
  T* p = nullptr;
  size_t n = 0;
  ~Pinned() { if (p) (void)hipHostFree(p); }
  hipError_t ensure(size_t want) {
    if (want <= n) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    void* q = nullptr;
    const hipError_t e = hipHostMalloc(&q, std::max<size_t>(want, 1) * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) { p = static_cast<T*>(q); n = want; }
    return e;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

}  // namespace

struct bodyfit_model {
  int device = 0;
  int n_cus = 0;
  int V = 0, nJ = 0, nS = 0, P = 0, nL = 0;   // nL: the caller's one-hot landmarks (the device model's nL counts slots)
  int nReg = 0;                               // sparse keypoint regressors over posed vertices
  std::vector<int> reg_slot;                  // [nReg] first landmark slot of the row's pseudo-vertices
  bool mesh_ok = true;
  DevModel d{};
  std::vector<int> parent;
  std::vector<double> J0, S, offset;
  Allocs mem;
};

struct bodyfit_gmm {
  int device = 0;
  DevGmm d{};
  std::vector<double> prec_cho, neg_log_w, mean, prec;
  Allocs mem;
};

struct bodyfit_problem {
  const bodyfit_model* m = nullptr;
  bodyfit_problem_desc desc{};
  bodyfit_layout lay{};
  DevProblem d{};
  MeshCoef mc{};
  DevGmm gmm{};
  bool has_gmm = false;
  int n_param_rows = 0, n_pairs = 0;
  int row_prior = 0, row_shape = 0, row_temporal = 0;
  // device buffers
  double* d_params = nullptr;
  double* d_beta = nullptr;
  double* d_r = nullptr;
  double* d_J = nullptr;
  double* d_joints = nullptr;
  double* d_partials = nullptr;
  double* d_normal = nullptr;
  int* d_comp = nullptr;
  float* d_cloud = nullptr;
  double* d_frame_normal = nullptr;
  unsigned char* lm_pool = nullptr;    // device LM state of bodyfit_solve, one allocation kept across solves
  unsigned char* win_pool = nullptr;   // device window LM (k_window_lm.hip): state + cyclic-reduction buffers
  size_t win_pool_bytes = 0;
  hipStream_t lm_stream = nullptr;
  double* d_writeback = nullptr;
  // one-launch sweep (k_sweep_roles): in-launch synchronisation words [error | pad | one counter per 32-frame unit], launch counter
  unsigned char* d_fused = nullptr;
  size_t fused_bytes = 0;
  long last_exchanges = 0;             // all-gathers issued by the last sharded solve (tests: exchanges per iteration)
  double exchange_timeout_s = 0.0;     // bodyfit_set_exchange_timeout: bound of one exchange / status read of a sharded solve
  int test_poison_rank = -1, test_poison_iter = -1;   // bodyfit_internal_set_test_poison (tests only)
  int proxy_ranks = 0, proxy_rank = 0;                // bodyfit_set_shard_proxy (measurement aid): 0 = off
  unsigned fused_epoch = 0;
  bool fused_enabled = true, fused_unchecked = false;
  long fused_timeouts = 0;             // one-launch sweeps found incomplete (bodyfit_internal_fused_timeouts)
  hipStream_t async_stream = nullptr;  // stream of the last bodyfit_evaluate_device / bodyfit_reduce_shared_device
  bool async_pending = false;          // ... and whether anything was enqueued there since the last synchronous entry point
  hipEvent_t async_event = nullptr;
  double* d_frame_partials = nullptr; // [F][258] per-frame beta partials written by k_frame_resjac (shared-beta problems)
  int partials_tiles = 0;             // prior tiles that added their plain-cost rows behind the frame rows
  std::vector<double> gmm_jt;   // [K][nJ - 1][prior rows][3]: beta_pose L_k^T per joint block, what a GMM prior block's Jacobian is (host path)
  unsigned long long role_timeout_ticks = kRoleTimeoutDefault;   // bound of the one-launch sweep's in-launch waits
  double* armed_out66 = nullptr;      // bodyfit_arm_shared_reduction: where a folding sweep deposits [cost | g_beta | H_bb]
  unsigned fold_count = 0;            // tickets taken by the folding sweeps since the sync buffer was zeroed
  bool fold_fresh = false;            // the last sweep folded into armed_out66
  bool partials_fresh = false;        // the last sweep produced them (want_jac)      // [F][76] update parameters + [F][9] R0' + [F] mean pixel error, on first use   // [F][87][88] per-frame normal-equation panels (window solver), on first use
  // host copies
  std::vector<int> kp_offset, kp_id, kp_frame;
  std::vector<double> kp_uv;
  // host cache of the last batched evaluation (serves bodyfit_evaluate_block)
  std::mutex mu;
  bool cache_valid = false, cache_has_jac = false;
  // (page-locked mirrors: the sweep's parameters go up from them, its residuals / Jacobian / components come back into them)
  Pinned<double> c_params, c_beta, c_r, c_J;
  // packed form of the cached Jacobian (bodyfit_evaluate_batch without a caller's Jacobian buffer): only the column blocks a
  // probe sweep found non-zero cross PCIe, k_pack_jacobian's layout
  Pinned<double> c_Jp;
  std::vector<unsigned> pk_mask, pk_off;   // [K] block masks, [K + 1] offsets (doubles) into the packed buffer
  std::vector<short> pk_src;               // [K][32]: where block b of keypoint k starts inside the keypoint's packed row, -1: absent
                                           // (bodyfit_evaluate_block_cached serves a block with one table look-up instead of a walk
                                           // over the mask)
  unsigned* d_pk_mask = nullptr;
  unsigned* d_pk_off = nullptr;
  double* d_Jp = nullptr;
  bool pk_ready = false, cache_packed = false;
  Pinned<int> c_comp;
  size_t c_npar = 0, c_nbeta = 0;       // valid entries of c_params / c_beta
  hipStream_t copy_stream = nullptr;   // the Ceres-kept path's own stream (H2D, sweep, D2H)
  Allocs mem;
};

namespace {

int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return (e && *e) ? std::atoi(e) : dflt;
}

bool chol_lower(std::vector<double>& A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
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

// The bounded waits of the one-launch sweep's mesh role set an error word instead of hanging (the tile's part of the cloud is
// then missing; r, J, joints and the folded reduction never depend on a wait).  fused_timed_out reads and clears the word
// (the caller has synchronised the stream the sweep ran on) and switches the problem to the two-launch sweep for the rest of
// its life.  The synchronous entry points re-issue their sweep at once, so their callers never see the event; asynchronous
// callers ask bodyfit_sweep_status.
bool fused_timed_out(bodyfit_problem* p) {
  if (!p->fused_unchecked || !p->d_fused) return false;
  p->fused_unchecked = false;
  unsigned err = 0;
  if (hipMemcpy(&err, p->d_fused, sizeof(err), hipMemcpyDeviceToHost) != hipSuccess) return false;
  if (!err) return false;
  (void)hipMemset(p->d_fused, 0, 4);
  p->fused_enabled = false;
  ++p->fused_timeouts;
  return true;
}
int fused_check(bodyfit_problem* p) {
  if (fused_timed_out(p))
    return fail(BODYFIT_ERR_HIP, "one-launch sweep: an in-launch wait timed out (the cloud of that sweep is incomplete); "
                                 "the problem now uses the two-launch sweep");
  return BODYFIT_OK;
}

// bodyfit_evaluate_batch (and the other entry points that own a stream) may run while asynchronous sweeps of the same problem
// are still in flight on a caller's stream: both write the problem's r / J / partials and the one-launch sweep's counters, so
// they must not overlap.  The asynchronous entry points only note their stream (no event per sweep: that would cost the
// resident path a microsecond per step); the synchronous ones record ONE event behind everything enqueued there so far and
// make their own stream wait for it.
// The caller's stream must stay alive until the problem's next synchronous entry point (or bodyfit_sweep_status on it) has
// returned: the event is recorded on it (include/bodyfit.h, bodyfit_evaluate_device).
int order_after_async(bodyfit_problem* p, hipStream_t own) {
  if (!p->async_pending) return BODYFIT_OK;
  if (p->async_stream != own) {                    // (same stream: ordered anyway)
    if (!p->async_event) HIP_TRY(hipEventCreateWithFlags(&p->async_event, hipEventDisableTiming));
    const hipError_t er = hipEventRecord(p->async_event, p->async_stream);
    if (er == hipErrorInvalidHandle || er == hipErrorInvalidResourceHandle || er == hipErrorContextIsDestroyed) {
      // the caller has destroyed that stream: a stream can only be destroyed once its work is done (hipStreamDestroy waits), so
      // there is nothing left to order behind
      (void)hipGetLastError();
    } else {
      HIP_TRY(er);
      HIP_TRY(hipStreamWaitEvent(own, p->async_event, 0));
    }
  }
  p->async_pending = false;                        // only once the ordering is in place
  return BODYFIT_OK;
}

// One evaluation sweep on the caller's stream: ONE launch (k_sweep_roles: frame, mesh and prior workgroups side by side)
// when the mesh is on, otherwise (no mesh, device LM with frame flags, models with more than 12 landmark slots) two.  The prior residuals are produced by extra
// workgroups (priors_inl.h) of the mesh launch when the mesh is on (its vertex tiles leave 40 CUs idle), otherwise
// of the k_frame_resjac launch.  ev (optional, 4 events): the dispatches' own begin / end timestamps,
// [0],[1] k_frame_resjac, [2],[3] k_mesh_blend_lbs.
int sweep(bodyfit_problem* p, const double* d_params, const double* d_beta, int want_jac, bool mesh,
          hipStream_t st, hipEvent_t* ev = nullptr, double* r_base = nullptr, int* comp_out = nullptr,
          const int* frame_flags = nullptr, int frame_mask = 0, const double* R0_override = nullptr,
          bool skip_priors = false, double* J_base = nullptr) {
  const bodyfit_model* m = p->m;
  DevProblem dp = p->d;
  if (R0_override) dp.R0 = R0_override;
  dp.beta_partials = (want_jac && !frame_flags) ? p->d_frame_partials : nullptr;
  dp.huber = p->desc.huber_delta;
  p->partials_fresh = dp.beta_partials != nullptr;
  p->fold_fresh = false;
  dp.frame_flags = frame_flags;
  dp.frame_mask = frame_mask;
  double* d_r = r_base ? r_base : p->d_r;
  double* d_J = J_base ? J_base : p->d_J;
  int* d_comp = comp_out ? comp_out : p->d_comp;
  MeshCoef mc = p->mc;
  if (!mesh) mc = MeshCoef{};
  const bodyfit_problem_desc& D = p->desc;
  PriorArgs pa{};
  pa.F = p->d.F; pa.nS = m->nS; pa.beta_stride = p->d.beta_stride;
  pa.has_gmm = p->has_gmm ? 1 : 0;
  if (p->has_gmm) pa.g = p->gmm;
  pa.beta_pose = D.beta_pose;
  pa.beta_shape = p->lay.shape_rows > 0 ? D.beta_shape : 0.0;
  pa.lambda_t = D.lambda_temporal;
  pa.n_pairs = p->n_pairs;
  pa.beta = d_beta;
  pa.r_prior = d_r + p->row_prior; pa.r_shape = d_r + p->row_shape; pa.r_temporal = d_r + p->row_temporal;
  pa.comp = d_comp;
  const bool priors = D.beta_pose > 0.0 || pa.beta_shape > 0.0 || D.lambda_temporal > 0.0;
  pa.n_tiles = (priors && !skip_priors) ? (p->d.F + 15) / 16 : 0;
  pa.plain_cost = dp.beta_partials ? dp.beta_partials + (size_t)p->d.F * kReducePartial : nullptr;
  p->partials_tiles = dp.beta_partials ? pa.n_tiles : 0;
  PriorArgs none = pa;
  none.n_tiles = 0;
  if (mesh && !frame_flags && p->fused_enabled && p->d_fused && role_sweep_fits(m->d, dp)) {
    // ONE launch: frame, mesh and prior roles, operands handed over inside the launch (k_sweep.hip)
    FusedSync sy{};
    sy.error = reinterpret_cast<unsigned*>(p->d_fused);
    sy.flag = reinterpret_cast<unsigned*>(p->d_fused + kFusedSyncHeader);
    if (p->fused_epoch >= (1u << 26) || p->fold_count >= (1u << 31)) {
      // epoch x 32 is about to wrap the 32-bit unit counters (or the fold ticket): start over (stream-ordered)
      (void)hipMemsetAsync(p->d_fused, 0, p->fused_bytes, st);
      p->fused_epoch = 0;
      p->fold_count = 0;
    }
    sy.epoch = ++p->fused_epoch;
    sy.resident_blocks = 2 * m->n_cus;
    sy.timeout_ticks = p->role_timeout_ticks;
#ifdef BODYFIT_TUNE_ENV   // diagnostic builds only (tools/probes/sweep_tune.py): the shipped library reads no tuning word from outside
    static const int tune_prio = env_int("BODYFIT_MESH_PRIO", kTuneMeshPrio), tune_start = env_int("BODYFIT_TRICKLE_START", kTuneTrickleStart),
                     tune_sleep = env_int("BODYFIT_TRICKLE_SLEEP", kTuneTrickleSleep), tune_jscope = env_int("BODYFIT_J_SCOPE", kTuneJScope);
#else
    constexpr int tune_prio = kTuneMeshPrio, tune_start = kTuneTrickleStart, tune_sleep = kTuneTrickleSleep, tune_jscope = kTuneJScope;
#endif
    sy.mesh_prio_early = tune_prio; sy.trickle_start = tune_start; sy.trickle_sleep = tune_sleep; sy.j_scope = tune_jscope;
    p->fused_unchecked = true;
    FoldTail fold{};
    const int n_partials = p->d.F + pa.n_tiles;
    if (p->armed_out66 && dp.beta_partials && !p->desc.beta_per_frame && d_beta && n_partials <= kFoldMaxPartials) {
      // the shared-shape reduction rides on this launch's tail (bodyfit_arm_shared_reduction)
      fold.ticket = reinterpret_cast<unsigned*>(p->d_fused + kFoldTicketOffset);
      fold.want = (p->fold_count += (unsigned)n_partials);
      fold.n_partials = n_partials;
      fold.partials = dp.beta_partials;
      fold.beta = d_beta;
      fold.shape_rows = p->lay.shape_rows;
      fold.beta_shape = D.beta_shape;
      fold.out66 = p->armed_out66;
      p->fold_fresh = true;
    }
    launch_sweep_roles(m->d, dp, d_params, d_beta, d_r, want_jac ? d_J : nullptr, p->d_joints, mc, want_jac, pa,
                       p->d_cloud, sy, fold, st, ev ? ev[4] : nullptr, ev ? ev[5] : nullptr);
  } else {
    launch_frame_resjac(m->d, dp, d_params, d_beta, d_r, want_jac ? d_J : nullptr, p->d_joints, mc, want_jac,
                        mesh ? none : pa, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
    if (mesh) launch_mesh(m->d, p->d, p->mc, p->d_cloud, pa, d_params, st, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BODYFIT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return BODYFIT_OK;
}

}  // namespace

extern "C" {

const char* bodyfit_last_error(void) { return g_err.c_str(); }

int bodyfit_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
int bodyfit_model_create(const bodyfit_model_desc* desc, int device, bodyfit_model** out) {
  if (!desc || !out) return fail(BODYFIT_ERR_INVALID, "null argument");
  *out = nullptr;
  const int V = desc->n_verts, nJ = desc->n_joints, nS = desc->n_shape;
  const int P = desc->posedirs ? desc->n_pose_feat : 0;
  const int nL = desc->n_landmarks;
  if (V <= 0 || nJ <= 0 || nJ > kMaxJoints || nS < 0 || nS > kMaxShape)
    return fail(BODYFIT_ERR_INVALID, "unsupported model size (n_joints <= 24, n_shape <= 10)");
  if (P != 0 && P != 9 * (nJ - 1)) return fail(BODYFIT_ERR_INVALID, "n_pose_feat must be 9 (n_joints - 1)");
  if (nL < 0 || nL > kMaxLandmarks) return fail(BODYFIT_ERR_INVALID, "too many landmarks (<= 32)");
  if (!desc->v_template || !desc->shapedirs || !desc->j_regressor || !desc->weights || !desc->parent)
    return fail(BODYFIT_ERR_INVALID, "missing model tensor");
  if (desc->parent[0] != -1) return fail(BODYFIT_ERR_INVALID, "parent[0] must be -1 (npz_fixer convention)");
  for (int j = 1; j < nJ; ++j)
    if (desc->parent[j] < 0 || desc->parent[j] >= j)
      return fail(BODYFIT_ERR_INVALID, "kintree must be topologically ordered with a single root");
  for (int l = 0; l < nL; ++l)
    if (desc->landmark_vid[l] < 0 || desc->landmark_vid[l] >= V) return fail(BODYFIT_ERR_INVALID, "landmark vertex id");
  const int nReg = desc->n_kp_regressors;
  if (nReg < 0 || (nReg > 0 && (!desc->kpreg_offset || !desc->kpreg_vid || !desc->kpreg_weight)))
    return fail(BODYFIT_ERR_INVALID, "keypoint regressors: missing arrays");
  for (int r = 0; r < nReg; ++r) {
    if (desc->kpreg_offset[r + 1] <= desc->kpreg_offset[r] || desc->kpreg_offset[0] != 0)
      return fail(BODYFIT_ERR_INVALID, "keypoint regressors: offsets must start at 0 and every row needs an entry");
    for (int e = desc->kpreg_offset[r]; e < desc->kpreg_offset[r + 1]; ++e)
      if (desc->kpreg_vid[e] < 0 || desc->kpreg_vid[e] >= V) return fail(BODYFIT_ERR_INVALID, "keypoint regressor vertex id");
  }

  HIP_TRY(hipSetDevice(device));
  bodyfit_model* m = new bodyfit_model();
  std::unique_ptr<bodyfit_model> guard(m);
  m->device = device;
  HIP_TRY(hipDeviceGetAttribute(&m->n_cus, hipDeviceAttributeMultiprocessorCount, device));
  m->V = V; m->nJ = nJ; m->nS = nS; m->P = P; m->nL = nL;
  m->parent.assign(desc->parent, desc->parent + nJ);

  // joint regression on the device: J0 = Jreg . v_template, S = Jreg . shapedirs
  m->J0.assign((size_t)nJ * 3, 0.0);
  m->S.assign((size_t)nJ * 3 * std::max(nS, 1), 0.0);
  {
    Allocs tmp;
    double *d_reg, *d_vt, *d_sd, *d_j0, *d_s;
    HIP_TRY(tmp.alloc(&d_reg, (size_t)nJ * V));
    HIP_TRY(tmp.alloc(&d_vt, (size_t)V * 3));
    HIP_TRY(tmp.alloc(&d_sd, (size_t)V * 3 * std::max(nS, 1)));
    HIP_TRY(tmp.alloc(&d_j0, (size_t)nJ * 3));
    HIP_TRY(tmp.alloc(&d_s, (size_t)nJ * 3 * std::max(nS, 1)));
    HIP_TRY(hipMemcpy(d_reg, desc->j_regressor, (size_t)nJ * V * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_vt, desc->v_template, (size_t)V * 3 * sizeof(double), hipMemcpyHostToDevice));
    launch_regress(nJ, V, 3, d_reg, d_vt, d_j0, nullptr);
    if (nS > 0) {
      HIP_TRY(hipMemcpy(d_sd, desc->shapedirs, (size_t)V * 3 * nS * sizeof(double), hipMemcpyHostToDevice));
      launch_regress(nJ, V, 3 * nS, d_reg, d_sd, d_s, nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(m->J0.data(), d_j0, (size_t)nJ * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (nS > 0)
      HIP_TRY(hipMemcpy(m->S.data(), d_s, (size_t)nJ * 3 * nS * sizeof(double), hipMemcpyDeviceToHost));
  }
  const std::vector<double>& J0 = m->J0;
  const std::vector<double>& S = m->S;

  // rest offsets (include/Sim3BA.h:372-392) and the shape-difference tables
  m->offset.assign((size_t)nJ * 3, 0.0);
  std::vector<double> Jc0((size_t)nJ * 3), dS((size_t)nJ * 3 * std::max(nS, 1), 0.0), Sc(dS.size(), 0.0);
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) Jc0[j * 3 + a] = J0[j * 3 + a] - J0[a];
  for (int j = 1; j < nJ; ++j)
    for (int a = 0; a < 3; ++a) m->offset[j * 3 + a] = Jc0[j * 3 + a] - Jc0[m->parent[j] * 3 + a];
  for (int j = 0; j < nJ; ++j)
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < nS; ++k) {
        const double sj = S[(size_t)(j * 3 + a) * nS + k];
        const int pj = m->parent[j];
        dS[(size_t)(j * 3 + a) * nS + k] = sj - (pj >= 0 ? S[(size_t)(pj * 3 + a) * nS + k] : 0.0);
        Sc[(size_t)(j * 3 + a) * nS + k] = sj - S[(size_t)a * nS + k];
      }
  // depth levels and ancestor masks
  std::vector<int> depth(nJ, 0);
  int maxd = 0;
  for (int j = 1; j < nJ; ++j) { depth[j] = depth[m->parent[j]] + 1; maxd = std::max(maxd, depth[j]); }
  std::vector<int> level_off(maxd + 1, 0), level_joint;
  for (int d = 1; d <= maxd; ++d) {
    level_off[d - 1] = (int)level_joint.size();
    for (int j = 1; j < nJ; ++j)
      if (depth[j] == d) level_joint.push_back(j);
  }
  level_off[maxd] = (int)level_joint.size();
  std::vector<unsigned> anc(nJ, 0u);
  for (int j = 1; j < nJ; ++j)
    for (int k = m->parent[j]; k > 0; k = m->parent[k]) anc[j] |= (1u << k);

  DevModel& d = m->d;
  d.V = V; d.nJ = nJ; d.nS = nS; d.P = P; d.nL = nL; d.nLevels = maxd;
  d.nVTiles = (V + kVTile - 1) / kVTile;
  // (parent, anc_mask, anc_chain, offset, dS, Jc0, Sc and the landmark tables: one block, bodyfit_device.h kTab*)
  std::vector<unsigned char> tabA(kTabBytes, 0);
  auto put = [&](int off, const void* src, size_t bytes) { if (bytes) std::memcpy(tabA.data() + off, src, bytes); };
  if (nJ > kMaxJoints || nS > kMaxShape) return fail(BODYFIT_ERR_INVALID, "model: at most 24 joints and 10 shape coefficients");
  put(kTabParent, m->parent.data(), m->parent.size() * sizeof(int));
  HIP_TRY(m->mem.upload(&d.level_off, level_off));
  HIP_TRY(m->mem.upload(&d.level_joint, level_joint));
  put(kTabAnc, anc.data(), anc.size() * sizeof(unsigned));
  // the same ancestors as a packed walk list: nearest first, 5 bits each, 0-terminated (joint ids 1..23; depth <= 12)
  std::vector<unsigned long long> chain(nJ, 0ull);
  for (int j = 1; j < nJ; ++j) {
    int lvl = 0;
    for (int k = m->parent[j]; k > 0 && lvl < 12; k = m->parent[k], ++lvl) chain[j] |= (unsigned long long)k << (5 * lvl);
  }
  put(kTabChain, chain.data(), chain.size() * sizeof(unsigned long long));
  put(kTabOffset, m->offset.data(), m->offset.size() * sizeof(double));
  put(kTabDS, dS.data(), dS.size() * sizeof(double));
  put(kTabJc0, Jc0.data(), Jc0.size() * sizeof(double));
  put(kTabSc, Sc.data(), Sc.size() * sizeof(double));

  // landmark slots of the frame kernel: the caller's one-hot landmarks, then the pseudo-vertices of the regressor rows.
  // A row  k = sum_i a_i posed(v_i),  posed(v) = sum_j W_vj (A_j (rest_v - Jc_j) + P_j),  collapses per skinning joint j to
  //   s_j (A_j (r_j - Jc_j) + P_j),   s_j = sum_i a_i W_ij,   r_j = sum_i a_i W_ij rest_i / s_j
  // (rest_i = template + shapedirs beta + posedirs feat is linear in the vertex rows, and the coefficients a_i W_ij / s_j
  // sum to one, so r_j is itself a "vertex" with rows combined the same way): one slot per joint with s_j != 0, skinned to
  // that joint alone with weight s_j.  The kernel adds the slots of a row up (position and Jacobian terms) into the first.
  {
    struct Slot { std::vector<std::pair<int, double>> w; std::vector<std::pair<int, double>> src; };   // (joint, weight), (vertex, coefficient)
    std::vector<Slot> slots;
    std::vector<int> gcount;
    for (int l = 0; l < nL; ++l) {
      Slot sl;
      const int vid = desc->landmark_vid[l];
      for (int j = 0; j < nJ; ++j) {
        const double w = desc->weights[(size_t)vid * nJ + j];
        if (w != 0.0) sl.w.emplace_back(j, w);
      }
      if ((int)sl.w.size() > kMaxLmNnz) return fail(BODYFIT_ERR_INVALID, "landmark vertex has more than 8 skinning weights");
      sl.src.emplace_back(vid, 1.0);
      slots.push_back(std::move(sl));
      gcount.push_back(1);
    }
    m->nReg = nReg;
    for (int r = 0; r < nReg; ++r) {
      m->reg_slot.push_back((int)slots.size());
      const int first = (int)slots.size();
      for (int j = 0; j < nJ; ++j) {
        double sj = 0.0;
        for (int e = desc->kpreg_offset[r]; e < desc->kpreg_offset[r + 1]; ++e)
          sj += desc->kpreg_weight[e] * desc->weights[(size_t)desc->kpreg_vid[e] * nJ + j];
        if (sj == 0.0) continue;
        Slot sl;
        sl.w.emplace_back(j, sj);
        for (int e = desc->kpreg_offset[r]; e < desc->kpreg_offset[r + 1]; ++e) {
          const double c = desc->kpreg_weight[e] * desc->weights[(size_t)desc->kpreg_vid[e] * nJ + j];
          if (c != 0.0) sl.src.emplace_back(desc->kpreg_vid[e], c / sj);
        }
        slots.push_back(std::move(sl));
        gcount.push_back(0);
      }
      if ((int)slots.size() == first) return fail(BODYFIT_ERR_INVALID, "keypoint regressor row without skinning weight");
      gcount[first] = (int)slots.size() - first;
    }
    const int nSlots = (int)slots.size();
    if (nSlots > kMaxLandmarks)
      return fail(BODYFIT_ERR_INVALID, "landmarks + regressor pseudo-vertices (one per row and skinning joint) must be <= 32");
    d.nL = nSlots;
    // fixed-stride skinning weights per slot: kMaxLmNnz entries padded with weight 0; woff[l] = count
    std::vector<int> woff(nSlots + 1, 0), wj((size_t)std::max(nSlots, 1) * kMaxLmNnz, 0);
    std::vector<double> ww((size_t)std::max(nSlots, 1) * kMaxLmNnz, 0.0), vt((size_t)nSlots * 3, 0.0),
        sd((size_t)nSlots * 3 * std::max(nS, 1), 0.0), pd((size_t)std::max(nSlots, 1) * 27 * 32, 0.0);
    for (int l = 0; l < nSlots; ++l) {
      const Slot& sl = slots[l];
      for (size_t i = 0; i < sl.w.size(); ++i) {
        wj[(size_t)l * kMaxLmNnz + i] = sl.w[i].first;
        ww[(size_t)l * kMaxLmNnz + i] = sl.w[i].second;
      }
      woff[l] = (int)sl.w.size();
      for (const auto& [vid, c] : sl.src)
        for (int a = 0; a < 3; ++a) {
          vt[l * 3 + a] += c * (desc->v_template[(size_t)vid * 3 + a] - J0[a]);
          for (int k = 0; k < nS; ++k)
            sd[(size_t)(l * 3 + a) * nS + k] += c * (desc->shapedirs[((size_t)vid * 3 + a) * nS + k] - S[(size_t)a * nS + k]);
          // [l][a * 9 + e][k - 1]: the 9 (k - 1) + e pose-feature column of joint k, joint-minor, so that the lanes of a
          // half-wave (one joint each) read 184 contiguous bytes per (a, e)
          for (int k = 0; k < P; ++k)
            pd[((size_t)l * 27 + a * 9 + k % 9) * 32 + k / 9] += c * desc->posedirs[((size_t)vid * 3 + a) * P + k];
        }
    }
    put(kTabLmWoff, woff.data(), woff.size() * sizeof(int));
    put(kTabLmWj, wj.data(), wj.size() * sizeof(int));
    put(kTabLmWw, ww.data(), ww.size() * sizeof(double));
    put(kTabLmVt, vt.data(), vt.size() * sizeof(double));
    put(kTabLmSd, sd.data(), sd.size() * sizeof(double));
    {
      const unsigned char* dev = nullptr;
      HIP_TRY(m->mem.upload(&dev, tabA));
      d.tabA = dev;
      d.parent = reinterpret_cast<const int*>(dev + kTabParent);
      d.anc_mask = reinterpret_cast<const unsigned*>(dev + kTabAnc);
      d.anc_chain = reinterpret_cast<const unsigned long long*>(dev + kTabChain);
      d.offset = reinterpret_cast<const double*>(dev + kTabOffset);
      d.dS = reinterpret_cast<const double*>(dev + kTabDS);
      d.Jc0 = reinterpret_cast<const double*>(dev + kTabJc0);
      d.Sc = reinterpret_cast<const double*>(dev + kTabSc);
      d.lm_woff = reinterpret_cast<const int*>(dev + kTabLmWoff);
      d.lm_wj = reinterpret_cast<const int*>(dev + kTabLmWj);
      d.lm_ww = reinterpret_cast<const double*>(dev + kTabLmWw);
      d.lm_vt = reinterpret_cast<const double*>(dev + kTabLmVt);
      d.lm_sd = reinterpret_cast<const double*>(dev + kTabLmSd);
    }
    HIP_TRY(m->mem.upload(&d.lm_pd, pd));
    d.lm_gcount = nullptr;
    if (nReg > 0) {
      gcount.resize(kMaxLandmarks, 0);
      HIP_TRY(m->mem.upload(&d.lm_gcount, gcount));
    }
  }

  // mesh operands in MFMA fragment order
  {
    const int nVT = d.nVTiles;
    std::vector<uint16_t> dirsB((size_t)nVT * kBlendKSteps * 3 * 2 * 64 * 8, 0);
    std::vector<float> vtB((size_t)nVT * 3 * 32, 0.0f);
    auto put = [&](int vt_i, int c, int col, int k, float x) {   // 32x32x16: B[k = 16 ks + 8 h + j][col]
      const int ks = k >> 4, hh = (k >> 3) & 1, jj = k & 7;
      const uint16_t hi = f32_to_bf16(x);
      const uint16_t lo = f32_to_bf16(x - bf16_to_f32(hi));
      const size_t base = ((((size_t)vt_i * kBlendKSteps + ks) * 3 + c) * 2) * 64 * 8;
      dirsB[base + (size_t)(hh * 32 + col) * 8 + jj] = hi;
      dirsB[base + (size_t)64 * 8 + (size_t)(hh * 32 + col) * 8 + jj] = lo;
    };
    std::vector<uint32_t> wIdx((size_t)nVT * 32, 0u);
    std::vector<float> wVal((size_t)nVT * 32 * 4, 0.0f);
    for (int vt_i = 0; vt_i < nVT; ++vt_i)
      for (int col = 0; col < 32; ++col) {
        const int v = vt_i * 32 + col;
        if (v >= V) continue;
        for (int c = 0; c < 3; ++c) {
          vtB[((size_t)vt_i * 3 + c) * 32 + col] = (float)(desc->v_template[(size_t)v * 3 + c] - J0[c]);
          {
            // the template rides in the contraction on two of its padding slots (coefficient 1 in k_frame_resjac's
            // fragments): slot 217 carries its first 16 bits (bf16 hi + lo), slot 218 what those left over
            const float t0 = vtB[((size_t)vt_i * 3 + c) * 32 + col];
            const uint16_t h0 = f32_to_bf16(t0);
            const uint16_t l0 = f32_to_bf16(t0 - bf16_to_f32(h0));
            put(vt_i, c, col, kPoseFeat + kMaxShape, t0);
            put(vt_i, c, col, kPoseFeat + kMaxShape + 1, t0 - (bf16_to_f32(h0) + bf16_to_f32(l0)));
          }
          for (int k = 0; k < nS && k < kMaxShape; ++k)
            put(vt_i, c, col, kPoseFeat + k, (float)(desc->shapedirs[((size_t)v * 3 + c) * nS + k] - S[(size_t)c * nS + k]));
          for (int k = 0; k < P && k < kPoseFeat; ++k) put(vt_i, c, col, k, (float)desc->posedirs[((size_t)v * 3 + c) * P + k]);
        }
      }
    // Skinning weights: <= 4 (joint, weight) entries per vertex, lane = vertex.  The kernel's i-th ds_read_b128 of a row
    // has each lane fetch 16 B of its i-th joint's transform (48-byte records), served in groups of 16 lanes; two lanes
    // of a group collide on LDS banks exactly when their i-th joints differ by 16.  The order of a vertex's entries is
    // free, so it is chosen (greedily, per tile) to keep "joint mod 16" unique per group and slot; unused slots take
    // weight 0 and a joint that broadcasts or falls on a free residue.
    const int group_of_col[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    for (int vt_i = 0; vt_i < nVT; ++vt_i) {
      int used[2][kMeshNnz][16];   // joint + 1 occupying residue r of (group, slot), 0 = free
      std::memset(used, 0, sizeof(used));
      for (int col = 0; col < 32; ++col) {
        const int v = vt_i * 32 + col, g = group_of_col[col];
        int js[kMeshNnz], cnt = 0;
        double ws[kMeshNnz];
        if (v < V)
          for (int j = 0; j < nJ; ++j) {
            const double w = desc->weights[(size_t)v * nJ + j];
            if (w == 0.0) continue;
            if (cnt < kMeshNnz) { js[cnt] = j; ws[cnt] = w; }
            ++cnt;
          }
        if (cnt > kMeshNnz) { m->mesh_ok = false; cnt = kMeshNnz; }
        int perm[kMeshNnz] = {0, 1, 2, 3}, best[kMeshNnz] = {0, 1, 2, 3}, best_cost = 1 << 30;
        do {   // slot perm[e] receives entry e (e < cnt)
          int cost = 0;
          for (int e = 0; e < cnt; ++e) {
            const int u = used[g][perm[e]][js[e] & 15];
            cost += (u != 0 && u != js[e] + 1);
          }
          if (cost < best_cost) { best_cost = cost; std::copy(perm, perm + kMeshNnz, best); }
        } while (best_cost > 0 && std::next_permutation(perm, perm + kMeshNnz));
        int slot_j[kMeshNnz];
        float slot_w[kMeshNnz];
        bool filled[kMeshNnz] = {false, false, false, false};
        for (int e = 0; e < cnt; ++e) { slot_j[best[e]] = js[e]; slot_w[best[e]] = (float)ws[e]; filled[best[e]] = true; }
        for (int sl = 0; sl < kMeshNnz; ++sl) {
          if (!filled[sl]) {
            int pick = -1;
            for (int r = 0; r < 16 && pick < 0; ++r) if (used[g][sl][r]) pick = used[g][sl][r] - 1;   // broadcast
            slot_j[sl] = pick < 0 ? 0 : pick;
            slot_w[sl] = 0.0f;
          }
          if (!used[g][sl][slot_j[sl] & 15]) used[g][sl][slot_j[sl] & 15] = slot_j[sl] + 1;
        }
        uint32_t packed = 0;
        for (int sl = 0; sl < kMeshNnz; ++sl) {
          packed |= ((uint32_t)slot_j[sl]) << (8 * sl);
          wVal[((size_t)vt_i * 32 + col) * 4 + sl] = slot_w[sl];
        }
        wIdx[(size_t)vt_i * 32 + col] = packed;
      }
    }
    HIP_TRY(m->mem.upload(&d.dirsB, dirsB));
    HIP_TRY(m->mem.upload(&d.vtB, vtB));
    HIP_TRY(m->mem.upload(&d.wIdx, wIdx));
    HIP_TRY(m->mem.upload(&d.wVal, wVal));
  }
  *out = guard.release();
  return BODYFIT_OK;
}

void bodyfit_model_destroy(bodyfit_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  BlockPool::get().trim(m->device);   // (blocks of this device's destroyed problems)
  delete m;
}

int bodyfit_model_get_derived(const bodyfit_model* m, double* joints0, double* joint_shape_reg, double* offset) {
  if (!m) return fail(BODYFIT_ERR_INVALID, "null model");
  if (joints0) std::memcpy(joints0, m->J0.data(), (size_t)m->nJ * 3 * sizeof(double));
  if (joint_shape_reg) std::memcpy(joint_shape_reg, m->S.data(), (size_t)m->nJ * 3 * m->nS * sizeof(double));
  if (offset) std::memcpy(offset, m->offset.data(), (size_t)m->nJ * 3 * sizeof(double));
  return BODYFIT_OK;
}

// ------------------------------------------------------------------------------------------------
// GMM (ark::GaussianMixture restated: precision Cholesky + SMPLify max-mixture constants)
// ------------------------------------------------------------------------------------------------
int bodyfit_gmm_create(int K, int D, const double* weights, const double* means, const double* covs,
                       double resid_scale, int device, bodyfit_gmm** out) {
  if (!out || !weights || !means || !covs || K <= 0 || K > 8 || D <= 0 || D > 72)
    return fail(BODYFIT_ERR_INVALID, "bad GMM arguments (1..8 components, dimension <= 72)");
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<bodyfit_gmm> g(new bodyfit_gmm());
  g->device = device;
  g->prec_cho.assign((size_t)K * D * D, 0.0);
  g->neg_log_w.assign(K, 0.0);
  g->mean.assign(means, means + (size_t)K * D);
  std::vector<double> hld(K);
  for (int k = 0; k < K; ++k) {
    std::vector<double> C(covs + (size_t)k * D * D, covs + (size_t)(k + 1) * D * D);
    if (!chol_lower(C, D)) return fail(BODYFIT_ERR_NUMERIC, "GMM covariance is not SPD");
    double ld = 0;
    for (int i = 0; i < D; ++i) ld += std::log(C[(size_t)i * D + i]);
    hld[k] = ld;
    // Y = C^{-1} by forward substitution on the identity; precision = Y^T Y
    std::vector<double> Y((size_t)D * D, 0.0);
    for (int c = 0; c < D; ++c)
      for (int r = c; r < D; ++r) {
        double s = (r == c) ? 1.0 : 0.0;
        for (int t = c; t < r; ++t) s -= C[(size_t)r * D + t] * Y[(size_t)t * D + c];
        Y[(size_t)r * D + c] = s / C[(size_t)r * D + r];
      }
    std::vector<double> Pm((size_t)D * D, 0.0);
    for (int r = 0; r < D; ++r)
      for (int c = 0; c <= r; ++c) {
        double s = 0;
        for (int t = r; t < D; ++t) s += Y[(size_t)t * D + r] * Y[(size_t)t * D + c];
        Pm[(size_t)r * D + c] = s;
        Pm[(size_t)c * D + r] = s;
      }
    g->prec.insert(g->prec.end(), Pm.begin(), Pm.end());
    if (!chol_lower(Pm, D)) return fail(BODYFIT_ERR_NUMERIC, "GMM precision is not SPD");
    std::memcpy(&g->prec_cho[(size_t)k * D * D], Pm.data(), (size_t)D * D * sizeof(double));
  }
  const double mn = *std::min_element(hld.begin(), hld.end());
  for (int k = 0; k < K; ++k)
    g->neg_log_w[k] = -(std::log(weights[k]) - 0.5 * D * std::log(2.0 * M_PI) - (hld[k] - mn));
  g->d.K = K; g->d.D = D; g->d.resid_scale = resid_scale;
  HIP_TRY(g->mem.upload(&g->d.mean, g->mean));
  HIP_TRY(g->mem.upload(&g->d.prec_cho, g->prec_cho));
  HIP_TRY(g->mem.upload(&g->d.prec, g->prec));
  {
    // B-fragment order of v_mfma_f64_16x16x4_f64 (B[k = lane>>4][j = lane&15]), two column tiles per
    // 16-byte load: frag[k][ks][pair][lane][t] = L[4 ks + (lane>>4)][16 (2 pair + t) + (lane&15)], zero padded
    std::vector<double> frag((size_t)K * 18 * 3 * 64 * 2, 0.0);
    for (int k = 0; k < K; ++k)
      for (int ks = 0; ks < 18; ++ks)
        for (int pr = 0; pr < 3; ++pr)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 2; ++t) {
              const int r = 4 * ks + (lane >> 4), c = 16 * (2 * pr + t) + (lane & 15);
              if (r < D && c < D && 2 * pr + t < 5)
                frag[((((size_t)k * 18 + ks) * 3 + pr) * 64 + lane) * 2 + t] = g->prec_cho[((size_t)k * D + r) * D + c];
            }
    HIP_TRY(g->mem.upload(&g->d.prec_frag, frag));
  }
  HIP_TRY(g->mem.upload(&g->d.neg_log_w, g->neg_log_w));
  *out = g.release();
  return BODYFIT_OK;
}

void bodyfit_gmm_destroy(bodyfit_gmm* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  delete g;
}

int bodyfit_gmm_get(const bodyfit_gmm* g, double* prec_cho, double* neg_log_w) {
  if (!g) return fail(BODYFIT_ERR_INVALID, "null gmm");
  if (prec_cho) std::memcpy(prec_cho, g->prec_cho.data(), g->prec_cho.size() * sizeof(double));
  if (neg_log_w) std::memcpy(neg_log_w, g->neg_log_w.data(), g->neg_log_w.size() * sizeof(double));
  return BODYFIT_OK;
}

// ------------------------------------------------------------------------------------------------
// problem
// ------------------------------------------------------------------------------------------------
int bodyfit_problem_create(const bodyfit_model* m, const bodyfit_problem_desc* desc, bodyfit_problem** out) {
  if (!m || !desc || !out) return fail(BODYFIT_ERR_INVALID, "null argument");
  *out = nullptr;
  const int F = desc->n_frames, nJ = m->nJ, nS = m->nS;
  const int npose = 7 + 3 * (nJ - 1);
  if (F <= 0 || !desc->kp_offset || !desc->R0) return fail(BODYFIT_ERR_INVALID, "bad frame arrays");
  if (desc->n_cols != npose && desc->n_cols != npose + nS) return fail(BODYFIT_ERR_INVALID, "n_cols must be 76 or 76 + n_shape");
  if (desc->use_shape && desc->n_cols == npose) return fail(BODYFIT_ERR_INVALID, "use_shape needs the shape block (n_cols = 86)");
  if (desc->kp_offset[0] != 0) return fail(BODYFIT_ERR_INVALID, "kp_offset[0] must be 0");
  for (int f = 0; f < F; ++f)
    if (desc->kp_offset[f + 1] < desc->kp_offset[f]) return fail(BODYFIT_ERR_INVALID, "kp_offset must be non-decreasing");
  const int K = desc->kp_offset[F];
  if (K > 0 && (!desc->kp_id || !desc->kp_uv)) return fail(BODYFIT_ERR_INVALID, "missing keypoints");
  for (int k = 0; k < K; ++k)
    if (desc->kp_id[k] < 0 || desc->kp_id[k] >= nJ + m->nL + m->nReg) return fail(BODYFIT_ERR_INVALID, "keypoint id out of range");
  if (desc->gmm && desc->gmm->d.D != 3 * (nJ - 1)) return fail(BODYFIT_ERR_INVALID, "GMM dimension must be 3 (n_joints - 1)");
  if (desc->want_mesh && (size_t)((F + kFTile - 1) / kFTile) * kFTile * m->d.nVTiles * kVTile * 12 >= ((size_t)1 << 32))
    return fail(BODYFIT_ERR_INVALID, "mesh path: the cloud of one problem must stay below 4 GiB (split the frames)");
  if (desc->want_mesh && !m->mesh_ok)
    return fail(BODYFIT_ERR_INVALID, "mesh path needs <= 4 skinning weights per vertex");
  if (desc->want_mesh && (nJ != 24 || (m->P != 0 && m->P != 207) || nS > 10))
    return fail(BODYFIT_ERR_INVALID, "mesh path is built for the SMPL shape (24 joints, 207 pose features)");

  HIP_TRY(hipSetDevice(m->device));
  std::unique_ptr<bodyfit_problem> p(new bodyfit_problem());
  // a creation that fails half way returns its blocks to the pool, like bodyfit_problem_destroy: behind a device synchronisation
  // (memsets and uploads may still be in flight on them; later takers use non-blocking streams).  Declared after p: runs first.
  struct SyncOnFailure { std::unique_ptr<bodyfit_problem>& q; ~SyncOnFailure() { if (q) (void)hipDeviceSynchronize(); } } sync_on_failure{p};
  p->mem.pooled = true;
  p->m = m;
  p->desc = *desc;
  p->desc.kp_offset = nullptr; p->desc.kp_id = nullptr; p->desc.kp_uv = nullptr; p->desc.R0 = nullptr;
  p->kp_offset.assign(desc->kp_offset, desc->kp_offset + F + 1);
  p->kp_id.assign(desc->kp_id, desc->kp_id + K);
  p->kp_uv.assign(desc->kp_uv, desc->kp_uv + (size_t)2 * K);
  p->kp_frame.resize(K);
  for (int f = 0; f < F; ++f)
    for (int k = p->kp_offset[f]; k < p->kp_offset[f + 1]; ++k) p->kp_frame[k] = f;
  p->has_gmm = desc->gmm != nullptr && desc->beta_pose > 0.0;
  if (p->has_gmm) p->gmm = desc->gmm->d;

  bodyfit_layout& L = p->lay;
  L.n_keypoints = K;
  L.n_cols = desc->n_cols;
  L.reproj_rows = 2 * K;
  L.prior_rows_per_frame = desc->beta_pose > 0.0 ? (p->has_gmm ? 3 * (nJ - 1) + 1 : 3 * (nJ - 1)) : 0;
  const bool has_beta = desc->n_cols > npose;
  L.shape_rows = (desc->beta_shape > 0.0 && has_beta && nS > 0) ? (desc->beta_per_frame ? F * nS : nS) : 0;
  p->n_pairs = desc->lambda_temporal > 0.0 ? (F - 1 + (desc->temporal_halo ? 1 : 0)) : 0;
  L.temporal_rows = p->n_pairs * (6 + 3 * (nJ - 1));
  if (p->has_gmm) {
    // the GMM prior block's Jacobian per mixture component and joint block, in Ceres' layout (include/Sim3BA.h:293-299)
    const int D = 3 * (nJ - 1), nRes = L.prior_rows_per_frame;
    const size_t Kc = desc->gmm->prec_cho.size() / ((size_t)D * D);
    p->gmm_jt.assign(Kc * (nJ - 1) * nRes * 3, 0.0);
    for (size_t k = 0; k < Kc; ++k) {
      const double* Lk = desc->gmm->prec_cho.data() + k * D * D;
      for (int j = 0; j < nJ - 1; ++j) {
        double* Jb = p->gmm_jt.data() + (k * (nJ - 1) + j) * nRes * 3;
        for (int row = 0; row < D; ++row)
          for (int c = 0; c < 3; ++c) Jb[(size_t)row * 3 + c] = Lk[(size_t)(3 * j + c) * D + row] * desc->beta_pose;
      }
    }
  }
  p->row_prior = L.reproj_rows;
  p->row_shape = p->row_prior + F * L.prior_rows_per_frame;
  p->row_temporal = p->row_shape + L.shape_rows;
  L.total_rows = p->row_temporal + L.temporal_rows;
  p->n_param_rows = F + (desc->temporal_halo ? 1 : 0);

  DevProblem& d = p->d;
  d.F = F; d.K = K; d.ncols = desc->n_cols; d.use_shape = desc->use_shape ? 1 : 0;
  d.beta_stride = desc->beta_per_frame ? nS : 0;
  d.pose_blend = (desc->pose_blend && m->P > 0) ? 1 : 0;
  d.nFTiles = (F + kFTile - 1) / kFTile;
  d.fx = desc->fx; d.fy = desc->fy; d.cx = desc->cx; d.cy = desc->cy;
  {
    // the device copies carry one keypoint chunk (32 entries) of zero padding: k_frame_resjac prefetches a frame's first
    // chunk with unconditional loads.  One block for the three tables (bodyfit_device.h ptab_*_off).
    std::vector<int> ids(p->kp_id);
    for (int& id : ids)        // a regressor row is addressed by the first of its landmark slots on the device
      if (id >= nJ + m->nL) id = nJ + m->reg_slot[id - nJ - m->nL];
    std::vector<double> uv(p->kp_uv);
    ids.resize(ids.size() + 32, 0);
    uv.resize(uv.size() + 64, 0.0);
    const int id_off = ptab_id_off(F), uv_off = ptab_uv_off(F, K);
    std::vector<unsigned char> ptab((size_t)uv_off + uv.size() * sizeof(double), 0);
    std::memcpy(ptab.data(), p->kp_offset.data(), p->kp_offset.size() * sizeof(int));
    std::memcpy(ptab.data() + id_off, ids.data(), ids.size() * sizeof(int));
    std::memcpy(ptab.data() + uv_off, uv.data(), uv.size() * sizeof(double));
    const unsigned char* dev = nullptr;
    HIP_TRY(p->mem.upload(&dev, ptab));
    d.ptab = dev;
    d.kp_offset = reinterpret_cast<const int*>(dev);
    d.kp_id = reinterpret_cast<const int*>(dev + id_off);
    d.kp_uv = reinterpret_cast<const double*>(dev + uv_off);
  }
  std::vector<double> R0(desc->R0, desc->R0 + (size_t)F * 9);
  HIP_TRY(p->mem.upload(&d.R0, R0));

  HIP_TRY(p->mem.alloc(&p->d_params, (size_t)p->n_param_rows * npose));
  HIP_TRY(p->mem.alloc(&p->d_beta, (size_t)std::max(1, desc->beta_per_frame ? F * nS : nS)));
  HIP_TRY(p->mem.alloc(&p->d_r, (size_t)L.total_rows));
  HIP_TRY(p->mem.alloc(&p->d_J, (size_t)L.reproj_rows * L.n_cols));
  HIP_TRY(p->mem.alloc(&p->d_joints, (size_t)F * nJ * 3));
  HIP_TRY(p->mem.alloc(&p->d_comp, (size_t)F));
  HIP_TRY(p->mem.alloc(&p->d_partials, (size_t)reduce_partials_doubles()));
  if (L.n_cols > npose && !desc->beta_per_frame && nS == kMaxShape) {
    const size_t nfp = (size_t)(F + (F + 15) / 16) * kReducePartial;   // one row per frame + one per prior tile
    HIP_TRY(p->mem.alloc(&p->d_frame_partials, nfp));
    HIP_TRY(hipMemset(p->d_frame_partials, 0, nfp * sizeof(double)));
  }
  HIP_TRY(p->mem.alloc(&p->d_normal, (size_t)66));
  HIP_TRY(hipMemset(p->d_r, 0, (size_t)std::max(1, L.total_rows) * sizeof(double)));
  HIP_TRY(hipMemset(p->d_comp, 0, (size_t)F * sizeof(int)));
  HIP_TRY(hipMemset(p->d_beta, 0, (size_t)std::max(1, desc->beta_per_frame ? F * nS : nS) * sizeof(double)));
  if (desc->want_mesh) {
    const size_t nfa = (size_t)d.nFTiles * kBlendKSteps * 2 * 64 * 8;
    HIP_TRY(p->mem.alloc(&p->mc.featA, nfa));
    const size_t nsk = (size_t)d.nFTiles * kFTile * nJ * 12;   // whole frame tiles, zero beyond F
    HIP_TRY(p->mem.alloc(&p->mc.skinT, nsk));
    HIP_TRY(hipMemset(p->mc.skinT, 0, nsk * sizeof(float)));
    // frames padded to whole 32-frame tiles, each frame to whole 32-vertex tiles: k_mesh_blend_lbs stores
    // unconditionally, whole 128-byte lines per half-wave
    HIP_TRY(p->mem.alloc(&p->d_cloud, (size_t)d.nFTiles * kFTile * m->d.nVTiles * kVTile * 3));
    HIP_TRY(hipMemset(p->mc.featA, 0, nfa * sizeof(uint16_t)));
    const size_t nfu = kFusedSyncHeader + (size_t)((F + 255) / 256) * 8 * kUnitCounterStride * 4;
    HIP_TRY(p->mem.alloc(&p->d_fused, nfu));
    HIP_TRY(hipMemset(p->d_fused, 0, nfu));
    p->fused_bytes = nfu;
    // BODYFIT_ONE_LAUNCH=0 keeps the two-launch sweep (k_frame_resjac, then k_mesh_blend_lbs): A/B measurements, fallback
    const char* fe = std::getenv("BODYFIT_ONE_LAUNCH");
    p->fused_enabled = !(fe && fe[0] == '0');
  }
  *out = p.release();
  return BODYFIT_OK;
}

void bodyfit_problem_destroy(bodyfit_problem* p) {
  if (!p) return;
  (void)hipSetDevice(p->m->device);
  if (p->lm_stream) (void)hipStreamDestroy(p->lm_stream);
  if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
  if (p->async_event) (void)hipEventDestroy(p->async_event);
  // the problem's blocks go to the BlockPool, not to hipFree: what hipFree did implicitly — wait for every kernel that may
  // still touch them — is done here once
  (void)hipDeviceSynchronize();
  delete p;
}

int bodyfit_problem_layout(const bodyfit_problem* p, bodyfit_layout* out) {
  if (!p || !out) return fail(BODYFIT_ERR_INVALID, "null argument");
  *out = p->lay;
  return BODYFIT_OK;
}

int bodyfit_problem_views(bodyfit_problem* p, bodyfit_device_views* out) {
  if (!p || !out) return fail(BODYFIT_ERR_INVALID, "null argument");
  out->residuals = p->d_r;
  out->jacobian = p->d_J;
  out->gmm_comp = p->d_comp;
  out->cloud = p->d_cloud;
  out->cloud_frame_stride = (long long)p->m->d.nVTiles * kVTile * 3;
  out->joints = p->d_joints;
  out->normal_eq = p->d_normal;
  return BODYFIT_OK;
}

int bodyfit_evaluate_device(bodyfit_problem* p, const double* d_frame_params, const double* d_beta,
                            int want_jacobian, void* stream) {
  if (!p || !d_frame_params) return fail(BODYFIT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(p->m->device));
  // Eager launches.  A hipGraph capture of this fork/join sweep was measured SLOWER on MI355X / ROCm 7.2
  // (256 frames: 79.8 us per replay vs 59.8 us eager), so no graph is used here.
  p->async_stream = static_cast<hipStream_t>(stream);
  p->async_pending = true;
  return sweep(p, d_frame_params, d_beta, want_jacobian, p->desc.want_mesh != 0, static_cast<hipStream_t>(stream));
}

// Which column blocks of each reprojection block's Jacobian are STRUCTURALLY non-zero: found once per problem by a probe
// sweep at a generic (pseudo-random) point — a block that is zero there is zero everywhere (the chain walk of
// include/Sim3BA.h:173-207 reaches a keypoint's kinematic ancestors only; landmark / regressor keypoints come out dense).
// No model-specific reasoning on the host: the kernel's own output decides.  Called under the problem's lock.
static int build_pack_tables(bodyfit_problem* p, hipStream_t st) {
  const bodyfit_model* m = p->m;
  const int nJ = m->nJ, nS = m->nS, npose = 7 + 3 * (nJ - 1), n = p->lay.n_cols, K = p->lay.n_keypoints;
  const bool has_beta = n > npose;
  const size_t npar = (size_t)p->n_param_rows * npose;
  const size_t nbeta = has_beta ? (size_t)(p->desc.beta_per_frame ? p->d.F * nS : nS) : 0;
  const size_t nJd = (size_t)p->lay.reproj_rows * n;
  HIP_TRY(p->c_J.ensure(nJd));
  std::vector<double> x(npar), b(nbeta);
  unsigned long long sd = 0x9e3779b97f4a7c15ull;
  auto u = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return (double)(sd >> 11) / 9007199254740992.0 - 0.5; };
  for (int f = 0; f < p->n_param_rows; ++f) {
    double* q = x.data() + (size_t)f * npose;
    q[0] = 1.0 + 0.2 * u();
    for (int i = 1; i < 4; ++i) q[i] = 0.6 * u();
    q[4] = 0.2 * u(); q[5] = 0.2 * u(); q[6] = 3.0 + 0.4 * u();
    for (int i = 7; i < npose; ++i) q[i] = 0.6 * u();
  }
  for (auto& v : b) v = u();
  HIP_TRY(hipMemcpyAsync(p->d_params, x.data(), npar * sizeof(double), hipMemcpyHostToDevice, st));
  if (nbeta) HIP_TRY(hipMemcpyAsync(p->d_beta, b.data(), nbeta * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));   // (x, b are pageable: the copies above have left them)
  if (int rc = sweep(p, p->d_params, has_beta ? p->d_beta : nullptr, 1, false, st)) return rc;
  HIP_TRY(hipMemcpyAsync(p->c_J.data(), p->d_J, nJd * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int nblocks = 3 + (nJ - 1) + (has_beta ? 1 : 0);
  p->pk_mask.assign((size_t)K, 0u);
  p->pk_off.assign((size_t)K + 1, 0u);
  p->pk_src.assign((size_t)K * 32, (short)-1);
  size_t total = 0;
  for (int k = 0; k < K; ++k) {
    const double* J0 = p->c_J.data() + (size_t)(2 * k) * n;
    const double* J1 = J0 + n;
    unsigned mask = 0;
    int ncol = 0;
    for (int blk = 0; blk < nblocks; ++blk) {
      const int off = blk == 0 ? 0 : (blk < 3 + (nJ - 1) ? 1 + 3 * (blk - 1) : npose);
      const int sz = blk == 0 ? 1 : (blk < 3 + (nJ - 1) ? 3 : n - npose);
      bool any = false;
      for (int i = 0; i < sz; ++i) any = any || J0[off + i] != 0.0 || J1[off + i] != 0.0;
      if (any) { mask |= 1u << blk; ncol += sz; }
    }
    p->pk_mask[k] = mask;
    p->pk_off[k] = (unsigned)total;
    {
      // a present block's [2][size] row-major image (what Ceres asks for) starts at src[blk] inside the keypoint's packed span:
      // the span holds the present blocks in block order, each as its two rows back to back (k_pack_jacobian's layout)
      short* src = p->pk_src.data() + (size_t)k * 32;
      int at = 0;
      for (int blk = 0; blk < 32; ++blk) {
        const int sz = blk == 0 ? 1 : (blk < 3 + (nJ - 1) ? 3 : n - npose);
        const bool present = blk < nblocks && ((mask >> blk) & 1u);
        src[blk] = present ? (short)at : (short)-1;
        if (present) at += 2 * sz;
      }
    }
    total += 2 * (size_t)ncol;
  }
  if (total >= ((size_t)1 << 32)) return fail(BODYFIT_ERR_INVALID, "packed Jacobian exceeds 2^32 doubles");
  p->pk_off[K] = (unsigned)total;
  HIP_TRY(p->mem.alloc(&p->d_pk_mask, (size_t)std::max(K, 1)));
  HIP_TRY(p->mem.alloc(&p->d_pk_off, (size_t)K + 1));
  HIP_TRY(p->mem.alloc(&p->d_Jp, std::max<size_t>(total, 1)));
  HIP_TRY(p->c_Jp.ensure(std::max<size_t>(total, 1)));
  HIP_TRY(hipMemcpy(p->d_pk_mask, p->pk_mask.data(), (size_t)K * sizeof(unsigned), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_pk_off, p->pk_off.data(), ((size_t)K + 1) * sizeof(unsigned), hipMemcpyHostToDevice));
  p->pk_ready = true;
  return BODYFIT_OK;
}

int bodyfit_evaluate_batch(bodyfit_problem* p, const double* frame_params, const double* beta, double* residuals,
                           double* jacobian, int* gmm_comp, int want_jacobian) {
  if (!p || !frame_params) return fail(BODYFIT_ERR_INVALID, "null argument");
  const bodyfit_model* m = p->m;
  const int npose = 7 + 3 * (m->nJ - 1);
  const bool has_beta = p->lay.n_cols > npose;
  if (has_beta && !beta) return fail(BODYFIT_ERR_INVALID, "beta required when the shape block is present");
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  const size_t npar = (size_t)p->n_param_rows * npose;
  const size_t nbeta = has_beta ? (size_t)(p->desc.beta_per_frame ? p->d.F * m->nS : m->nS) : 0;
  const int wj = (want_jacobian && p->lay.reproj_rows > 0) ? 1 : 0;
  const size_t nr = (size_t)p->lay.total_rows, nJ = (size_t)p->lay.reproj_rows * p->lay.n_cols;
  // everything crosses PCIe from / into page-locked mirrors, on the problem's own stream
  HIP_TRY(p->c_params.ensure(npar)); HIP_TRY(p->c_beta.ensure(nbeta)); HIP_TRY(p->c_r.ensure(nr));
  HIP_TRY(p->c_comp.ensure((size_t)p->d.F));
  if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
  hipStream_t st = p->copy_stream;
  p->cache_valid = false;
  // No caller's Jacobian buffer: the sweep is cached for bodyfit_evaluate_block (Ceres' EvaluationCallback pattern), and only
  // the structurally non-zero column blocks cross PCIe (BODYFIT_PACKED_J=0: the dense panel, as with a caller's buffer)
  static const bool packed_enabled = [] { const char* e = std::getenv("BODYFIT_PACKED_J"); return !(e && e[0] == '0'); }();
  const bool packed = wj && !jacobian && packed_enabled && p->lay.n_cols <= 128;
  if (int ro = order_after_async(p, st)) return ro;   // behind any asynchronous sweep of this problem still in flight
  if (packed && !p->pk_ready)
    if (int rcp = build_pack_tables(p, st)) return rcp;
  if (wj && !packed) HIP_TRY(p->c_J.ensure(nJ));
  std::memcpy(p->c_params.data(), frame_params, npar * sizeof(double));
  if (nbeta) std::memcpy(p->c_beta.data(), beta, nbeta * sizeof(double));
  p->c_npar = npar; p->c_nbeta = nbeta;
  static const bool pack_direct = [] { const char* e = std::getenv("BODYFIT_PACK_DIRECT"); return !(e && e[0] == '0'); }();
  for (int attempt = 0;; ++attempt) {   // (a one-launch sweep whose in-launch wait ran out is re-issued as two launches)
    // (measured and rejected, round 5: the sweep reading the parameters straight from the page-locked mirrors instead of these two
    //  copies — 180.2 / 181.8 / 182.9 us per cached sweep against 181.1 / 182.6 / 180.0: no difference)
    HIP_TRY(hipMemcpyAsync(p->d_params, p->c_params.data(), npar * sizeof(double), hipMemcpyHostToDevice, st));
    if (nbeta) HIP_TRY(hipMemcpyAsync(p->d_beta, p->c_beta.data(), nbeta * sizeof(double), hipMemcpyHostToDevice, st));
    const double* xs = p->d_params;
    const double* bs = has_beta ? p->d_beta : nullptr;
    int rc = sweep(p, xs, bs, wj, p->desc.want_mesh != 0, st);
    if (rc) return rc;
    const bool one_kernel_down = wj && packed && pack_direct;   // residuals and components ride on the packing kernel
    if (!one_kernel_down) {
      HIP_TRY(hipMemcpyAsync(p->c_r.data(), p->d_r, nr * sizeof(double), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(p->c_comp.data(), p->d_comp, (size_t)p->d.F * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    size_t n_down = 0;
    const double* src_down = nullptr;
    double* dst_down = nullptr;
    // (measured and rejected: the Jacobian's two halves on two streams / copy engines — 234 against 221 us per sweep)
    if (one_kernel_down) {
      // the packing kernel stores straight into the page-locked host cache (it is device-addressable), residuals and GMM
      // components with it: ONE kernel behind the sweep instead of a kernel and three copy commands (219 -> 207 -> see DESIGN 6)
      launch_pack_jacobian(p->lay.n_keypoints, p->lay.n_cols, m->nJ - 1, p->d_J, p->d_pk_mask, p->d_pk_off, p->c_Jp.data(), p->d_r,
                           (int)nr, p->c_r.data(), p->d_comp, p->d.F, p->c_comp.data(), st);
    } else if (wj && packed) {
      launch_pack_jacobian(p->lay.n_keypoints, p->lay.n_cols, m->nJ - 1, p->d_J, p->d_pk_mask, p->d_pk_off, p->d_Jp, nullptr, 0,
                           nullptr, nullptr, 0, nullptr, st);
      n_down = (size_t)p->pk_off[p->lay.n_keypoints]; src_down = p->d_Jp; dst_down = p->c_Jp.data();
    } else if (wj) {
      n_down = nJ; src_down = p->d_J; dst_down = p->c_J.data();
    }
    if (n_down) HIP_TRY(hipMemcpyAsync(dst_down, src_down, n_down * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (attempt == 0 && fused_timed_out(p)) continue;
    break;
  }
  p->cache_valid = true;
  p->cache_has_jac = wj != 0;
  p->cache_packed = packed;
  if (residuals) std::memcpy(residuals, p->c_r.data(), nr * sizeof(double));
  if (jacobian && wj) std::memcpy(jacobian, p->c_J.data(), nJ * sizeof(double));
  if (gmm_comp) std::memcpy(gmm_comp, p->c_comp.data(), (size_t)p->d.F * sizeof(int));
  return BODYFIT_OK;
}

int bodyfit_internal_frame_normals(bodyfit_problem* p, const double* frame_params, const double* beta, double* residuals,
                                   int* gmm_comp, double* H) {
  if (!p || !frame_params || !residuals || !H) return fail(BODYFIT_ERR_INVALID, "null argument");
  const bodyfit_model* m = p->m;
  const int npose = 7 + 3 * (m->nJ - 1), F = p->d.F;
  const bool has_beta = p->lay.n_cols > npose;
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  p->cache_valid = false;
  const size_t npar = (size_t)p->n_param_rows * npose;
  const size_t nbeta = has_beta ? (size_t)(p->desc.beta_per_frame ? F * m->nS : m->nS) : 0;
  const size_t nH = (size_t)F * kNormalRows * kNormalLd;
  if (!p->d_frame_normal) HIP_TRY(p->mem.alloc(&p->d_frame_normal, nH));
  if (int ro = order_after_async(p, nullptr)) return ro;
  HIP_TRY(hipMemcpyAsync(p->d_params, frame_params, npar * sizeof(double), hipMemcpyHostToDevice, nullptr));
  if (nbeta) HIP_TRY(hipMemcpyAsync(p->d_beta, beta, nbeta * sizeof(double), hipMemcpyHostToDevice, nullptr));
  int rc = sweep(p, p->d_params, has_beta ? p->d_beta : nullptr, 1, false, nullptr);
  if (rc) return rc;
  launch_frame_normal(F, p->lay.n_cols, p->d.kp_offset, p->desc.huber_delta, p->d_r, p->d_J, p->d_frame_normal, nullptr);
  HIP_TRY(hipMemcpyAsync(residuals, p->d_r, (size_t)p->lay.total_rows * sizeof(double), hipMemcpyDeviceToHost, nullptr));
  if (gmm_comp) HIP_TRY(hipMemcpyAsync(gmm_comp, p->d_comp, (size_t)F * sizeof(int), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipMemcpyAsync(H, p->d_frame_normal, nH * sizeof(double), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return BODYFIT_OK;
}

int bodyfit_frame_normals(bodyfit_problem* p, const double* frame_params, const double* beta, double* residuals,
                          int* gmm_comp, double* normals) {
  if (p) {
    int maxk = 0;
    for (int f = 0; f < p->d.F; ++f) maxk = std::max(maxk, p->kp_offset[f + 1] - p->kp_offset[f]);
    if (maxk > 32) return fail(BODYFIT_ERR_INVALID, "bodyfit_frame_normals: at most 32 keypoints per frame");
  }
  return bodyfit_internal_frame_normals(p, frame_params, beta, residuals, gmm_comp, normals);
}

int bodyfit_arm_shared_reduction(bodyfit_problem* p, double* d_out66) {
  if (!p) return fail(BODYFIT_ERR_INVALID, "null argument");
  if (d_out66 && (p->desc.beta_per_frame || p->m->nS != kMaxShape || p->lay.n_cols <= 7 + 3 * (p->m->nJ - 1)))
    return fail(BODYFIT_ERR_INVALID, "bodyfit_arm_shared_reduction: needs a problem with a shared 10-coefficient shape block");
  std::lock_guard<std::mutex> lock(p->mu);
  p->armed_out66 = d_out66;
  p->fold_fresh = false;
  return BODYFIT_OK;
}

int bodyfit_reduce_shared_device(bodyfit_problem* p, double* d_out66, void* stream) {
  if (!p) return fail(BODYFIT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(p->m->device));
  const int npose = 7 + 3 * (p->m->nJ - 1);
  const int shared_shape_rows = (!p->desc.beta_per_frame) ? p->lay.shape_rows : 0;
  if (p->fold_fresh && d_out66 && d_out66 == p->armed_out66) return BODYFIT_OK;   // the sweep's own tail has written it
  p->async_stream = static_cast<hipStream_t>(stream);
  p->async_pending = true;
  if (p->d_frame_partials && p->partials_fresh) {
    // the sweep's k_frame_resjac already reduced every frame's reprojection rows: sum the per-frame partials and the
    // prior / temporal rows, pack
    launch_reduce_frames(p->d.F + p->partials_tiles, p->d_r, p->row_shape, shared_shape_rows,
                         p->desc.beta_shape, p->d_frame_partials, p->d_partials, d_out66 ? d_out66 : p->d_normal,
                         static_cast<hipStream_t>(stream));
  } else {
    launch_reduce_shared_ex(p->lay.n_keypoints, p->lay.n_cols, npose, p->m->nS, p->lay.total_rows, p->d_r, p->d_J,
                            p->desc.huber_delta, p->row_shape, shared_shape_rows, p->desc.beta_shape, p->d_partials,
                            d_out66 ? d_out66 : p->d_normal, static_cast<hipStream_t>(stream));
  }
  HIP_TRY(hipGetLastError());
  return BODYFIT_OK;
}

// Per-kernel timing with HIP events on `stream`: `iters` sweeps, avg_ms[0..4] = {frame_resjac, priors,
// mesh_blend_lbs, reduce_shared, sweep_roles} average launch durations in milliseconds.
int bodyfit_profile_sweep(bodyfit_problem* p, const double* d_frame_params, const double* d_beta,
                          int want_jacobian, int with_reduce, int iters, void* stream, double* avg_ms) {
  if (!p || !d_frame_params || !avg_ms || iters <= 0) return fail(BODYFIT_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(p->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // per sweep: [0],[1] begin / end of the k_frame_resjac dispatch, [2],[3] of the mesh dispatch, [4],[5] of the fused
  // dispatch (all taken from the dispatch's own timestamps, so they match rocprofv3's kernel durations; a sweep is
  // either the first two pairs or the third), [6],[7] around the reduction launches
  const bool mesh = p->desc.want_mesh != 0;
  std::vector<hipEvent_t> ev((size_t)iters * 8);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  int rc = BODYFIT_OK;
  const unsigned epoch0 = p->fused_epoch;
  for (int it = 0; it < iters && rc == BODYFIT_OK; ++it) {
    hipEvent_t* e = ev.data() + (size_t)it * 8;
    rc = sweep(p, d_frame_params, d_beta, want_jacobian, mesh, st, e);
    if (rc == BODYFIT_OK && with_reduce) {
      (void)hipEventRecord(e[6], st);
      rc = bodyfit_reduce_shared_device(p, p->armed_out66, stream);   // (armed: the sweep's own tail did it, nothing is launched)
      (void)hipEventRecord(e[7], st);
    }
  }
  const bool fused = p->fused_epoch != epoch0;
  hipError_t se = hipStreamSynchronize(st);
  for (int k = 0; k < 5; ++k) avg_ms[k] = 0.0;
  if (rc == BODYFIT_OK && se == hipSuccess) {
    for (int it = 0; it < iters; ++it) {
      hipEvent_t* e = ev.data() + (size_t)it * 8;
      float ms = 0.f;
      if (fused) {
        (void)hipEventElapsedTime(&ms, e[4], e[5]); avg_ms[4] += ms / iters;
      } else {
        (void)hipEventElapsedTime(&ms, e[0], e[1]); avg_ms[0] += ms / iters;
        if (mesh) { (void)hipEventElapsedTime(&ms, e[2], e[3]); avg_ms[2] += ms / iters; }
      }
      if (with_reduce) { (void)hipEventElapsedTime(&ms, e[6], e[7]); avg_ms[3] += ms / iters; }
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (se != hipSuccess) return fail(BODYFIT_ERR_HIP, std::string("profile sync: ") + hipGetErrorString(se));
  if (rc == BODYFIT_OK) rc = fused_check(p);
  return rc;
}

// Device-resident LM over independent frames (k_lm_batched.hip).  Called by bodyfit_solve.
int bodyfit_internal_solve_batched_device(bodyfit_problem* p, double* frame_params, double* beta,
                                          const unsigned char* param_constant, const bodyfit_fit_options* opt,
                                          bodyfit_fit_summary* summaries, int n_summaries) {
  const bodyfit_model* m = p->m;
  const int F = p->d.F, npose = 7 + 3 * (m->nJ - 1), n = p->lay.n_cols, nb = n - npose;
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  p->cache_valid = false;
  LmState S{};
  LmProblem P{};
  P.F = F; P.ncols = n; P.kp_offset = p->d.kp_offset;
  P.huber = p->desc.huber_delta; P.beta_pose = p->desc.beta_pose; P.beta_shape = p->desc.beta_shape;
  P.scale_lo = opt->scale_lo; P.scale_hi = opt->scale_hi;
  P.prior_rows = p->lay.prior_rows_per_frame; P.row_prior = p->row_prior;
  P.shape_rows_per_frame = (p->lay.shape_rows > 0) ? m->nS : 0; P.row_shape = p->row_shape;
  P.prec = p->has_gmm ? p->gmm.prec : nullptr; P.prec_cho = p->has_gmm ? p->gmm.prec_cho : nullptr;
  P.gmm_mean = p->has_gmm ? p->gmm.mean : nullptr; P.gmm_scale = p->has_gmm ? p->gmm.resid_scale : 0.0;
  double* d_r_new = nullptr;
  double* d_J_new = nullptr;
  int* d_comp_new = nullptr;
  unsigned char* d_const = nullptr;
  {
    // one pooled allocation (sizes depend on the problem only), made on the first solve and reused: seventeen
    // hipMalloc / hipFree pairs per solve were a quarter of a single-frame fit
    const size_t nbb = (size_t)std::max(nb, 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_x = take((size_t)F * npose * 8), o_b = take((size_t)F * nbb * 8), o_xn = take((size_t)F * npose * 8),
                 o_bn = take((size_t)F * nbb * 8), o_rad = take((size_t)F * 8), o_dec = take((size_t)F * 8),
                 o_cost = take((size_t)F * 8), o_ic = take((size_t)F * 8), o_model = take((size_t)F * 8),
                 o_scale = take((size_t)F * 86 * 8), o_flags = take((size_t)F * 4), o_iters = take((size_t)F * 4),
                 o_ok = take((size_t)F * 4), o_bad = take((size_t)F * 4), o_act = take(4),
                 o_rn = take((size_t)std::max(1, p->lay.total_rows) * 8), o_cn = take((size_t)F * 4), o_const = take((size_t)npose),
                 o_Jn = take((size_t)std::max(1, p->lay.reproj_rows) * p->lay.n_cols * 8);
    if (!p->lm_pool) HIP_TRY(p->mem.alloc(&p->lm_pool, off));
    unsigned char* B = p->lm_pool;
    S.x = reinterpret_cast<double*>(B + o_x); S.beta = reinterpret_cast<double*>(B + o_b);
    S.x_new = reinterpret_cast<double*>(B + o_xn); S.beta_new = reinterpret_cast<double*>(B + o_bn);
    S.radius = reinterpret_cast<double*>(B + o_rad); S.dec = reinterpret_cast<double*>(B + o_dec);
    S.cost = reinterpret_cast<double*>(B + o_cost); S.initial_cost = reinterpret_cast<double*>(B + o_ic);
    S.model = reinterpret_cast<double*>(B + o_model); S.scale = reinterpret_cast<double*>(B + o_scale);
    S.flags = reinterpret_cast<int*>(B + o_flags); S.iters = reinterpret_cast<int*>(B + o_iters);
    S.n_ok = reinterpret_cast<int*>(B + o_ok); S.n_bad = reinterpret_cast<int*>(B + o_bad);
    S.active_count = reinterpret_cast<int*>(B + o_act);
    d_r_new = reinterpret_cast<double*>(B + o_rn); d_comp_new = reinterpret_cast<int*>(B + o_cn);
    d_J_new = reinterpret_cast<double*>(B + o_Jn);
    if (param_constant) d_const = B + o_const;
  }
  if (!p->lm_stream) HIP_TRY(hipStreamCreateWithFlags(&p->lm_stream, hipStreamNonBlocking));
  hipStream_t st = p->lm_stream;
  if (int ro = order_after_async(p, st)) return ro;   // (the solve writes the r / J / partials an asynchronous sweep may still be writing)
  HIP_TRY(hipMemsetAsync(S.active_count, 0, sizeof(int), st));
  HIP_TRY(hipMemcpyAsync(S.x, frame_params, (size_t)F * npose * sizeof(double), hipMemcpyHostToDevice, st));
  if (nb) HIP_TRY(hipMemcpyAsync(S.beta, beta, (size_t)F * nb * sizeof(double), hipMemcpyHostToDevice, st));
  if (param_constant) HIP_TRY(hipMemcpyAsync(d_const, param_constant, (size_t)npose, hipMemcpyHostToDevice, st));
  const double* bptr = nb ? S.beta : nullptr;
  int rc = sweep(p, S.x, bptr, 1, false, st);
  if (rc) return rc;
  launch_lm_init(P, S, p->d_r, st);
  int n_sweeps = 1;
  // Speculative iteration (default): the sweep at the candidate also produces its Jacobian (into second buffers), and the
  // next k_lm_step judges the candidate before it builds its system from whichever point won: two launches per iteration
  // (step, sweep) instead of four (step, residual sweep, accept, Jacobian sweep).  A rejected candidate's Jacobian is
  // wasted work that costs no time (the sweep is latency-bound).  BODYFIT_LM_PLAIN=1 keeps the four-launch form.
  const bool plain = [] { const char* e = std::getenv("BODYFIT_LM_PLAIN"); return e && e[0] == '1'; }();
  auto iteration = [&](int first) -> int {
    if (!plain) {
      launch_lm_step(P, S, p->d_r, p->d_J, p->d_comp, d_r_new, d_J_new, d_comp_new, d_const, first, st);
      return sweep(p, S.x_new, nb ? S.beta_new : nullptr, 1, false, st, nullptr, d_r_new, d_comp_new, S.flags, kLmHasCand,
                   nullptr, false, d_J_new);
    }
    launch_lm_step(P, S, p->d_r, p->d_J, p->d_comp, nullptr, nullptr, nullptr, d_const, first, st);
    // candidate residuals only for frames that have a candidate; fresh Jacobians only for frames still active
    int rc2 = sweep(p, S.x_new, nb ? S.beta_new : nullptr, 0, false, st, nullptr, d_r_new, d_comp_new, S.flags, kLmHasCand);
    if (rc2) return rc2;
    launch_lm_accept(P, S, d_r_new, p->d_r, d_comp_new, p->d_comp, st);
    // (prior rows of accepted frames were carried over by k_lm_accept: no prior workgroups on this sweep)
    return sweep(p, S.x, bptr, 1, false, st, nullptr, nullptr, nullptr, S.flags, kLmActive, nullptr, /*skip_priors=*/true);
  };
  // (a hipGraph replay of this ~20-launch iteration was measured slower than eager launches on ROCm 7.2:
  //  256 frames to convergence 25.6 ms vs 23.0 ms; so the loop stays eager)
  for (int it = 0; it < opt->max_iters; ++it) {
    rc = iteration(it == 0 ? 1 : 0);
    if (rc) return rc;
    n_sweeps += plain ? 2 : 1;
    if ((it & 7) == 7 || it + 1 == opt->max_iters) {   // poll the number of frames still iterating (speculative form:
      int active = 0;                                  // as of the previous iteration's candidates)
      HIP_TRY(hipMemcpyAsync(&active, S.active_count, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (active <= 0) break;
    }
  }
  // the last candidates are still unjudged in the speculative form (no further step: only x, cost and the counters matter)
  if (!plain) launch_lm_accept(P, S, d_r_new, p->d_r, d_comp_new, p->d_comp, st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(frame_params, S.x, (size_t)F * npose * sizeof(double), hipMemcpyDeviceToHost));
  if (nb) HIP_TRY(hipMemcpy(beta, S.beta, (size_t)F * nb * sizeof(double), hipMemcpyDeviceToHost));
  if (summaries && n_summaries > 0) {
    std::vector<int> fl(F), itv(F), ok(F), bad(F);
    std::vector<double> c0(F), c1(F);
    HIP_TRY(hipMemcpy(fl.data(), S.flags, F * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(itv.data(), S.iters, F * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ok.data(), S.n_ok, F * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(bad.data(), S.n_bad, F * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c0.data(), S.initial_cost, F * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c1.data(), S.cost, F * sizeof(double), hipMemcpyDeviceToHost));
    for (int f = 0; f < F && f < n_summaries; ++f) {
      bodyfit_fit_summary& s2 = summaries[f];
      s2.iterations = itv[f];
      s2.termination = (fl[f] & kLmActive) ? 1 : ((fl[f] & kLmTermMask) >> kLmTermShift);
      s2.usable = s2.termination != 2;
      s2.n_successful = ok[f]; s2.n_unsuccessful = bad[f];
      s2.n_sweeps = n_sweeps; s2.n_sweeps_issued = n_sweeps;
      s2.initial_cost = c0[f]; s2.final_cost = c1[f];
    }
  }
  return BODYFIT_OK;
}

// Cyclic-reduction schedule over `n` chain nodes (ids base .. base + n - 1): per level the eliminated nodes (j, left, right)
// and the remaining ones that receive an update (a, jl, jr, next).  pinned: the two end nodes are never eliminated (a shard's
// interface with its neighbours); otherwise the last level is the root (j, -1, -1).
struct CrLevel { int elim_off, n_elim, surv_off, n_surv; };
static void build_cr_schedule(int n, bool pinned, std::vector<int>& sched, std::vector<CrLevel>& levels) {
  std::vector<int> active(n);
  for (int f = 0; f < n; ++f) active[f] = f;
  for (;;) {
    const int na = (int)active.size();
    std::vector<char> el(na, 0);
    int ne = 0;
    // every other node goes; with an odd number of free-ended nodes the EVEN positions (one more of them) go, so that
    // 20 frames take 20 -> 10 -> 5 -> 2 -> 1 -> root, one level less than always eliminating the odd positions
    const int first = (!pinned && (na & 1) && na > 1) ? 0 : 1;
    for (int pos = first; pos < na; pos += 2)
      if (!(pinned && pos == na - 1)) { el[pos] = 1; ++ne; }
    if (ne == 0) break;
    CrLevel lv{};
    lv.elim_off = (int)sched.size();
    for (int pos = 0; pos < na; ++pos)
      if (el[pos]) {
        sched.push_back(active[pos]); sched.push_back(pos > 0 ? active[pos - 1] : -1); sched.push_back(pos + 1 < na ? active[pos + 1] : -1);
        ++lv.n_elim;
      }
    lv.surv_off = (int)sched.size();
    std::vector<int> next;
    for (int pos = 0; pos < na; ++pos) {
      if (el[pos]) continue;
      next.push_back(active[pos]);
      const int jl = (pos > 0 && el[pos - 1]) ? active[pos - 1] : -1;
      const int jr = (pos + 1 < na && el[pos + 1]) ? active[pos + 1] : -1;
      if (jl < 0 && jr < 0) continue;
      sched.push_back(active[pos]); sched.push_back(jl); sched.push_back(jr);
      sched.push_back((jr >= 0 && pos + 2 < na) ? active[pos + 2] : -1);
      ++lv.n_surv;
    }
    levels.push_back(lv);
    active.swap(next);
  }
  if (!pinned) {
    CrLevel root{};
    root.elim_off = (int)sched.size(); root.n_elim = 1;
    sched.push_back(active[0]); sched.push_back(-1); sched.push_back(-1);
    levels.push_back(root);
  }
}

// carve the cyclic-reduction buffers of `n` nodes out of a pool
static size_t carve_cr(unsigned char* base, size_t off, int n, WinBuf& W, bool dry) {
  const size_t blk = (size_t)kWinBlock * kWinBlock * 8, rhs = (size_t)kWinRhs * kWinBlock * 8;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_D = take(n * blk), o_U = take(n * blk), o_L = take(n * blk), o_P = take(n * blk), o_Q = take(n * blk),
               o_R = take(n * rhs), o_R0 = take(n * rhs), o_Y = take(n * rhs), o_X = take(n * rhs),
               o_Li = take((size_t)n * (kWinBlock / 16) * 256 * 8), o_fail = take(8), o_ticket = take(4 * (size_t)(2 + n / 32 + 1));
  if (!dry) {
    auto dp = [&](size_t o) { return reinterpret_cast<double*>(base + o); };
    W.D = dp(o_D); W.U = dp(o_U); W.L = dp(o_L); W.Pt = dp(o_P); W.Qt = dp(o_Q); W.Rt = dp(o_R); W.Rt0 = dp(o_R0);
    W.Yt = dp(o_Y); W.Xt = dp(o_X); W.Li = dp(o_Li); W.fail = reinterpret_cast<int*>(base + o_fail);
    W.ticket = reinterpret_cast<int*>(base + o_ticket);
  }
  return off;
}

// Device-resident LM for ONE problem over all frames with a shared beta (k_window_lm.hip): the outer loop of
// OptimizeMultiFrame (include/MultiFrameBA.h:144-151) with every piece of linear algebra on the device.  Per LM iteration
// the host launches: [Jacobian sweep + k_frame_normal when the point moved] -> assemble -> cyclic reduction up and down ->
// beta Schur + step + model change -> residual sweep at the candidate -> accept, and reads back one status record.
//
// comm != NULL: this problem is ONE SHARD (contiguous frames) of the window, one process per GPU (bodyfit_solve_sharded).
// Every rank reduces its own chain down to its two end frames (cyclic reduction with the ends pinned), the 2 N interface
// blocks are all-gathered and solved redundantly by every rank, then each rank substitutes back through its own levels.
// What crosses ranks per LM iteration, as THREE all-gathers of device buffers ordered on the solve's stream (RCCL: nothing
// touches the host, no stream synchronisation between the host's status reads):
//   1. the interface blocks of the shard's two end frames (225 KB) with the shard's beta terms [C, g_beta] (110 doubles)
//      riding on the same buffer;
//   2. the shard's beta Schur partials (110 doubles) — they need the interface solution, the step needs them;
//   3. the shard's scalars [model change, |d|^2, |x|^2, max |g|, failure flag, cost at the candidate] (8 doubles): ONE
//      decision kernel then applies Ceres' tests and the accept / reject rules on every rank, on the same numbers.
// Sums are taken by every rank in rank order from the gathered partials (bit-identical totals, identical decisions, no
// broadcast).  The steps of the neighbouring shards' boundary frames — the halo row of the temporal pair this shard owns, the
// frame in front of its first — are not exchanged at all: every rank holds the whole interface solution and computes them
// with the neighbour's own arithmetic (k_win_halo_step).  The first iteration has two more small gathers (the beta terms
// before the first scaling, the boundary frames' Jacobi scaling), the start one (the boundary rows).
static int solve_window_device(bodyfit_problem* p, double* frame_params, double* beta,
                               const unsigned char* param_constant, const bodyfit_fit_options* opt,
                               bodyfit_fit_summary* summary, Transport* comm, bool force_sharded) {
  const bodyfit_model* m = p->m;
  const int F = p->d.F, npose = 7 + 3 * (m->nJ - 1), n = p->lay.n_cols, nb = n - npose;
  // shard proxy (bodyfit_set_shard_proxy): through a ONE-rank communicator this problem runs as rank proxy_rank of proxy_ranks
  // identical shards — every kernel, buffer and exchange of that geometry, the gathered slots filled with copies of its own
  const bool proxy = comm != nullptr && comm->size == 1 && p->proxy_ranks > 1;
  const bool sharded = comm != nullptr && (comm->size > 1 || force_sharded || proxy);
  const int halo = p->desc.temporal_halo ? 1 : 0;
  if (npose != kFrameParams || nb != kMaxShape || p->desc.beta_per_frame || p->has_gmm)
    return fail(BODYFIT_ERR_INVALID, "device window solver: needs 24 joints, a shared 10-coefficient beta and the L2 pose prior");
  if (halo && !sharded) return fail(BODYFIT_ERR_INVALID, "device window solver: a halo row needs bodyfit_solve_sharded");
  if (sharded && F < 2) return fail(BODYFIT_ERR_INVALID, "bodyfit_solve_sharded: every shard needs at least two frames");
  const int R = proxy ? p->proxy_rank : (sharded ? comm->rank : 0), N = proxy ? p->proxy_ranks : (sharded ? comm->size : 1);
  const bool has_left = sharded && R > 0;
  if (sharded && (halo != 0) != (R + 1 < N))
    return fail(BODYFIT_ERR_INVALID, "bodyfit_solve_sharded: every shard but the last needs temporal_halo");
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  p->cache_valid = false;
  // ---- schedules ----
  std::vector<int> sched;
  std::vector<CrLevel> levels, ilevels;
  build_cr_schedule(F, sharded, sched, levels);
  const int NI = 2 * N;   // interface nodes
  if (sharded) build_cr_schedule(NI, false, sched, ilevels);
  // ---- one pooled allocation, kept across solves ----
  WinBuf W{}, Wi{};
  double *d_x, *d_b, *d_xn, *d_bn, *d_rn, *d_xl, *d_sh, *d_dh, *d_xln, *d_sl, *d_cg, *d_send, *d_gath, *d_Jn;
  int *d_compn, *d_sched;
  unsigned char* d_const = nullptr;
  {
    size_t off = carve_cr(nullptr, 0, F, W, true);
    if (sharded) off = carve_cr(nullptr, off, NI, Wi, true);
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_A = take((size_t)F * npose * npose * 8), o_B = take((size_t)F * npose * nb * 8),
                 o_g = take((size_t)F * npose * 8), o_E = take((size_t)F * npose * 8), o_sc = take(((size_t)F * npose + nb) * 8),
                 o_Cs = take(100 * 8), o_rb = take(16 * 8), o_Cr = take(112 * 8), o_dsb = take(16 * 8),
                 o_sred = take(112 * 8), o_fin = take(8 * 8),
                 o_part = take((size_t)F * kWinPart * 8), o_gm = take((size_t)(F + 1) * 8), o_d = take(((size_t)F * npose + nb) * 8),
                 o_st = take(kWsCount * 8), o_x = take((size_t)(F + 1) * npose * 8), o_b = take(nb * 8),
                 o_xn = take((size_t)(F + 1) * npose * 8), o_bn = take(nb * 8), o_rn = take((size_t)std::max(1, p->lay.total_rows) * 8),
                 o_cn = take((size_t)F * 4), o_sched = take(sched.size() * 4), o_const = take((size_t)npose),
                 o_xl = take(npose * 8), o_sh = take(npose * 8), o_dh = take(npose * 8), o_xln = take(npose * 8), o_sl = take(npose * 8),
                 o_Jn = take((size_t)std::max(1, p->lay.reproj_rows) * p->lay.n_cols * 8),
                 o_cg = take(112 * 8), o_send = take(sharded ? (size_t)iface_doubles(112) * 8 : 8),
                 o_gath = take(sharded ? (size_t)N * iface_doubles(112) * 8 : 8);
    if (!p->win_pool || p->win_pool_bytes < off) {
      HIP_TRY(p->mem.alloc(&p->win_pool, off));
      HIP_TRY(hipMemset(p->win_pool, 0, off));
      p->win_pool_bytes = off;
    }
    unsigned char* Bp = p->win_pool;
    size_t o2 = carve_cr(Bp, 0, F, W, false);
    if (sharded) carve_cr(Bp, o2, NI, Wi, false);
    auto dp = [&](size_t o) { return reinterpret_cast<double*>(Bp + o); };
    W.Araw = dp(o_A); W.Braw = dp(o_B); W.graw = dp(o_g); W.Eraw = dp(o_E);
    W.scale = dp(o_sc); W.Cs = dp(o_Cs); W.rhsb = dp(o_rb); W.Craw = dp(o_Cr); W.gbraw = dp(o_Cr) + 100; W.dsb = dp(o_dsb);
    W.sred = dp(o_sred); W.fin = dp(o_fin);
    W.part = dp(o_part); W.gmaxp = dp(o_gm); W.d = dp(o_d); W.status = dp(o_st);
    d_x = dp(o_x); d_b = dp(o_b); d_xn = dp(o_xn); d_bn = dp(o_bn); d_rn = dp(o_rn);
    d_compn = reinterpret_cast<int*>(Bp + o_cn);
    d_sched = reinterpret_cast<int*>(Bp + o_sched);
    if (param_constant) d_const = Bp + o_const;
    d_xl = dp(o_xl); d_sh = dp(o_sh); d_dh = dp(o_dh); d_xln = dp(o_xln); d_sl = dp(o_sl);
    d_cg = dp(o_cg); d_send = dp(o_send); d_gath = dp(o_gath);
    d_Jn = dp(o_Jn);
  }
  if (!p->d_frame_normal) HIP_TRY(p->mem.alloc(&p->d_frame_normal, (size_t)F * kNormalRows * kNormalLd));
  if (!p->lm_stream) HIP_TRY(hipStreamCreateWithFlags(&p->lm_stream, hipStreamNonBlocking));
  hipStream_t st = p->lm_stream;
  if (int ro = order_after_async(p, st)) return ro;
  WinProblem P{};
  P.F = F; P.K = p->lay.n_keypoints; P.total_rows = p->lay.total_rows; P.nb = nb; P.halo = halo;
  P.prior_rows = p->lay.prior_rows_per_frame; P.row_prior = p->row_prior;
  P.shape_rows = p->lay.shape_rows; P.row_shape = p->row_shape; P.row_temporal = p->row_temporal;
  P.huber = p->desc.huber_delta; P.beta_pose = p->desc.beta_pose; P.beta_shape = p->desc.beta_shape;
  P.lambda_t = p->desc.lambda_temporal; P.scale_lo = opt->scale_lo; P.scale_hi = opt->scale_hi;
  {
    // every sweep of this loop is a Jacobian sweep without frame flags: with a shared shape block (the partials exist) it leaves
    // the point's cost as F + tiles partial sums (sweep(): dp.beta_partials, pa.plain_cost)
    const bool priors = p->desc.beta_pose > 0.0 || (p->lay.shape_rows > 0 && p->desc.beta_shape > 0.0) || p->desc.lambda_temporal > 0.0;
    P.cost_partials = (p->d_frame_partials && n > npose && m->nS == kMaxShape) ? p->d_frame_partials : nullptr;
    P.cost_tiles = priors ? (F + 15) / 16 : 0;
  }
  const int rows_x = F + halo;
  HIP_TRY(hipMemcpyAsync(d_x, frame_params, (size_t)rows_x * npose * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_b, beta, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_sched, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice, st));
  if (param_constant) HIP_TRY(hipMemcpyAsync(d_const, param_constant, (size_t)npose, hipMemcpyHostToDevice, st));
  // ---- exchanges of a sharded solve: all-gathers of device buffers on the solve's stream (collectives.h) ----
  auto comm_fail = [&](const char* what) {
    return fail(BODYFIT_ERR_INVALID, std::string("bodyfit_solve_sharded: ") + what + " failed: " + (comm ? comm->error : ""));
  };
  // gather n doubles per rank from d_send into d_gath [N][n]
  auto gather = [&](const double* d_send, int cnt, const char* what) -> int {
    if (comm->allgather(d_send, d_gath, cnt, st)) return comm_fail(what);
    if (proxy) launch_replicate_ranks(d_gath, cnt, N, st);
    return BODYFIT_OK;
  };
  // the host's wait for a status record: bounded for sharded solves with bodyfit_set_exchange_timeout (a peer that left after a
  // transport failure never enters the collectives queued on the stream)
  auto wait_status = [&](hipStream_t s2) -> int {
    hipError_t he = hipSuccess;
    const int w = wait_stream(s2, sharded ? p->exchange_timeout_s : 0.0, &he);
    if (w == 1) return fail(BODYFIT_ERR_HIP, "bodyfit_solve_sharded: the solve's stream did not drain within the exchange timeout "
                                             "(a peer has left the collective); the problem's stream is unusable from here on");
    if (w < 0) return fail(BODYFIT_ERR_HIP, std::string("status read: ") + hipGetErrorString(he));
    return BODYFIT_OK;
  };
  if (sharded) {
    // the boundary rows of the starting point: [first row | last row] of every shard -> the frame in front of this shard's
    // first (d_xl) and the halo row behind its last
    HIP_TRY(hipMemcpyAsync(d_send, d_x, npose * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_send + npose, d_x + (size_t)(F - 1) * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (int rcg = gather(d_send, 2 * npose, "allgather (boundary rows)")) return rcg;
    if (has_left)
      HIP_TRY(hipMemcpyAsync(d_xl, d_gath + ((size_t)(R - 1) * 2 + 1) * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (halo)
      HIP_TRY(hipMemcpyAsync(d_x + (size_t)F * npose, d_gath + (size_t)(R + 1) * 2 * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  // The loop never sweeps twice at the same point.  The sweep at a candidate also leaves the candidate's Jacobian (second
  // buffers); the next iteration's k_frame_normal takes the starting point's (r, J), the accepted candidate's, or nothing at
  // all (rejected step: the panels are current), as the device's own record says (W.status[kWsJsel]; sharded solves: every
  // rank's k_win_decide writes the same).
  // Sharded solves: a rank whose own work fails (a kernel launch, a HIP call) must not simply return — its peers would wait for
  // it in the next all-gather for ever.  It marks slot 6 of its scalars (`poison`), keeps taking part in the exchanges of the
  // iteration, and k_win_decide ends the solve on EVERY rank in that same iteration (kWsPoison).  Only a failure of the
  // transport itself returns at once (bodyfit_set_exchange_timeout bounds how long the peers then wait).
  int poison = BODYFIT_OK;
  std::string poison_msg;
  int rc = sweep(p, d_x, d_b, 1, false, st);
  if (rc && sharded) { poison = rc; poison_msg = g_err; rc = BODYFIT_OK; }   // (between two exchanges: stay in step)
  if (rc) return rc;
  if (!sharded) {
    launch_win_init(P, W, p->d_r, 0, st);
  } else {
    launch_win_init(P, W, p->d_r, 1, st);
    if ((rc = gather(W.fin, 1, "allgather (initial cost)"))) return rc;
    launch_sum_ranks(d_gath, N, 1, 1, W.fin, st);
    launch_win_init(P, W, p->d_r, 2, st);
  }
  int n_sweeps = 1;
  double status[kWsCount] = {0};
  bool first = true;
  const size_t rhs = (size_t)kWinRhs * kWinBlock;
  if (sharded) HIP_TRY(hipMemsetAsync(W.fin, 0, 8 * sizeof(double), st));
  // test hook (tests/test_gpu_sharded_solve.py, bodyfit_internal_set_test_poison): that rank's sweep "fails" in that iteration
  const int test_poison_rank = p->test_poison_rank, test_poison_iter = p->test_poison_iter;
  // a HIP call inside the loop: unsharded, its failure returns; sharded, it poisons (this rank stays in the exchanges)
  auto guard = [&](hipError_t e, const char* what) -> int {
    if (e == hipSuccess) return BODYFIT_OK;
    const int rcg = fail(BODYFIT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    if (!sharded) return rcg;
    if (!poison) { poison = rcg; poison_msg = g_err; }
    return BODYFIT_OK;
  };
  for (int it = 0; it < opt->max_iters; ++it) {
    launch_frame_normal_sel(F, n, p->d.kp_offset, p->desc.huber_delta, p->d_r, p->d_J, d_rn, d_Jn, W.status + kWsJsel,
                            p->lay.total_rows, p->d_frame_normal, st);
    // ---- beta block and per-frame blocks ----
    if (!sharded) {
      launch_win_beta(P, W, p->d_frame_normal, p->d_r, first ? 1 : 0, 0, st);
      launch_win_assemble(P, W, p->d_frame_normal, p->d_r, d_x, d_const, first ? 1 : 0, nullptr, nullptr, st);
    } else {
      launch_win_beta(P, W, p->d_frame_normal, p->d_r, first ? 1 : 0, 1, st);   // this shard's [C (100) | g_beta (10)] -> W.Craw
      if (first) {
        // first iteration only: the Jacobi scaling needs the complete C before the blocks are assembled, and the boundary
        // frames' scaling rows (this shard's first and last frame) complete the couplings across the shard boundaries
        if ((rc = gather(W.Craw, 112, "allgather (beta terms)"))) return rc;
        launch_sum_ranks(d_gath, N, 112, 110, W.Craw, st);
        launch_win_beta(P, W, p->d_frame_normal, p->d_r, 1, 2, st);
        launch_win_assemble(P, W, p->d_frame_normal, p->d_r, d_x, d_const, 1, has_left ? d_xl : nullptr, nullptr, st);
        if (int g = guard(hipMemcpyAsync(d_send, W.scale, npose * sizeof(double), hipMemcpyDeviceToDevice, st), "scaling rows")) return g;
        if (int g = guard(hipMemcpyAsync(d_send + npose, W.scale + (size_t)(F - 1) * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st), "scaling rows")) return g;
        if ((rc = gather(d_send, 2 * npose, "allgather (scaling rows)"))) return rc;
        if (halo) if (int g = guard(hipMemcpyAsync(d_sh, d_gath + (size_t)(R + 1) * 2 * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st), "scaling halo")) return g;
        if (has_left) if (int g = guard(hipMemcpyAsync(d_sl, d_gath + ((size_t)(R - 1) * 2 + 1) * npose, npose * sizeof(double), hipMemcpyDeviceToDevice, st), "scaling halo")) return g;
      }
      launch_win_assemble(P, W, p->d_frame_normal, p->d_r, d_x, d_const, 0, has_left ? d_xl : nullptr, halo ? d_sh : nullptr, st);
    }
    // ---- cyclic reduction over the local chain ----
    for (size_t l = 0; l < levels.size(); ++l) {
      const CrLevel& lv = levels[l];
      launch_cr_factor(W, d_sched + lv.elim_off, lv.n_elim, st);
      launch_cr_update(W, d_sched + lv.surv_off, lv.n_surv, st);
    }
    if (sharded) {
      // ---- interface system of the 2 N end frames: ONE all-gather (the shard's beta terms ride on it from the second
      //      iteration on), then every rank solves the same chain ----
      const int n_extra = 112;
      launch_iface_pack(W, F, W.Craw, n_extra, d_send, st);
      if ((rc = gather(d_send, iface_doubles(n_extra), "allgather (interface blocks)"))) return rc;
      launch_iface_unpack(Wi, d_gath, N, n_extra, d_cg, st);
      if (!first) {
        if (int g = guard(hipMemcpyAsync(W.Craw, d_cg, 110 * sizeof(double), hipMemcpyDeviceToDevice, st), "beta terms")) return g;
        launch_win_beta(P, W, p->d_frame_normal, p->d_r, 0, 2, st);
      }
      for (size_t l = 0; l < ilevels.size(); ++l) {
        const CrLevel& lv = ilevels[l];
        launch_cr_factor(Wi, d_sched + lv.elim_off, lv.n_elim, st);
        launch_cr_update(Wi, d_sched + lv.surv_off, lv.n_surv, st);
      }
      for (size_t l = ilevels.size(); l-- > 0;) launch_cr_back(Wi, d_sched + ilevels[l].elim_off, ilevels[l].n_elim, st);
      if (int g = guard(hipMemcpyAsync(W.Xt, Wi.Xt + (size_t)(2 * R) * rhs, rhs * 8, hipMemcpyDeviceToDevice, st), "interface solution")) return g;
      if (int g = guard(hipMemcpyAsync(W.Xt + (size_t)(F - 1) * rhs, Wi.Xt + (size_t)(2 * R + 1) * rhs, rhs * 8, hipMemcpyDeviceToDevice, st), "interface solution")) return g;
    }
    for (size_t l = levels.size(); l-- > 0;) launch_cr_back(W, d_sched + levels[l].elim_off, levels[l].n_elim, st);
    // ---- beta Schur complement, step, model change, decision ----
    launch_win_schur_part(P, W, st);
    if (!sharded) {
      launch_win_beta_solve(P, W, d_b, d_bn, 0, st);
      if (F <= 256) {
        launch_win_tail(P, W, d_x, d_b, d_xn, d_bn, st);      // step + model change + (last workgroup) decision in one launch
      } else {                                                // (long windows: the three kernels are bandwidth-bound, not
        launch_win_step(P, W, d_x, d_xn, st);                 //  launch-bound, and their separate grids fill the chip better)
        launch_win_model(P, W, d_x, nullptr, st);
        launch_win_finish(P, W, d_x, d_b, d_xn, d_bn, 0, st);
      }
    } else {
      // (a failed interface factorisation is everybody's failure: every rank factors the same chain and sees the same flag,
      //  k_win_finish folds it into the shard's own)
      launch_win_beta_solve(P, W, d_b, d_bn, 1, st);                     // this shard's Schur partials -> W.sred
      if ((rc = gather(W.sred, 112, "allgather (Schur partials)"))) return rc;
      launch_sum_ranks(d_gath, N, 112, 110, W.sred, st);
      launch_win_beta_solve(P, W, d_b, d_bn, 2, st);
      launch_win_step(P, W, d_x, d_xn, st);
      // the neighbours' boundary frames move by the steps their own shards compute (same arithmetic, same numbers)
      launch_win_halo_step(P, Wi.Xt, W.dsb, halo ? 2 * (R + 1) : -1, d_sh, d_x + (size_t)F * npose, d_dh, d_xn + (size_t)F * npose,
                           has_left ? 2 * R - 1 : -1, d_sl, d_xl, d_xln, st);
      launch_win_model(P, W, d_x, halo ? d_dh : nullptr, st);
      launch_win_fold_fail(W, Wi, st);
      launch_win_finish(P, W, d_x, d_b, d_xn, d_bn, 1, st);              // this shard's scalars -> W.fin[0..4]
    }
    if (!sharded) {
      rc = sweep(p, d_xn, d_bn, 1, false, st, nullptr, d_rn, d_compn, nullptr, 0, nullptr, false, d_Jn);
      if (rc) return rc;
      ++n_sweeps;
      launch_win_accept(P, W, d_rn, d_x, d_b, d_xn, d_bn, 0, st);
    } else {
      // the candidate is evaluated whatever the decision will be (it is taken once, below, from everybody's scalars)
      rc = sweep(p, d_xn, d_bn, 1, false, st, nullptr, d_rn, d_compn, nullptr, 0, nullptr, false, d_Jn);
      if (R == test_poison_rank && it == test_poison_iter) rc = fail(BODYFIT_ERR_HIP, "test hook: this rank's sweep failed");
      if (rc && !poison) { poison = rc; poison_msg = g_err; }
      ++n_sweeps;
      launch_win_accept(P, W, d_rn, d_x, d_b, d_xn, d_bn, 3, st);       // this shard's cost at the candidate -> W.fin[5]
      if (poison) {
        static const double one = 1.0;
        (void)hipMemcpyAsync(W.fin + 6, &one, sizeof(double), hipMemcpyHostToDevice, st);
      }
      if ((rc = gather(W.fin, 8, "allgather (scalars)"))) return rc;
      launch_win_decide(P, W, d_x, d_b, d_xn, d_bn, d_gath, N, halo ? d_x + (size_t)F * npose : nullptr, d_xn + (size_t)F * npose,
                        has_left ? d_xl : nullptr, d_xln, st);
    }
    first = false;
    if (!opt->verbose && (it & 3) != 3 && it + 1 < opt->max_iters) {
      // The device takes every decision itself, so the host only looks at the status record every fourth iteration (a
      // read-back drains the launch pipeline: ~30 us of a ~250 us iteration at 20 frames).  Iterations launched after the
      // solve has terminated leave the state untouched (every kernel checks the active / candidate flags).
      continue;
    }
    HIP_TRY(hipMemcpyAsync(status, W.status, sizeof(status), hipMemcpyDeviceToHost, st));
    if (int rw = wait_status(st)) return rw;
    if (opt->verbose && R == 0)
      std::printf("[bodyfit-dev] it %3d cost %.6e radius %.3e accepted %d gmax %.2e\n", (int)status[kWsIters], status[kWsCost],
                  status[kWsRadius], (int)status[kWsAccepted], status[kWsGmax]);
    if (status[kWsActive] == 0.0) break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(status, W.status, sizeof(status), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(frame_params, d_x, (size_t)rows_x * npose * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(beta, d_b, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, st));
  if (int rw = wait_status(st)) return rw;
  if (sharded && (poison || status[kWsPoison] != 0.0)) {
    if (summary) {
      *summary = bodyfit_fit_summary{};
      summary->iterations = (int)status[kWsIters]; summary->termination = 2; summary->usable = 0;
      summary->n_successful = (int)status[kWsOk]; summary->n_unsuccessful = (int)status[kWsBad];
      summary->n_sweeps = 1 + (int)status[kWsIters]; summary->n_sweeps_issued = n_sweeps;
      summary->initial_cost = status[kWsInitialCost]; summary->final_cost = status[kWsCost];
    }
    if (poison) return fail(poison, "sharded solve: this rank failed (" + poison_msg + "); every rank left at the same exchange");
    return fail(BODYFIT_ERR_HIP, "sharded solve: another rank reported a device failure; every rank left at the same exchange");
  }
  if (summary) {
    summary->iterations = (int)status[kWsIters];
    summary->termination = status[kWsActive] != 0.0 ? 1 : (int)status[kWsTermination];
    summary->usable = summary->termination != 2;
    summary->n_successful = (int)status[kWsOk]; summary->n_unsuccessful = (int)status[kWsBad];
    // the evaluations the solve NEEDED, from the device's own record: one at the start and one per iteration at the candidate —
    // that sweep also leaves the candidate's Jacobian (second buffers), so an accepted step costs no sweep of its own.  The loop
    // ISSUES a few more: between two status reads it runs up to three iterations past the termination (those kernels find the
    // solve inactive and leave the state alone); n_sweeps_issued is the host's own count.
    summary->n_sweeps = 1 + (int)status[kWsIters];
    summary->n_sweeps_issued = n_sweeps;
    summary->initial_cost = status[kWsInitialCost]; summary->final_cost = status[kWsCost];
  }
  return BODYFIT_OK;
}

// the unsharded entry (host_solver.cpp's router)
int bodyfit_internal_solve_window_device(bodyfit_problem* p, double* frame_params, double* beta,
                                         const unsigned char* param_constant, const bodyfit_fit_options* opt,
                                         bodyfit_fit_summary* summary, const bodyfit_comm* comm) {
  (void)comm;
  return solve_window_device(p, frame_params, beta, param_constant, opt, summary, nullptr, false);
}

static int sharded_common(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                          Transport* tr, const bodyfit_fit_options* opt_in, bodyfit_fit_summary* summary, long* n_exchanges) {
  bodyfit_fit_options opt;
  opt.max_iters = 100; opt.scale_lo = -1e300; opt.scale_hi = 1e300; opt.verbose = 0; opt.solver = 3;
  if (opt_in) opt = *opt_in;
  int maxk = 0;
  for (int f = 0; f < p->d.F; ++f) maxk = std::max(maxk, p->kp_offset[f + 1] - p->kp_offset[f]);
  if (maxk > 32) return fail(BODYFIT_ERR_INVALID, "bodyfit_solve_sharded: at most 32 keypoints per frame");
  // BODYFIT_FORCE_SHARDED=1 (tests): a communicator of ONE rank still takes the sharded code path (interface system of its two
  // end frames, every exchange issued), which is how the RCCL transport is exercised on a box with a single GPU
  const char* fs = std::getenv("BODYFIT_FORCE_SHARDED");
  const long before = tr->n_calls;
  const int rc = solve_window_device(p, frame_params, beta, param_constant, &opt, summary, tr, fs && fs[0] == '1');
  if (n_exchanges) *n_exchanges = tr->n_calls - before;
  return rc;
}

// One window sharded over several processes (one per GPU): this rank's shard of the frames, exchanges through the caller's
// callbacks on host buffers (the transport of tests and of MPI hosts; bodyfit_solve_sharded_rccl keeps them on the device).
int bodyfit_solve_sharded(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                          const bodyfit_comm* comm, const bodyfit_fit_options* opt_in, bodyfit_fit_summary* summary) {
  if (!p || !frame_params || !beta || !comm || !comm->allgather || comm->size < 1 || comm->rank < 0 || comm->rank >= comm->size)
    return fail(BODYFIT_ERR_INVALID, "bodyfit_solve_sharded: bad argument");
  HostTransport tr;
  tr.rank = comm->rank; tr.size = comm->size; tr.cb = *comm; tr.timeout_s = p->exchange_timeout_s;
  long n = 0;
  const int rc = sharded_common(p, frame_params, beta, param_constant, &tr, opt_in, summary, &n);
  p->last_exchanges = n;
  return rc;
}

// ---- RCCL transport ---------------------------------------------------------------------------------------------------------
struct bodyfit_rccl {
  RcclTransport tr;
  bool owns = false;
};

int bodyfit_rccl_unique_id(unsigned char* id128) {
  if (!id128) return fail(BODYFIT_ERR_INVALID, "null argument");
  RcclApi& A = RcclApi::get();
  if (!A.ok()) return fail(BODYFIT_ERR_HIP, A.error);
  RcclApi::unique_id id;
  const int rc = A.GetUniqueId(&id);
  if (rc != 0) return fail(BODYFIT_ERR_HIP, std::string("ncclGetUniqueId: ") + A.GetErrorString(rc));
  std::memcpy(id128, id.internal, 128);
  return BODYFIT_OK;
}

int bodyfit_rccl_create(const unsigned char* id128, int rank, int size, int device, bodyfit_rccl** out) {
  if (!id128 || !out || size < 1 || rank < 0 || rank >= size) return fail(BODYFIT_ERR_INVALID, "bodyfit_rccl_create: bad argument");
  *out = nullptr;
  RcclApi& A = RcclApi::get();
  if (!A.ok()) return fail(BODYFIT_ERR_HIP, A.error);
  HIP_TRY(hipSetDevice(device));
  RcclApi::unique_id id;
  std::memcpy(id.internal, id128, 128);
  std::unique_ptr<bodyfit_rccl> c(new bodyfit_rccl());
  const int rc = A.CommInitRank(&c->tr.comm, size, id, rank);
  if (rc != 0) return fail(BODYFIT_ERR_HIP, std::string("ncclCommInitRank: ") + A.GetErrorString(rc));
  c->tr.rank = rank; c->tr.size = size; c->owns = true;
  *out = c.release();
  return BODYFIT_OK;
}

int bodyfit_rccl_wrap(void* nccl_comm, int rank, int size, bodyfit_rccl** out) {
  if (!nccl_comm || !out || size < 1 || rank < 0 || rank >= size) return fail(BODYFIT_ERR_INVALID, "bodyfit_rccl_wrap: bad argument");
  RcclApi& A = RcclApi::get();
  if (!A.ok()) return fail(BODYFIT_ERR_HIP, A.error);
  bodyfit_rccl* c = new bodyfit_rccl();
  c->tr.comm = nccl_comm; c->tr.rank = rank; c->tr.size = size; c->owns = false;
  *out = c;
  return BODYFIT_OK;
}

void bodyfit_rccl_destroy(bodyfit_rccl* c) {
  if (!c) return;
  if (c->owns && c->tr.comm) (void)RcclApi::get().CommDestroy(c->tr.comm);
  delete c;
}

int bodyfit_solve_sharded_rccl(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                               bodyfit_rccl* comm, const bodyfit_fit_options* opt_in, bodyfit_fit_summary* summary) {
  if (!p || !frame_params || !beta || !comm || !comm->tr.comm) return fail(BODYFIT_ERR_INVALID, "bodyfit_solve_sharded_rccl: bad argument");
  long n = 0;
  comm->tr.timeout_s = p->exchange_timeout_s;
  const int rc = sharded_common(p, frame_params, beta, param_constant, &comm->tr, opt_in, summary, &n);
  p->last_exchanges = n;
  return rc;
}

// The evaluation path's only collective ("one ncclAllReduce(sum, ncclDouble) per evaluation on [cost, g_beta, H_bb]"),
// in place on the caller's device buffer and stream: behind bodyfit_evaluate_device + bodyfit_reduce_shared_device (or the
// armed sweep's own tail) on the same stream it needs no host synchronisation and no Python hop.
int bodyfit_allreduce_shared_rccl(bodyfit_rccl* comm, double* d_buf66, void* stream) {
  if (!comm || !comm->tr.comm || !d_buf66) return fail(BODYFIT_ERR_INVALID, "bodyfit_allreduce_shared_rccl: bad argument");
  RcclApi& A = RcclApi::get();
  if (!A.ok()) return fail(BODYFIT_ERR_HIP, A.error);
  const int rc = A.AllReduce(d_buf66, d_buf66, 66, RcclApi::kDouble, RcclApi::kSum, comm->tr.comm, static_cast<hipStream_t>(stream));
  if (rc != 0) return fail(BODYFIT_ERR_HIP, std::string("ncclAllReduce: ") + (A.GetErrorString ? A.GetErrorString(rc) : "error"));
  return BODYFIT_OK;
}

// ranks of the communicator as RCCL itself reports them (ncclCommCount), and this process's rank in it (ncclCommUserRank)
int bodyfit_rccl_count(bodyfit_rccl* comm, int* n_ranks, int* rank) {
  if (!comm || !comm->tr.comm) return fail(BODYFIT_ERR_INVALID, "bodyfit_rccl_count: bad argument");
  RcclApi& A = RcclApi::get();
  if (!A.ok() || !A.CommCount || !A.CommUserRank) return fail(BODYFIT_ERR_HIP, A.ok() ? "librccl lacks ncclCommCount" : A.error);
  int n = 0, r = 0;
  int rc = A.CommCount(comm->tr.comm, &n);
  if (rc == 0) rc = A.CommUserRank(comm->tr.comm, &r);
  if (rc != 0) return fail(BODYFIT_ERR_HIP, std::string("ncclCommCount: ") + (A.GetErrorString ? A.GetErrorString(rc) : "error"));
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  return BODYFIT_OK;
}

// Status of the problem's asynchronous sweeps (bodyfit_evaluate_device) enqueued on `stream` so far: waits for the stream,
// then reads the one-launch sweep's error word.
int bodyfit_sweep_status(bodyfit_problem* p, void* stream) {
  if (!p) return fail(BODYFIT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(p->m->device));
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  if (p->async_stream == static_cast<hipStream_t>(stream)) p->async_pending = false;
  return fused_check(p);
}

long bodyfit_sweep_timeouts(const bodyfit_problem* p) { return p ? p->fused_timeouts : 0; }
long bodyfit_internal_fused_timeouts(const bodyfit_problem* p) { return bodyfit_sweep_timeouts(p); }   // (the tests' older name)

int bodyfit_set_exchange_timeout(bodyfit_problem* p, double seconds) {
  if (!p || !(seconds >= 0.0)) return fail(BODYFIT_ERR_INVALID, "bodyfit_set_exchange_timeout: bad argument");
  p->exchange_timeout_s = seconds;
  return BODYFIT_OK;
}

int bodyfit_set_shard_proxy(bodyfit_problem* p, int n_ranks, int rank) {
  if (!p || n_ranks < 0 || (n_ranks > 0 && (rank < 0 || rank >= n_ranks)))
    return fail(BODYFIT_ERR_INVALID, "bodyfit_set_shard_proxy: bad argument");
  p->proxy_ranks = n_ranks > 1 ? n_ranks : 0;
  p->proxy_rank = n_ranks > 1 ? rank : 0;
  return BODYFIT_OK;
}

// Test hook (not part of include/bodyfit.h): rank `rank`'s candidate sweep "fails" in LM iteration `iter` of the problem's
// next sharded solves (-1, -1: off).  tests/test_gpu_sharded_solve.py.
int bodyfit_internal_set_test_poison(bodyfit_problem* p, int rank, int iter) {
  if (!p) return fail(BODYFIT_ERR_INVALID, "null argument");
  p->test_poison_rank = rank; p->test_poison_iter = iter;
  return BODYFIT_OK;
}

long bodyfit_last_exchange_count(const bodyfit_problem* p) { return p ? p->last_exchanges : 0; }
long bodyfit_launch_count(void) { return g_launch_count.load(std::memory_order_relaxed); }

int bodyfit_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }

int bodyfit_internal_solver_view(bodyfit_problem* p, bodyfit_solver_view* out) {
  if (!p || !out) return BODYFIT_ERR_INVALID;
  out->n_frames = p->d.F; out->n_joints = p->m->nJ; out->n_shape = p->m->nS;
  out->beta_per_frame = p->desc.beta_per_frame; out->has_gmm = p->has_gmm ? 1 : 0;
  out->temporal_halo = p->desc.temporal_halo;
  out->beta_pose = p->desc.beta_pose; out->beta_shape = p->desc.beta_shape;
  out->lambda_temporal = p->desc.lambda_temporal; out->huber_delta = p->desc.huber_delta;
  out->kp_offset = p->kp_offset.data();
  out->max_kp_per_frame = 0;
  for (int f = 0; f < p->d.F; ++f)
    out->max_kp_per_frame = std::max(out->max_kp_per_frame, p->kp_offset[f + 1] - p->kp_offset[f]);
  out->prec_cho = p->has_gmm ? p->desc.gmm->prec_cho.data() : nullptr;
  return BODYFIT_OK;
}

// Test hook (not part of include/bodyfit.h): the bound of the one-launch sweep's in-launch waits, in 10 ns ticks.  A bound of
// one tick makes every wait run out, which is how tests/test_gpu_one_launch.py exercises the error word and the fall-back to
// the two-launch sweep.  0 restores the default.
int bodyfit_internal_set_role_timeout(bodyfit_problem* p, unsigned long long ticks) {
  if (!p) return fail(BODYFIT_ERR_INVALID, "null argument");
  p->role_timeout_ticks = ticks ? ticks : kRoleTimeoutDefault;
  return BODYFIT_OK;
}

#ifdef BODYFIT_STAMPS
// diagnostic builds only; not part of include/bodyfit.h
int bodyfit_debug_set_stamp_buffer(bodyfit_problem* p, unsigned long long* d_buf) {
  p->d.dbg = d_buf;
  return BODYFIT_OK;
}
#endif

int bodyfit_writeback_batch(bodyfit_problem* p, const double* frame_params, const double* beta, double* R0_out,
                            double* joints, float* cloud, double* mean_px) {
  if (!p || !frame_params) return fail(BODYFIT_ERR_INVALID, "null argument");
  const bodyfit_model* m = p->m;
  const int npose = 7 + 3 * (m->nJ - 1), F = p->d.F;
  const bool has_beta = p->lay.n_cols > npose;
  if (cloud && !p->desc.want_mesh) return fail(BODYFIT_ERR_INVALID, "problem was created without want_mesh");
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  p->cache_valid = false;
  const size_t npar = (size_t)p->n_param_rows * npose;
  const size_t nbeta = (has_beta && beta) ? (size_t)(p->desc.beta_per_frame ? F * m->nS : m->nS) : 0;
  if (!p->d_writeback) HIP_TRY(p->mem.alloc(&p->d_writeback, (size_t)p->n_param_rows * npose + (size_t)F * 10));
  double* d_upd = p->d_writeback;
  double* d_R0n = d_upd + (size_t)p->n_param_rows * npose;
  double* d_px = d_R0n + (size_t)F * 9;
  if (int ro = order_after_async(p, nullptr)) return ro;
  HIP_TRY(hipMemcpyAsync(p->d_params, frame_params, npar * sizeof(double), hipMemcpyHostToDevice, nullptr));
  if (nbeta) HIP_TRY(hipMemcpyAsync(p->d_beta, beta, nbeta * sizeof(double), hipMemcpyHostToDevice, nullptr));
  launch_writeback_prepare(F, npose, p->d_params, p->d.R0, d_upd, d_R0n, nullptr);
  for (int attempt = 0;; ++attempt) {   // (a one-launch sweep whose in-launch wait ran out is re-issued as two launches)
    int rc = sweep(p, d_upd, nbeta ? p->d_beta : nullptr, 0, p->desc.want_mesh != 0, nullptr, nullptr, nullptr, nullptr,
                   nullptr, 0, d_R0n);
    if (rc) return rc;
    launch_mean_pixel_error(F, m->nJ, p->d.kp_offset, p->d.kp_id, p->d.kp_uv, p->d_joints, p->d.fx, p->d.fy, p->d.cx,
                            p->d.cy, d_px, nullptr);
    if (R0_out) HIP_TRY(hipMemcpyAsync(R0_out, d_R0n, (size_t)F * 9 * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    if (mean_px) HIP_TRY(hipMemcpyAsync(mean_px, d_px, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    if (joints)
      HIP_TRY(hipMemcpyAsync(joints, p->d_joints, (size_t)F * m->nJ * 3 * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    if (cloud) {
      const size_t row = (size_t)m->V * 3 * sizeof(float), pitch = (size_t)m->d.nVTiles * kVTile * 3 * sizeof(float);
      HIP_TRY(hipMemcpy2DAsync(cloud, row, p->d_cloud, pitch, row, (size_t)F, hipMemcpyDeviceToHost, nullptr));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipGetLastError());
    if (attempt == 0 && fused_timed_out(p)) continue;
    break;
  }
  return BODYFIT_OK;
}

int bodyfit_forward(bodyfit_problem* p, const double* frame_params, const double* beta, double* joints,
                    float* cloud) {
  if (!p || !frame_params) return fail(BODYFIT_ERR_INVALID, "null argument");
  const bodyfit_model* m = p->m;
  const int npose = 7 + 3 * (m->nJ - 1);
  const bool has_beta = p->lay.n_cols > npose;
  if (cloud && !p->desc.want_mesh) return fail(BODYFIT_ERR_INVALID, "problem was created without want_mesh");
  HIP_TRY(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lock(p->mu);
  p->cache_valid = false;
  const size_t npar = (size_t)p->n_param_rows * npose;
  const size_t nbeta = (has_beta && beta) ? (size_t)(p->desc.beta_per_frame ? p->d.F * m->nS : m->nS) : 0;
  if (int ro = order_after_async(p, nullptr)) return ro;
  HIP_TRY(hipMemcpy(p->d_params, frame_params, npar * sizeof(double), hipMemcpyHostToDevice));
  if (nbeta) HIP_TRY(hipMemcpy(p->d_beta, beta, nbeta * sizeof(double), hipMemcpyHostToDevice));
  for (int attempt = 0;; ++attempt) {   // (a one-launch sweep whose in-launch wait ran out is re-issued as two launches)
    int rc = sweep(p, p->d_params, nbeta ? p->d_beta : nullptr, 0, cloud != nullptr, nullptr);
    if (rc) return rc;
    if (joints)
      HIP_TRY(hipMemcpy(joints, p->d_joints, (size_t)p->d.F * m->nJ * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (cloud) {
      const size_t row = (size_t)m->V * 3 * sizeof(float), pitch = (size_t)m->d.nVTiles * kVTile * 3 * sizeof(float);
      HIP_TRY(hipMemcpy2D(cloud, row, p->d_cloud, pitch, row, (size_t)p->d.F, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipDeviceSynchronize());
    if (attempt == 0 && fused_timed_out(p)) continue;
    break;
  }
  return BODYFIT_OK;
}

double bodyfit_mean_pixel_error(int n_kp, const int* jid, const double* uv, const double* joints, double fx,
                                double fy, double cx, double cy) {
  if (n_kp <= 0) return 0.0;  // include/Utils.h:106
  double sum = 0.0;
  for (int k = 0; k < n_kp; ++k) {
    const double* J = joints + 3 * jid[k];
    const double u = fx * J[0] / J[2] + cx, v = fy * J[1] / J[2] + cy;
    sum += std::hypot(u - uv[2 * k], v - uv[2 * k + 1]);
  }
  return sum / n_kp;
}

// ------------------------------------------------------------------------------------------------
// ceres::CostFunction::Evaluate for one block, served from the cached sweep when the caller's
// parameters match it; otherwise the affected frame is re-evaluated on the device first.
// ------------------------------------------------------------------------------------------------
// kinds 0 / 1 of bodyfit_evaluate_block from the cached sweep (page-locked mirrors of the last bodyfit_evaluate_batch)
static void serve_block(bodyfit_problem* p, int kind, int index, int frame, double* residuals, double** jacobians) {
  const bodyfit_model* m = p->m;
  const int nJ = m->nJ, nS = m->nS, npose = 7 + 3 * (nJ - 1), D = 3 * (nJ - 1);
  const bodyfit_layout& L = p->lay;
  const bool has_beta = L.n_cols > npose;
  {
  if (kind == 0) {
    residuals[0] = p->c_r[2 * (size_t)index];
    residuals[1] = p->c_r[2 * (size_t)index + 1];
    if (jacobians && p->cache_packed) {
      // packed cache: the keypoint's present blocks in block order, each ALREADY in Ceres' layout ([2][size] row-major); the
      // others are zero.  Where a block starts comes from a per-keypoint table made with the pack tables: a present 3-column
      // block is one 48-byte copy, an absent one 48 bytes of zeros (at C3 the 6,400 reprojection blocks of an evaluation point
      // are three quarters of the Ceres-side time)
      const double* __restrict__ P = p->c_Jp.data() + p->pk_off[index];
      const short* __restrict__ src = p->pk_src.data() + (size_t)index * 32;
      const int nj3 = 3 + (nJ - 1);
      if (double* J = jacobians[0]) {
        const int a = src[0];
        if (a >= 0) { J[0] = P[a]; J[1] = P[a + 1]; } else { J[0] = 0.0; J[1] = 0.0; }
      }
      for (int blk = 1; blk < nj3; ++blk) {
        double* __restrict__ J = jacobians[blk];
        if (!J) continue;
        const int a = src[blk];
        if (a >= 0) __builtin_memcpy(J, P + a, 48);
        else __builtin_memset(J, 0, 48);
      }
      if (has_beta) {
        if (double* J = jacobians[nj3]) {
          const int a = src[nj3];
          if (a >= 0) std::memcpy(J, P + a, (size_t)2 * nS * sizeof(double));
          else std::memset(J, 0, (size_t)2 * nS * sizeof(double));
        }
      }
    } else if (jacobians) {
      const double* J0 = p->c_J.data() + (size_t)(2 * index) * L.n_cols;
      const double* J1 = J0 + L.n_cols;
      const int nblocks = 3 + (nJ - 1) + (has_beta ? 1 : 0);
      for (int blk = 0; blk < nblocks; ++blk) {
        if (!jacobians[blk]) continue;
        const int off = blk == 0 ? 0 : (blk == 1 ? 1 : (blk == 2 ? 4 : (blk < 3 + (nJ - 1) ? 7 + 3 * (blk - 3) : npose)));
        const int sz = blk == 0 ? 1 : (blk < 3 + (nJ - 1) ? 3 : nS);
        for (int i = 0; i < sz; ++i) {
          jacobians[blk][i] = J0[off + i];
          jacobians[blk][sz + i] = J1[off + i];
        }
      }
    }
  } else {
    const int nRes = L.prior_rows_per_frame;
    const double* r = p->c_r.data() + p->row_prior + (size_t)frame * nRes;
    std::memcpy(residuals, r, (size_t)nRes * sizeof(double));
    if (jacobians) {
      const double bp = p->desc.beta_pose;
      const int comp = p->c_comp[frame];
      for (int j = 0; j < nJ - 1; ++j) {
        if (!jacobians[j]) continue;
        double* Jb = jacobians[j];  // nRes x 3 row-major (include/Sim3BA.h:293,306)
        if (!(p->has_gmm && !p->gmm_jt.empty())) std::fill(Jb, Jb + (size_t)nRes * 3, 0.0);
        if (p->has_gmm && !p->gmm_jt.empty()) {
          // beta_pose L_k^T, one joint's three columns as the contiguous [nRes][3] block Ceres asks for (built once per problem:
          // the transposed walk over L_k was most of a prior block's Evaluate)
          std::memcpy(Jb, p->gmm_jt.data() + ((size_t)comp * (nJ - 1) + j) * nRes * 3, (size_t)nRes * 3 * sizeof(double));
        } else if (p->has_gmm) {
          const double* Lk = p->desc.gmm->prec_cho.data() + (size_t)comp * D * D;
          for (int row = 0; row < D; ++row)
            for (int c = 0; c < 3; ++c) Jb[(size_t)row * 3 + c] = Lk[(size_t)(3 * j + c) * D + row] * bp;  // :298-299
        } else {
          for (int c = 0; c < 3; ++c) Jb[(size_t)(3 * j + c) * 3 + c] = bp;  // :308-309
        }
      }
    }
  }
}
}

int bodyfit_evaluate_block(bodyfit_problem* p, int kind, int index, const double* const* parameters,
                           double* residuals, double** jacobians) {
  if (!p || !parameters || !residuals) return fail(BODYFIT_ERR_INVALID, "null argument");
  const bodyfit_model* m = p->m;
  const int nJ = m->nJ, nS = m->nS, npose = 7 + 3 * (nJ - 1);
  const bodyfit_layout& L = p->lay;
  const bool has_beta = L.n_cols > npose;
  if (kind == 1) {  // pose prior: constant-structure Jacobian, evaluate through the batch of frame `index`
    if (index < 0 || index >= p->d.F || L.prior_rows_per_frame == 0) return fail(BODYFIT_ERR_INVALID, "bad prior block");
  }
  if (kind == 2) {
    if (L.shape_rows == 0) return fail(BODYFIT_ERR_INVALID, "no shape prior in this problem");
    const double bs = p->desc.beta_shape;
    for (int i = 0; i < nS; ++i) residuals[i] = bs * parameters[0][i];          // include/Sim3BA.h:336
    if (jacobians && jacobians[0]) {
      std::fill(jacobians[0], jacobians[0] + (size_t)nS * nS, 0.0);
      for (int i = 0; i < nS; ++i) jacobians[0][(size_t)i * nS + i] = bs;       // :338-340
    }
    return BODYFIT_OK;
  }
  if (kind == 3) {
    const double lam = p->desc.lambda_temporal;
    for (int i = 0; i < 3; ++i) residuals[i] = (parameters[0][i] - parameters[1][i]) * lam;  // MultiFrameBA.h:24
    if (jacobians) {
      for (int b = 0; b < 2; ++b)
        if (jacobians[b]) {
          std::fill(jacobians[b], jacobians[b] + 9, 0.0);
          for (int i = 0; i < 3; ++i) jacobians[b][i * 3 + i] = b == 0 ? lam : -lam;
        }
    }
    return BODYFIT_OK;
  }
  if (kind != 0 && kind != 1) return fail(BODYFIT_ERR_INVALID, "unknown block kind");
  int frame;
  if (kind == 0) {
    if (index < 0 || index >= L.n_keypoints) return fail(BODYFIT_ERR_INVALID, "keypoint index out of range");
    frame = p->kp_frame[index];
  } else {
    frame = index;
  }
  // gather the caller's parameter blocks into the packed frame row (stack arrays: this function runs once per residual
  // block and Ceres thread, nothing on its hit path allocates)
  double x[kFrameParams] = {0.0}, b[kMaxShape] = {0.0};
  if (kind == 0) {
    x[0] = parameters[0][0];
    for (int i = 0; i < 3; ++i) { x[1 + i] = parameters[1][i]; x[4 + i] = parameters[2][i]; }
    for (int j = 1; j < nJ; ++j)
      for (int i = 0; i < 3; ++i) x[7 + 3 * (j - 1) + i] = parameters[3 + (j - 1)][i];
    if (has_beta)
      for (int i = 0; i < nS; ++i) b[i] = parameters[3 + (nJ - 1)][i];
  } else {
    for (int j = 1; j < nJ; ++j)
      for (int i = 0; i < 3; ++i) x[7 + 3 * (j - 1) + i] = parameters[j - 1][i];
  }
  {
    std::unique_lock<std::mutex> lock(p->mu);
    bool hit = p->cache_valid && (p->cache_has_jac || !jacobians);
    if (hit) {
      const double* cx = p->c_params.data() + (size_t)frame * npose;
      const int i0 = (kind == 0) ? 0 : 7;
      hit = p->c_npar == (size_t)p->n_param_rows * npose && std::memcmp(cx + i0, x + i0, (size_t)(npose - i0) * sizeof(double)) == 0;
      if (hit && kind == 0 && has_beta) {
        const double* cb = p->c_beta.data() + (p->desc.beta_per_frame ? (size_t)frame * nS : 0);
        hit = std::memcmp(cb, b, (size_t)nS * sizeof(double)) == 0;
      }
    }
    if (!hit) {
      // refresh the cached parameter set with this frame's values and sweep again
      std::vector<double> par(p->c_params.data(), p->c_params.data() + p->c_npar), be(p->c_beta.data(), p->c_beta.data() + p->c_nbeta);
      if (par.size() != (size_t)p->n_param_rows * npose) {
        par.assign((size_t)p->n_param_rows * npose, 0.0);
        for (int f = 0; f < p->n_param_rows; ++f) { par[(size_t)f * npose] = 1.0; par[(size_t)f * npose + 6] = 3.0; }
      }
      const size_t nb = has_beta ? (size_t)(p->desc.beta_per_frame ? p->d.F * nS : nS) : 0;
      if (be.size() != nb) be.assign(nb, 0.0);
      const int i0 = (kind == 0) ? 0 : 7;
      std::memcpy(par.data() + (size_t)frame * npose + i0, x + i0, (size_t)(npose - i0) * sizeof(double));
      if (kind == 0 && has_beta)
        std::memcpy(be.data() + (p->desc.beta_per_frame ? (size_t)frame * nS : 0), b, (size_t)nS * sizeof(double));
      lock.unlock();
      int rc = bodyfit_evaluate_batch(p, par.data(), nb ? be.data() : nullptr, nullptr, nullptr, nullptr, 1);
      if (rc) return rc;
      lock.lock();
    }
    serve_block(p, kind, index, frame, residuals, jacobians);
  }
  return BODYFIT_OK;
}

// The EvaluationCallback form (include/bodyfit_ceres.h: SweepCallback): the caller guarantees that the cached sweep IS the point
// Ceres is evaluating (PrepareForEvaluation ran for it), so kinds 0 / 1 are served without gathering and comparing the block's
// 76 parameters and without the problem's lock (the cache is only written by the callback, between evaluations): ~4x less host
// time per block, and Ceres' evaluation threads do not serialise on it.  Kinds 2 / 3 are functions of their parameters alone.
int bodyfit_evaluate_block_cached(bodyfit_problem* p, int kind, int index, const double* const* parameters,
                                  double* residuals, double** jacobians) {
  if (!p || !residuals) return fail(BODYFIT_ERR_INVALID, "null argument");
  if (kind == 2 || kind == 3) return bodyfit_evaluate_block(p, kind, index, parameters, residuals, jacobians);
  if (kind != 0 && kind != 1) return fail(BODYFIT_ERR_INVALID, "unknown block kind");
  if (!p->cache_valid || (jacobians && !p->cache_has_jac))
    return fail(BODYFIT_ERR_INVALID, "bodyfit_evaluate_block_cached: no sweep cached for this evaluation (EvaluationCallback not run?)");
  int frame;
  if (kind == 0) {
    if (index < 0 || index >= p->lay.n_keypoints) return fail(BODYFIT_ERR_INVALID, "keypoint index out of range");
    frame = p->kp_frame[index];
  } else {
    if (index < 0 || index >= p->d.F || p->lay.prior_rows_per_frame == 0) return fail(BODYFIT_ERR_INVALID, "bad prior block");
    frame = index;
  }
  serve_block(p, kind, index, frame, residuals, jacobians);
  return BODYFIT_OK;
}

}  // extern "C"
