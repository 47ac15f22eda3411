// bodyfit_device.h — device-side views shared by the HIP kernels and the host API (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include "device_once.h"   // DeviceOnce / DeviceMax: per-device bookkeeping of kernel attributes

namespace bodyfit {

constexpr int kMaxJoints = 24;     // SMPL
constexpr int kMaxShape = 10;
constexpr int kMaxLandmarks = 32;  // vertex-landmark keypoints per model
constexpr int kMaxLmNnz = 8;       // skinning weights per landmark vertex (SMPL has <= 4)
constexpr int kFrameParams = 76;   // [s, rootAA, rootT, jointAA[1..23]]
constexpr int kMeshNnz = 4;        // packed skinning weights per mesh vertex

// mesh operand geometry (k_mesh_blend_lbs.hip)
constexpr int kVTile = 32;         // vertices per MFMA column tile
constexpr int kFTile = 32;         // frames per MFMA row tile
constexpr int kPoseFeat = 207;     // 9 x 23 pose-corrective features
constexpr int kBlendKSteps = 14;   // K = 207 pose features + 10 shape coefficients = 217 -> 224 = 14 x 16 (bf16 32x32x16)

// The small model tables the frame role reads in its first phase live in ONE device block at fixed offsets (sized for the
// maxima), and a problem's keypoint tables in another: the one-launch sweep then needs two pointers for them instead of
// sixteen, few enough to arrive as preloaded kernel arguments (k_sweep.hip FrameHead).  DevModel's / DevProblem's own pointers
// point into the same blocks.
constexpr int kTabParent = 0;                                   // int[32]
constexpr int kTabAnc = kTabParent + 32 * 4;                    // unsigned[32]
constexpr int kTabChain = kTabAnc + 32 * 4;                     // unsigned long long[32]
constexpr int kTabLmWoff = kTabChain + 32 * 8;                  // int[kMaxLandmarks + 8]
constexpr int kTabLmWj = kTabLmWoff + (kMaxLandmarks + 8) * 4;  // int[kMaxLandmarks][kMaxLmNnz]
constexpr int kTabLmWw = kTabLmWj + kMaxLandmarks * kMaxLmNnz * 4;       // double[kMaxLandmarks][kMaxLmNnz]
constexpr int kTabLmVt = kTabLmWw + kMaxLandmarks * kMaxLmNnz * 8;       // double[kMaxLandmarks][3]
constexpr int kTabLmSd = kTabLmVt + kMaxLandmarks * 3 * 8;               // double[kMaxLandmarks][3][kMaxShape]
constexpr int kTabDS = kTabLmSd + kMaxLandmarks * 3 * kMaxShape * 8;     // double[kMaxJoints][3][kMaxShape]
constexpr int kTabSc = kTabDS + kMaxJoints * 3 * kMaxShape * 8;
constexpr int kTabOffset = kTabSc + kMaxJoints * 3 * kMaxShape * 8;      // double[kMaxJoints][3]
constexpr int kTabJc0 = kTabOffset + kMaxJoints * 3 * 8;
constexpr int kTabBytes = kTabJc0 + kMaxJoints * 3 * 8;
static_assert(kTabLmWw % 8 == 0 && kTabChain % 8 == 0, "f64 / u64 tables on 8-byte offsets");
// a problem's keypoint block: [kp_offset int[F + 1]] [kp_id int[K + 32]] [kp_uv double[2 (K + 32)]], each on a 16-byte offset
__host__ __device__ inline int ptab_id_off(int F) { return ((F + 1) * 4 + 15) & ~15; }
__host__ __device__ inline int ptab_uv_off(int F, int K) { return ptab_id_off(F) + (((K + 32) * 4 + 15) & ~15); }

struct DevModel {
  int V, nJ, nS, P, nL, nLevels, nVTiles;
  const unsigned char* tabA;   // the block above
  // skeleton, f64
  const int* parent;           // [nJ]
  const int* level_off;        // [nLevels+1]  joints of depth d+1
  const int* level_joint;      // [nJ-1]
  const unsigned* anc_mask;    // [nJ] bit k: k is a proper ancestor of j, k != root
  const unsigned long long* anc_chain;   // [nJ] the same ancestors as a walk list: nearest first, 5 bits each, 0-terminated
  const double* offset;        // [nJ][3]       include/Sim3BA.h:372-392
  const double* dS;            // [nJ][3][nS]   S_j - S_par(j)   (j = 0: S_0)
  const double* Jc0;           // [nJ][3]       rest joints, root at origin, beta = 0
  const double* Sc;            // [nJ][3][nS]   S_j - S_0
  // vertex landmarks, f64
  const int* lm_woff;          // [nL]  number of skinning weights of the landmark vertex
  const int* lm_wj;            // [nL][kMaxLmNnz] joint ids, padded
  const double* lm_ww;         // [nL][kMaxLmNnz] weights, padded with 0
  const double* lm_vt;         // [nL][3]       v_template - J0_root
  const double* lm_sd;         // [nL][3][nS]   shapedirs  - S_root
  const double* lm_pd;         // [nL][27 = 3 coords x 9 entries][32 >= nJ - 1] posedirs of the landmark vertex, joint-minor
  const int* lm_gcount;        // [32] or NULL: slots of a keypoint regressor row are summed into its first slot, whose entry is the
                               // row's slot count (other entries: 1 = plain landmark, 0 = member of a row)
  // mesh operands (packed in MFMA fragment order at upload)
  const uint16_t* dirsB;       // [nVTiles][kBlendKSteps][3][2 hi/lo][64][8] bf16: posedirs, then shapedirs - S_root
  const float* vtB;            // [nVTiles][3][32] f32
  const uint32_t* wIdx;        // [nVTiles*32] 4 x u8 joint ids
  const float* wVal;           // [nVTiles*32][4]
};

struct DevProblem {
  int F, K, ncols, use_shape, beta_stride, pose_blend, nFTiles;
  const unsigned char* ptab;   // the keypoint block above (kp_offset / kp_id / kp_uv point into it)
  int feat_perm;          // row order of the blend-coefficient fragments (frame_part_inl.h), set per launch
  const int* kp_offset;   // [F+1]
  const int* kp_id;       // [K]
  const double* kp_uv;    // [K][2]
  const double* R0;       // [F][9]
  double fx, fy, cx, cy;
  unsigned long long* dbg;   // diagnostic builds only (-DBODYFIT_STAMPS): per-phase s_memtime stamps
  const int* frame_flags;    // optional [F]: frames whose bit `frame_mask` is clear are skipped (device LM)
  int frame_mask;
  // shared-beta reduction folded into k_frame_resjac: per frame [cost, g_beta, upper H_bb] of the robustified
  // reprojection rows, in k_reduce's 258-entry partial layout (null: not produced)
  double* beta_partials;
  double huber;
};

// Byte offset of one MFMA A fragment (row = lane & 31, k-half = lane >> 5; hl: 0 = bf16 hi part, 1 = lo part) inside the
// 2 KiB block of a k-step: [hi | lo][2 k-halves][32 rows][8 bf16], i.e. a wave's 64 fragments of one part are 1 KiB of
// contiguous memory (eight lines per load).  Measured and rejected: [32 rows][hi | lo][k-half] (a frame publishes 64
// contiguous bytes per k-step instead of four 16-byte pieces) — the hand-off was no earlier and every fragment load then
// touches sixteen lines (mesh role prologue 0.7 -> 1.1 us, step 24.1 -> 24.8 us).
__host__ __device__ inline unsigned feat_frag_off(int lane, int hl) { return (unsigned)(hl * 1024 + lane * 16); }

// operands the per-frame kernel prepares for the mesh kernel
struct MeshCoef {
  uint16_t* featA;   // [nFTiles][kBlendKSteps][2 hi/lo][64][8] bf16 (feat_frag_off): pose features, then beta
  float* skinT;      // [F][nJ][12] f32: rows of [s R_root R0 A_j | s R_root R0 (P_j - A_j Jc_j) + t]
};

struct DevGmm {
  int K, D;
  const double* mean;       // [K][D]
  const double* prec_cho;   // [K][D][D] lower
  const double* prec_frag;  // [K][18 k-steps][3 column-tile pairs][64 lanes][2]: L in f64-MFMA B-fragment order
  const double* prec;       // [K][D][D] precision matrices L L^T (normal-equation block of the prior)
  const double* neg_log_w;  // [K]
  double resid_scale;
};

// prior residuals computed by extra workgroups of the k_frame_resjac launch (priors_inl.h)
constexpr int kReducePartial = 258;   // entries per reduction partial: 16 x 16 Gram tile (H_bb upper + g_beta in column 10), huber cost, plain cost
// The per-frame partials of the sweep (k_frame_resjac / k_sweep_roles, priors_inl.h) use the first kFoldEntries slots of such a
// row, COMPACT: [huber cost | plain cost | Gram entries (i, j), i < 10, i <= j <= 10, row after row] — the 67 numbers the
// reduction needs, 5 lines instead of 17 for the workgroup that sums them inside the launch (one CU pulls every partial).
constexpr int kFoldEntries = 67;
__host__ __device__ inline int fold_slot_cost(int which) { return which; }                                  // 0: huber, 1: plain
__host__ __device__ inline int fold_slot_gram(int i, int j) { return 2 + 11 * i - (i * (i - 1)) / 2 + (j - i); }   // j >= i

struct PriorArgs {
  int F, nS, beta_stride, has_gmm, n_pairs, n_tiles;   // n_tiles = 0: no prior workgroups
  double beta_pose, beta_shape, lambda_t;
  DevGmm g;
  const double* beta;
  double* r_prior;
  double* r_shape;
  double* r_temporal;
  int* comp;
  double* plain_cost;   // optional [n_tiles][258]: entry 257 of row `tile` receives the tile's 1/2 sum r^2 (folded reduction)
};

// in-launch synchronisation words of the one-launch sweep (k_sweep_roles), one set per problem, zeroed at creation only.
// Per 32-frame unit u: flag[u] == epoch x (frames of the unit) once all of them have handed their mesh operands over in launch
// `epoch` (every frame workgroup adds 1 per launch; nothing is reset between launches).
constexpr size_t kFusedSyncHeader = 256;  // error word, pad (the counters start on a line of their own)
// One counter per 32-frame unit, 256 bytes apart: an agent-scope atomic add executes at the memory side at ~12 ns per add
// and LINE (measured: 512 adds into one line took 6 us and held back every store queued behind them on that channel), so a
// unit's 32 adds must not share a line, or a channel, with the other units'.
constexpr int kUnitCounterStride = 64;    // dwords
constexpr int kUnitCoefOffset = 32;   // unsigneds: the unit's SECOND counter (blend coefficients published), 128 bytes behind the first
// tuning words of the one-launch sweep (round 4 read them from the environment; the values are that round's measured optimum:
// profiles/r4_x_tuning_runs.txt).  A diagnostic build with -DBODYFIT_TUNE_ENV reads them from BODYFIT_MESH_PRIO /
// BODYFIT_TRICKLE_START / BODYFIT_TRICKLE_SLEEP / BODYFIT_J_SCOPE again.
constexpr int kTuneMeshPrio = 2, kTuneTrickleStart = 120, kTuneTrickleSleep = 7, kTuneJScope = 1;
struct FusedSync {
  unsigned* flag;              // [frames / 32, rounded up to whole groups of 8][kUnitCounterStride]: word 0 counts the frames whose
                               // skinning transforms are published, word kUnitCoefOffset those whose blend coefficients are
  unsigned* error;             // set when a workgroup's bounded wait ran out
  unsigned epoch;              // launch number, >= 1
  int resident_blocks;         // blocks resident from the start of the launch (2 per CU)
  unsigned long long timeout_ticks;   // bound of every in-launch wait (s_memrealtime ticks, 100 MHz); kRoleTimeoutDefault
  // tuning (defaults: RoleTuning; BODYFIT_MESH_PRIO / BODYFIT_TRICKLE_START / BODYFIT_TRICKLE_SLEEP override them for A/B runs)
  int mesh_prio_early;         // s_setprio of a mesh wave during k-steps 0-8 (the frame role runs at 2 and still owes the transforms)
  int trickle_start;           // operand stream of a mesh workgroup that runs beside its frames: first slab this many 10 ns ticks
  int trickle_sleep;           // after entry, then s_sleep(this) between slabs
  int j_scope;                 // frame role: cache policy of the Jacobian panel's stores (frame_part_inl.h store_J)
};
constexpr unsigned long long kRoleTimeoutDefault = 5000000;   // 50 ms: give up, set the error word, the host falls back
constexpr int kRoleMaxFrames = 16384;

// 8-byte write-through store (sc1): the payload form of a hand-off to a workgroup on another XCD inside the launch
__device__ __forceinline__ void store_f64_through(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Tail of a one-launch Jacobian sweep of a shared-shape problem (bodyfit_arm_shared_reduction): every frame / prior workgroup
// takes a ticket behind its partial ([258] doubles, k_reduce.hip's layout); the LAST one sums the partials in k_reduce_stage2's
// order and packs [cost | g_beta (10) | upper H_bb (55)] into out66, while the mesh workgroups are still running: the sharded
// evaluation needs no reduction launch behind the sweep.
struct FoldTail {
  unsigned* ticket;          // null: no fold.  Counts up over the launches that fold (never reset between them)
  unsigned want;             // ticket value after this launch's last workgroup
  int n_partials;            // frames + prior tiles, <= kFoldMaxPartials
  const double* partials;    // [n_partials][kReducePartial]
  const double* beta;        // shared shape coefficients (shape-prior terms of the gradient)
  int shape_rows;
  double beta_shape;
  double* out66;
};
constexpr int kFoldMaxPartials = 256;
constexpr int kFoldTicketOffset = 128;   // byte offset of the ticket inside the sync header (a line of its own)

// every kernel launch of the library goes through these two: a process-wide count (bodyfit_launch_count) lets the benchmarks
// report launches per LM iteration beside microseconds per iteration
extern std::atomic<long> g_launch_count;
#define BODYFIT_LAUNCH(...) do { ::bodyfit::g_launch_count.fetch_add(1, std::memory_order_relaxed); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define BODYFIT_LAUNCH_EXT(...) do { ::bodyfit::g_launch_count.fetch_add(1, std::memory_order_relaxed); hipExtLaunchKernelGGL(__VA_ARGS__); } while (0)

inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}

// ---- device-resident LM for batches of independent frames (k_lm_batched.hip) ------------------------
constexpr int kLmActive = 1;        // flags: frame still iterating
constexpr int kLmHasCand = 2;       //        a candidate point awaits its residual sweep
constexpr int kLmTermShift = 4;     //        termination code (0 convergence, 1 iteration limit, 2 failure) << 4
constexpr int kLmTermMask = 3 << kLmTermShift;

struct LmProblem {
  int F, ncols;
  const int* kp_offset;       // [F+1] device
  double huber, beta_pose, beta_shape, scale_lo, scale_hi;
  int prior_rows, row_prior;              // pose prior rows per frame, first row in the residual vector
  int shape_rows_per_frame, row_shape;    // per-frame shape prior rows
  const double* prec;         // [K][69][69] or null (L2 prior)
  const double* prec_cho;     // [K][69][69]
  const double* gmm_mean;     // [K][69]
  double gmm_scale;           // resid_scale of the mixture residual
};

struct LmState {              // all device pointers, one entry (or row) per frame
  double *x, *beta, *x_new, *beta_new;    // [F][76], [F][nb]
  double *radius, *dec, *cost, *initial_cost, *model, *scale;   // scale [F][86]
  int *flags, *iters, *n_ok, *n_bad;
  int* active_count;
};

size_t lm_step_lds_bytes();
void launch_lm_init(const LmProblem& P, const LmState& S, const double* d_r, hipStream_t s);
void launch_lm_step(const LmProblem& P, const LmState& S, double* d_r, double* d_J, int* d_comp, const double* d_r_cand,
                    const double* d_J_cand, const int* d_comp_cand, const unsigned char* d_constant, int first_iter,
                    hipStream_t s);
void launch_lm_accept(const LmProblem& P, const LmState& S, const double* d_r_new, double* d_r_cur, const int* d_comp_new,
                      int* d_comp_cur, hipStream_t s);
constexpr int kNormalRows = 87, kNormalLd = 88;   // per-frame normal-equation panel of k_frame_normal: (n + 1) x 88, lower
void launch_frame_normal(int F, int n, const int* d_kp_offset, double huber, const double* d_r, const double* d_J,
                         double* d_out, hipStream_t s);
// The window LM's form: `sel` (device, W.status + kWsJsel) says whether the panels are rebuilt from the starting point's (r, J)
// (0), from the accepted candidate's (r_alt, J_alt) (1: the candidate's rows are also copied into r, which the assembly reads), or
// left alone (2: the last step was rejected).
void launch_frame_normal_sel(int F, int n, const int* d_kp_offset, double huber, double* d_r, const double* d_J,
                             const double* d_r_alt, const double* d_J_alt, const double* d_sel, int total_rows, double* d_out,
                             hipStream_t s);

// ---- device-resident LM for one shared-beta window (k_window_lm.hip): block cyclic reduction over the frames ----------
constexpr int kWinBlock = 80;     // 76 frame parameters padded to whole 16-column panels (identity on the padding)
constexpr int kWinRhs = 16;       // [B (10 columns) | rhs] transposed, padded to one 16-row tile
constexpr int kWinPart = 128;     // per-frame partials: Schur S_f (100) + rb_f (10) + pad + model / |d|^2 / |x|^2 at 112..114
enum {   // status record of the window LM (doubles), read back by the host once per iteration
  kWsCost = 0, kWsRadius, kWsDec, kWsModel, kWsHasCand, kWsIters, kWsOk, kWsBad, kWsTermination, kWsActive,
  kWsInitialCost, kWsAccepted, kWsGmax, kWsNewCost,
  kWsJsel,          // which Jacobian the next k_frame_normal reads: 0 the starting point's, 1 the accepted candidate's, 2 none (rejected)
  kWsPoison,        // sharded solves: some rank reported a device failure in its scalars (slot 6); the solve has ended on every rank
  kWsCount = 16
};
struct WinProblem {
  int F, K, total_rows, nb;
  int halo;                         // shard of a window: a temporal pair links the last frame to the next shard's first
  int prior_rows, row_prior, shape_rows, row_shape, row_temporal;
  double huber, beta_pose, beta_shape, lambda_t, scale_lo, scale_hi;
  // the cost of the point a Jacobian sweep has just evaluated, as that sweep's own per-workgroup partials ([F + cost_tiles]
  // rows of kReducePartial doubles: frame rows carry 1/2 sum rho over their keypoints in slot 0, prior tiles 1/2 |r|^2 of their
  // rows in slot 1); null: the cost is summed from the residual vector
  const double* cost_partials;
  int cost_tiles;
};
struct WinBuf {
  double *D, *U, *L, *Pt, *Qt;      // [F][80][80]: diagonal blocks, couplings, factors, solved couplings (transposed)
  double *Rt, *Rt0, *Yt, *Xt;       // [F][16][80]: right-hand sides (working copy, as assembled), L^-1 R, solution
  double *Li;                       // [F][5][16][16]: L_pp^-T of the factor's diagonal blocks (k_cr_factor -> k_cr_back)
  double *Araw, *Braw, *graw, *Eraw;   // undamped, unscaled blocks for the model cost change: [F][76][76], [F][76][10], [F][76] x 2
  double *scale;                    // [F * 76 + 10] Jacobi scaling, fixed at the first iterate
  double *Cs, *rhsb, *Craw, *gbraw, *dsb;   // beta block: scaled damped C, scaled rhs, raw C, raw gradient, scaled step
  double *sred;                     // [110] Schur partial sums of this shard (sharded solve: all-reduced by the host)
  double *fin;                      // [8] scalars a sharded solve reduces over the shards
  double *part;                     // [F][kWinPart]
  double *gmaxp;                    // [F + 1] per-frame max |g| (entry F: beta)
  double *d;                        // [F * 76 + 10] the step
  double *status;                   // [kWsCount]
  int* fail;                        // a factorisation met a non-positive pivot
  int* ticket;                      // k_win_tail: workgroups that have delivered their partial (reset by the last one)
};
size_t win_factor_lds_bytes();
// (mode 0: single GPU, everything in the kernel; 1: this shard's partial sums only; 2: finish from the sums the host reduced)
void launch_win_init(const WinProblem& P, const WinBuf& W, const double* d_r, int mode, hipStream_t s);
void launch_win_beta(const WinProblem& P, const WinBuf& W, const double* d_Hpan, const double* d_r, int first, int mode,
                     hipStream_t s);
void launch_win_assemble(const WinProblem& P, const WinBuf& W, const double* d_Hpan, const double* d_r, const double* d_x,
                         const unsigned char* d_constant, int first, const double* d_x_left, const double* d_scale_halo,
                         hipStream_t s);
void launch_cr_factor(const WinBuf& W, const int* d_elim, int n_elim, hipStream_t s);
void launch_cr_update(const WinBuf& W, const int* d_surv, int n_surv, hipStream_t s);
void launch_cr_back(const WinBuf& W, const int* d_elim, int n_elim, hipStream_t s);
void launch_win_schur_part(const WinProblem& P, const WinBuf& W, hipStream_t s);
void launch_win_beta_solve(const WinProblem& P, const WinBuf& W, const double* d_beta, double* d_beta_new, int mode, hipStream_t s);
void launch_win_tail(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_beta, double* d_x_new, double* d_beta_new,
                     hipStream_t s);
void launch_win_step(const WinProblem& P, const WinBuf& W, const double* d_x, double* d_x_new, hipStream_t s);
void launch_win_model(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_halo_step, hipStream_t s);
void launch_win_finish(const WinProblem& P, const WinBuf& W, const double* d_x, const double* d_beta, double* d_x_new,
                       double* d_beta_new, int mode, hipStream_t s);
void launch_win_accept(const WinProblem& P, const WinBuf& W, const double* d_r_new, double* d_x, double* d_beta,
                       const double* d_x_new, const double* d_beta_new, int mode, hipStream_t s);

// sharded solves (bodyfit_solve_sharded*): the exchange steps
void launch_sum_ranks(const double* d_g, int N, int stride, int n, double* d_out, hipStream_t s);
void launch_replicate_ranks(double* d_g, int n, int N, hipStream_t s);
int iface_doubles(int n_extra);
void launch_iface_pack(const WinBuf& W, int F, const double* d_extra, int n_extra, double* d_send, hipStream_t s);
void launch_iface_unpack(const WinBuf& Wi, const double* d_g, int N, int n_extra, double* d_extra_sum, hipStream_t s);
void launch_win_halo_step(const WinProblem& P, const double* d_Xi, const double* d_dsb, int node_right, const double* d_scale_right,
                          const double* d_x_right, double* d_d_right, double* d_xn_right, int node_left,
                          const double* d_scale_left, const double* d_x_left, double* d_xn_left, hipStream_t s);
void launch_win_fold_fail(const WinBuf& W, const WinBuf& Wi, hipStream_t s);
void launch_win_decide(const WinProblem& P, const WinBuf& W, double* d_x, double* d_beta, double* d_x_new, double* d_beta_new,
                       const double* d_g, int N, double* d_x_halo, const double* d_xn_halo, double* d_x_left,
                       const double* d_xn_left, hipStream_t s);

// f32 -> bf16 round-to-nearest-even (finite inputs)
__host__ __device__ inline uint16_t f32_to_bf16(float x) {
  union { float f; uint32_t u; } c;
  c.f = x;
  return (uint16_t)((c.u + 0x7FFFu + ((c.u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t b) {
  union { float f; uint32_t u; } c;
  c.u = ((uint32_t)b) << 16;
  return c.f;
}

// kernel launchers (defined in the .hip files)
void launch_frame_resjac(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta,
                         double* d_r, double* d_J, double* d_joints, const MeshCoef& mc, int want_jac,
                         const PriorArgs& priors, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_mesh(const DevModel& M, const DevProblem& P, const MeshCoef& mc, float* d_cloud, const PriorArgs& pa,
                 const double* d_params, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
bool role_sweep_fits(const DevModel& M, const DevProblem& P);
void launch_sweep_roles(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta, double* d_r,
                        double* d_J, double* d_joints, const MeshCoef& mc, int want_jac, const PriorArgs& pa, float* d_cloud,
                        const FusedSync& sy, const FoldTail& fold, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// (ev_start / ev_stop: optional events that take the dispatch's own begin / end timestamps, hipExtLaunchKernelGGL)
void launch_reduce_shared_ex(int K, int ncols, int npose, int nS, int total_rows, const double* d_r,
                             const double* d_J, double huber_delta, int shape_row0, int shape_rows,
                             double beta_shape, double* d_partials, double* d_out66, hipStream_t s);
int reduce_partials_doubles();
// the non-zero column blocks of every reprojection block's Jacobian, contiguous (bodyfit_api.hip: packed cache of the host path)
void launch_pack_jacobian(int K, int ncols, int n_joint_blocks, const double* d_J, const unsigned* d_mask, const unsigned* d_off,
                          double* d_out, const double* d_r, int nr, double* r_out, const int* d_comp, int ncomp, int* comp_out,
                          hipStream_t s);
void launch_reduce_frames(int F, const double* d_r, int shape_row0, int shape_rows,
                          double beta_shape, const double* d_frame_partials, double* d_scratch, double* d_out66,
                          hipStream_t s);
void launch_writeback_prepare(int F, int npose, const double* d_params, const double* d_R0, double* d_params_upd,
                              double* d_R0_new, hipStream_t s);
void launch_mean_pixel_error(int F, int nJ, const int* d_kp_offset, const int* d_kp_id, const double* d_kp_uv,
                             const double* d_joints, double fx, double fy, double cx, double cy, double* d_out, hipStream_t s);
void launch_regress(int nJ, int V, int ncol, const double* d_reg, const double* d_x, double* d_out, hipStream_t s);

}  // namespace bodyfit
