/* bodyfit.h — C ABI of the MI355X-native SMPL residual/Jacobian evaluator (libbodyfit.so).
 *
 * Drop-in boundary for the hot path of jonH34400/3DBodyAnimation: everything the reference
 * executes inside ceres::Solve's iteration loop (CostFunction::Evaluate of its residual blocks)
 * plus the SMPL forward (ark::Avatar::update) those blocks are defined on.  Plain pointers and
 * sizes only; no C++/torch types.  Every entry point returns 0 on success and a non-zero
 * bodyfit_status otherwise (Ceres convention: Evaluate()==false marks an infeasible step, so a
 * HIP failure maps to "false", never to an exception).  bodyfit_last_error() gives the text.
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   bodyfit_model_*            ark::AvatarModel (external/avatar, absent) as used at
 *                              include/Sim3BA.h:360-372, include/MultiFrameBA.h:46-60;
 *                              tensors = SMPL npz keys (scripts/npz_fixer.py:4-17)
 *   bodyfit_gmm_*              ark::GaussianMixture (include/Sim3BA.h:249,257,266,280,288);
 *                              data format scripts/convert_gmm_to_avatar.py:14-29
 *   bodyfit_problem_create     the residual blocks added by include/Sim3BA.h:410-470,572-638 and
 *                              include/MultiFrameBA.h:71-142 (PixelKP list include/Sim3BA.h:9)
 *   bodyfit_evaluate_batch     one sweep of CostFunction::Evaluate over all those blocks:
 *                              ReprojCost / ReprojCostShape (include/Sim3BA.h:34-88,126-227) with an
 *                              analytic Jacobian in place of DynamicAutoDiffCostFunction (:420,581),
 *                              PosePriorAAAnalytic (:263-315), ShapePriorL2Analytic (:331-343),
 *                              Vec3DiffCost (include/MultiFrameBA.h:20-28)
 *   bodyfit_evaluate_block     ceres::CostFunction::Evaluate(parameters, residuals, jacobians)
 *                              for ONE block, same null conventions (include/Sim3BA.h:263-264)
 *   bodyfit_forward            ark::Avatar::update() (include/Sim3BA.h:371,538;
 *                              include/MultiFrameBA.h:53,173; src/main_single_frame.cpp:213,254)
 *   bodyfit_mean_pixel_error   mean_pixel_error (include/Utils.h:102-115)
 *   bodyfit_writeback_batch    the post-solve write-back loops (include/MultiFrameBA.h:154-174,
 *                              include/Sim3BA.h:481-505) + mean_pixel_error, for all frames at once
 *   bodyfit_reduce_shared_device  the shared shape block of OptimizeMultiFrame (include/MultiFrameBA.h:67-68)
 *                              reduced per GPU for frame-sharded solves
 *   bodyfit_frame_normals      per-frame normal equations of the reprojection blocks (what DENSE_QR
 *                              factors, include/MultiFrameBA.h:145-151), for structured window solvers
 *   bodyfit_solve              ceres::Solve as configured by OptimizePoseReprojection /
 *                              OptimizePoseShapeReprojection (include/Sim3BA.h:472-479,641-647) and
 *                              OptimizeMultiFrame (include/MultiFrameBA.h:144-151); the three functions
 *                              themselves are mirrored in include/bodyfit.hpp
 */
/* Environment variables the library reads.  None is needed in production; none changes a result (the A/B forms are tested
 * bit-identical to the default), they select between equivalent code paths for measurements and tests:
 *   BODYFIT_ONE_LAUNCH=0     problems created afterwards sweep as two launches (k_frame_resjac, k_mesh_blend_lbs) instead of one
 *   BODYFIT_LM_PLAIN=1       the batched LM in its four-launch form (step, residual sweep, accept, Jacobian sweep)
 *   BODYFIT_PACKED_J=0       bodyfit_evaluate_batch's cache keeps the dense Jacobian panel instead of the packed blocks
 *   BODYFIT_PACK_DIRECT=0    the packed Jacobian goes down by hipMemcpyAsync instead of by the packing kernel's own stores
 *   BODYFIT_FORCE_SHARDED=1  a one-rank communicator still takes the sharded code path (how RCCL is exercised on a one-GPU box)
 *   BODYFIT_WINDOW_MIN=n     shared-beta windows shorter than n frames iterate on the host (default 12)
 *   BODYFIT_HOST_NORMALS=1   the host loop builds its normal equations on the host; BODYFIT_TIMING=1 prints its phase times
 * The tuning words of the one-launch sweep (mesh priority, operand pacing, Jacobian store scope) are compile-time constants since
 * round 5 (csrc/bodyfit_device.h kTune*); round 4 read them from BODYFIT_MESH_PRIO / BODYFIT_TRICKLE_* / BODYFIT_J_SCOPE. */
#ifndef BODYFIT_H_
#define BODYFIT_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bodyfit_status {
  BODYFIT_OK = 0,
  BODYFIT_ERR_INVALID = 1,   /* bad argument / shape                                  */
  BODYFIT_ERR_HIP = 2,       /* a HIP runtime call failed (no device, OOM, fault ...) */
  BODYFIT_ERR_NUMERIC = 3    /* non-SPD covariance etc.                               */
} bodyfit_status;

typedef struct bodyfit_model bodyfit_model;     /* device-resident SMPL model            */
typedef struct bodyfit_gmm bodyfit_gmm;         /* device-resident max-mixture pose prior */
typedef struct bodyfit_problem bodyfit_problem; /* residual blocks of one solve           */

/* SMPL tensors, host pointers, row-major f64 (the reference keeps Eigen doubles).          */
typedef struct bodyfit_model_desc {
  int n_verts;               /* 6890 */
  int n_joints;              /* 24   */
  int n_shape;               /* 10   */
  int n_pose_feat;           /* 207 = 9 (n_joints-1); 0 disables pose-corrective blendshapes */
  const double* v_template;  /* [n_verts][3]              */
  const double* shapedirs;   /* [n_verts][3][n_shape]     */
  const double* posedirs;    /* [n_verts][3][n_pose_feat] or NULL */
  const double* j_regressor; /* [n_joints][n_verts]       */
  const double* weights;     /* [n_verts][n_joints]       */
  const int* parent;         /* [n_joints], root = -1 (scripts/npz_fixer.py) */
  int n_landmarks;           /* vertex-landmark keypoints (0 = none) */
  const int* landmark_vid;   /* [n_landmarks] vertex ids  */
  /* Sparse keypoint regressors over the POSED vertices (0 = none): keypoint id n_joints + n_landmarks + r is
   *   sum_i kpreg_weight[i] * posed_vertex(kpreg_vid[i]),  i in [kpreg_offset[r], kpreg_offset[r + 1])
   * (OpenPose-style extra keypoints: a weighted mean of a few surface vertices; the avatar library's keypoints of this
   * kind are what the reference's PixelKP::jid would address past the 24 SMPL joints, include/Sim3BA.h:28-33).
   * Evaluated exactly, with its Jacobian, in the frame kernel: per skinning joint j the row collapses to ONE pseudo-vertex
   * (sum_i w_i W_ij [v_i ; 1] is linear in the vertex rows), so a row costs as many landmark slots as its vertices have
   * distinct joints.  Landmarks + slots of all rows <= 32.                                                               */
  int n_kp_regressors;
  const int* kpreg_offset;    /* [n_kp_regressors + 1] */
  const int* kpreg_vid;       /* [nnz] vertex ids      */
  const double* kpreg_weight; /* [nnz]                 */
} bodyfit_model_desc;

int bodyfit_model_create(const bodyfit_model_desc* desc, int device, bodyfit_model** out);
void bodyfit_model_destroy(bodyfit_model* m);
/* initialJointPos [nJ][3], jointShapeReg [3 nJ][nS], rest offsets [nJ][3] (include/Sim3BA.h:372-392),
 * regressed on the device at create time.  Any pointer may be NULL.                         */
int bodyfit_model_get_derived(const bodyfit_model* m, double* joints0, double* joint_shape_reg,
                              double* offset);

/* pose_prior.txt contents: K weights, K x D means, K x D x D covariances (row-major).
 * resid_scale: factor on L_k^T (x - mu_k) inside residual(); sqrt(0.5) = recalled upstream.  */
int bodyfit_gmm_create(int n_comp, int dim, const double* weights, const double* means,
                       const double* covs, double resid_scale, int device, bodyfit_gmm** out);
void bodyfit_gmm_destroy(bodyfit_gmm* g);
/* prec_cho [K][D][D] (lower L, precision = L L^T) and -log of the normalised weights [K]. */
int bodyfit_gmm_get(const bodyfit_gmm* g, double* prec_cho, double* neg_log_w);

/* Frame parameter packing (order is load-bearing, include/Sim3BA.h:36-40,421-430):
 *   x[76] = [scale, rootAA(3), rootT(3), jointAA[1](3) ... jointAA[23](3)]                  */
#define BODYFIT_FRAME_PARAMS 76

typedef struct bodyfit_problem_desc {
  int n_frames;
  const int* kp_offset;   /* [n_frames+1] CSR over keypoints (frames may be empty)            */
  const int* kp_id;       /* [K] id < n_joints: SMPL joint (PixelKP::jid);
                                 n_joints <= id < n_joints + n_landmarks: vertex landmark id - n_joints;
                                 above: keypoint regressor row id - n_joints - n_landmarks    */
  const double* kp_uv;    /* [K][2] observed pixels (PixelKP::u,v)                            */
  double fx, fy, cx, cy;
  const double* R0;       /* [n_frames][9] row-major fixed root orientation (avatar.r[0])     */
  int n_cols;             /* 76: blocks of ReprojCost; 76+n_shape: + the shape block          */
  int use_shape;          /* jointShapeReg handed to the functor (betaShape > 0); else the
                             shape block, if present, gets zero columns (MultiFrameBA.h:88)   */
  int beta_per_frame;     /* 0: one shared beta[n_shape]; 1: beta[n_frames][n_shape]          */
  int pose_blend;         /* apply posedirs in vertex landmarks and the mesh                  */
  double beta_pose;       /* PosePriorAAAnalytic weight; 0 = no prior block                   */
  const bodyfit_gmm* gmm; /* NULL = L2 fallback (include/Sim3BA.h:283)                        */
  double beta_shape;      /* ShapePriorL2Analytic weight; 0 = none                            */
  double lambda_temporal; /* Vec3DiffCost weight between frames f, f+1; 0 = none              */
  int temporal_halo;      /* 1: frame_params holds n_frames+1 rows; the last row is the next
                             shard's first frame and only feeds the last temporal block       */
  double huber_delta;     /* HuberLoss on reprojection blocks (3.0 in the reference)          */
  int want_mesh;          /* also produce the 6890-vertex cloud per frame on each evaluation  */
} bodyfit_problem_desc;

int bodyfit_problem_create(const bodyfit_model* m, const bodyfit_problem_desc* desc, bodyfit_problem** out);
void bodyfit_problem_destroy(bodyfit_problem* p);

/* Row layout of the batched residual vector (bodyfit_problem_layout):
 *   [reproj: 2 per keypoint, frame-major][pose prior: n_prior_rows per frame]
 *   [shape prior: n_shape per beta][temporal: 75 per adjacent pair: rootT, rootAA, joints 1..23]
 * The reprojection Jacobian is a dense row-major [2K][n_cols] panel (columns = the frame's 76
 * parameters, then beta).  Prior/temporal Jacobians are constant (beta I, beta_p L_k^T, +-lambda I)
 * and are reproduced by bodyfit_evaluate_block / the host solver from gmm_comp.               */
typedef struct bodyfit_layout {
  int n_keypoints;      /* K                                  */
  int n_cols;
  int reproj_rows;      /* 2 K                                */
  int prior_rows_per_frame; /* 0, 69 (L2) or 70 (GMM)         */
  int shape_rows;       /* 0, nS or F nS                      */
  int temporal_rows;    /* 0 or 75 (F-1 [+1 with halo])       */
  int total_rows;
} bodyfit_layout;
int bodyfit_problem_layout(const bodyfit_problem* p, bodyfit_layout* out);

/* Host-pointer form: H2D of the parameters, one device sweep, D2H of the results.
 *   frame_params [F(+1)][76], beta [nS] or [F][nS] (NULL if n_cols == 76)
 *   residuals [total_rows]; jacobian [2K][n_cols] or NULL (want_jacobian = 0);
 *   gmm_comp [F] or NULL: selected mixture component per frame.                               */
int bodyfit_evaluate_batch(bodyfit_problem* p, const double* frame_params, const double* beta,
                           double* residuals, double* jacobian, int* gmm_comp, int want_jacobian);

/* Device-pointer form (inputs already resident in HBM, asynchronous on `stream`, a hipStream_t
 * passed as void*; NULL = the default stream).  Results stay in the problem's device buffers.
 * One problem, one sweep at a time: successive calls must be ordered (same stream, or events); the synchronous entry points
 * (bodyfit_evaluate_batch, bodyfit_forward, bodyfit_writeback_batch, bodyfit_frame_normals) order themselves behind whatever
 * was enqueued through this call (they run on a private stream / the NULL stream).
 * Errors of an asynchronous sweep.  With want_mesh the sweep is ONE launch whose mesh workgroups wait, inside the launch, for
 * operands the frame workgroups publish (k_sweep.hip).  Every such wait is bounded; a mesh workgroup whose wait runs out leaves
 * its 32-vertex tile of the cloud unwritten and sets an error word.  Residuals, Jacobian, joints, GMM components and the
 * shared reduction (bodyfit_reduce_shared_device / bodyfit_arm_shared_reduction) never depend on a wait and are complete
 * regardless.  The synchronous entry points notice the word and re-issue their sweep as two launches themselves; a caller of
 * THIS function learns of it from bodyfit_sweep_status, which it should call before it consumes the cloud.             */
int bodyfit_evaluate_device(bodyfit_problem* p, const double* d_frame_params, const double* d_beta,
                            int want_jacobian, void* stream);
/* Waits for `stream`, then reports whether every asynchronous sweep of the problem since the last check completed:
 * BODYFIT_OK, or BODYFIT_ERR_HIP ("... timed out"): the cloud of at least one of them is incomplete; the problem uses the
 * two-launch sweep from then on, so re-issuing the evaluation gives the complete result.                                */
int bodyfit_sweep_status(bodyfit_problem* p, void* stream);
/* One-launch sweeps of this problem found incomplete (an in-launch wait ran out) since it was created, whoever noticed: a
 * synchronous entry point that re-issued its sweep, or bodyfit_sweep_status.  0 in every healthy run; benchmarks report it. */
long bodyfit_sweep_timeouts(const bodyfit_problem* p);
/* Lifetime of `stream`: the synchronous entry points and the device solves order themselves behind the last asynchronous call
 * by recording an event ON THAT STREAM, so it must stay alive until the problem's next synchronous call (or
 * bodyfit_sweep_status on it) has returned. */

typedef struct bodyfit_device_views {
  double* residuals;    /* [total_rows]          */
  double* jacobian;     /* [2K][n_cols]          */
  int* gmm_comp;        /* [F]                   */
  float* cloud;         /* [F][cloud_frame_stride] f32, camera frame (want_mesh): frame f's
                           [n_verts][3] block starts at cloud + f * cloud_frame_stride          */
  double* joints;       /* [F][n_joints][3] camera-frame posed joints             */
  double* normal_eq;    /* [66] see bodyfit_reduce_shared_device                  */
  long long cloud_frame_stride; /* floats; 3 n_verts rounded up to whole 32-vertex tiles (a multiple of
                           128 bytes, so that every wave of the mesh kernel stores whole cache lines)     */
} bodyfit_device_views;
int bodyfit_problem_views(bodyfit_problem* p, bodyfit_device_views* out);

/* Shared-shape reduction for frame-sharded solves: after an evaluation, reduce over the local
 * frames  out[0] = cost = sum 1/2 rho(|r|^2) (+ 1/2 |r|^2 of the prior/temporal rows),
 *         out[1..10] = g_beta = sum J_beta^T (rho' r),  out[11..65] = upper(H_bb) = sum rho' J_beta^T J_beta.
 * d_out66 (device, 66 doubles) is what the caller all-reduces across ranks (RCCL).            */
int bodyfit_reduce_shared_device(bodyfit_problem* p, double* d_out66, void* stream);
/* Arm the reduction: every following Jacobian sweep of this problem (want_jacobian, want_mesh, shared beta, at most
 * 256 frames + prior tiles: a shard of a sharded window) deposits the 66 doubles in d_out66 at ITS OWN TAIL — the last
 * frame workgroup to finish sums the per-frame partials while the mesh workgroups are still running — and
 * bodyfit_reduce_shared_device(p, d_out66, stream) on the same stream then launches nothing.  Sweeps that cannot fold
 * (two-launch sweep, longer shards) leave the work to bodyfit_reduce_shared_device as before; the numbers are
 * bit-identical either way.  d_out66 = NULL disarms.                                                             */
int bodyfit_arm_shared_reduction(bodyfit_problem* p, double* d_out66);
/* (A timed-out in-launch wait — see bodyfit_evaluate_device — does not touch the 66 doubles: the frame and prior workgroups
 *  that produce them wait for nothing they need.)                                                                  */

/* Measurement aid: `iters` sweeps with HIP events around every kernel on `stream`;
 * avg_ms[5] = average launch duration (ms) of {frame_resjac, 0 (the prior workgroups ride on another launch),
 * mesh_blend_lbs, reduce_shared, sweep_roles}.  A sweep with the mesh on is ONE launch (sweep_roles: frame, mesh and prior
 * workgroups side by side); then entries 0 and 2 are 0.  Without the mesh, or with BODYFIT_ONE_LAUNCH=0, entry 4 is 0.
 * Entry 3 is ~0 when the reduction rode on the sweep's own tail (bodyfit_arm_shared_reduction). */
int bodyfit_profile_sweep(bodyfit_problem* p, const double* d_frame_params, const double* d_beta,
                          int want_jacobian, int with_reduce, int iters, void* stream, double* avg_ms);

/* ceres::CostFunction::Evaluate for ONE residual block, Ceres pointer conventions
 * (jacobians may be NULL; jacobians[b] may be NULL; blocks row-major num_residuals x size).
 *   kind 0: reprojection block `index` (keypoint), parameters = 26 or 27 blocks
 *   kind 1: pose prior of frame `index`, parameters = 23 blocks of 3
 *   kind 2: shape prior (index = frame when beta_per_frame), parameters = 1 block of nS
 *   kind 3: temporal link `index` = 25*pair + slot, parameters = 2 blocks of 3                */
int bodyfit_evaluate_block(bodyfit_problem* p, int kind, int index, const double* const* parameters,
                           double* residuals, double** jacobians);

/* The same, for callers that follow ceres::EvaluationCallback (bodyfit_ceres::SweepCallback): the cached sweep is trusted to be
 * the point under evaluation, so reprojection / pose-prior blocks are served without comparing their parameters with the cache
 * and without a lock (Ceres evaluates blocks on several threads).  Fails if no sweep is cached.                           */
int bodyfit_evaluate_block_cached(bodyfit_problem* p, int kind, int index, const double* const* parameters,
                                  double* residuals, double** jacobians);

/* ark::Avatar::update(): camera-frame joints [F][nJ][3] (f64) and cloud [F][V][3] (f32) for the
 * problem's frames at the given parameters; either output may be NULL.                       */
int bodyfit_forward(bodyfit_problem* p, const double* frame_params, const double* beta,
                    double* joints, float* cloud);

/* The post-solve write-back of a whole solve on the device: for every frame
 *   r[0] <- R(rootAA) r[0]  (left-multiplied, so it compounds over repeated solves),  p <- rootT,
 *   r[j] <- R(jointAA[j]),  Avatar::update()  (the Sim3 scale is dropped),
 * as OptimizeMultiFrame / OptimizePose*Reprojection do per frame on the host (include/MultiFrameBA.h:154-174,
 * include/Sim3BA.h:481-505), followed by mean_pixel_error (include/Utils.h:102-115) of the frame's FK-joint
 * keypoints against the updated joints (0 for a frame without any).
 * Outputs, each optional: R0_out [F][9] row-major, joints [F][nJ][3], cloud [F][V][3] (needs want_mesh; the posed
 * cloud also stays resident, see bodyfit_problem_views), mean_px [F].                                      */
int bodyfit_writeback_batch(bodyfit_problem* p, const double* frame_params, const double* beta, double* R0_out,
                            double* joints, float* cloud, double* mean_px);

/* include/Utils.h:102-115 on the joints of bodyfit_forward (no Sim3 scale: pass scale = 1). */
double bodyfit_mean_pixel_error(int n_kp, const int* jid, const double* uv, const double* joints,
                                double fx, double fy, double cx, double cy);

/* ---- outer loop (what the reference hands to ceres::Solve) ------------------------------------
 * Ceres-like trust-region Levenberg-Marquardt over one bodyfit_problem (see host_solver.cpp for the
 * restated algorithm).  Every evaluation is one device sweep; the linear solve is a block-tridiagonal
 * Cholesky with a Schur complement on the shared shape block: on the device by block cyclic reduction over the
 * frames (k_window_lm.hip), for short windows on the host (host_solver.cpp).
 *   frame_params [F][76] in/out, beta [nS] or [F][nS] in/out (NULL when n_cols == 76)
 *   param_constant [76] flags or NULL: 1 = SetParameterBlockConstant (include/Sim3BA.h:608-611)
 *   independent_frames 1: every frame is its own problem with its own LM state (3dba_single: frames
 *   are fitted independently, src/main_single_frame.cpp:192); 0: one problem over all frames
 *   (OptimizeMultiFrame).  summaries: one per problem (F or 1).                                  */
typedef struct bodyfit_fit_options {
  int max_iters;            /* ceres::Solver::Options::max_num_iterations */
  double scale_lo, scale_hi;/* SetParameterLowerBound / UpperBound on the scale: 0.3, 3.0 */
  int verbose;
  int solver;               /* 0 auto: independent frames iterate on the device (k_lm_batched); a shared-beta window
                               of >= 12 frames iterates on the device too (k_window_lm: block cyclic reduction over the
                               frames), shorter ones on the host; 1 force the host loop; 2 force the device loop for
                               independent frames; 3 force the device loop for a shared-beta window */
} bodyfit_fit_options;
typedef struct bodyfit_fit_summary {
  int iterations;           /* LM iterations (successful + unsuccessful) */
  int termination;          /* 0 convergence, 1 iteration limit, 2 failure */
  int usable;               /* Summary::IsSolutionUsable() */
  int n_successful, n_unsuccessful;
  int n_sweeps;             /* device evaluations the solve needed.  Device window LM: 1 + one per iteration (the sweep at a
                               candidate also leaves its Jacobian); host loop and the batched LM in its four-launch form: 1 + one
                               per iteration + one per accepted step */
  double initial_cost, final_cost;
  int n_sweeps_issued;      /* ... and the sweeps actually launched: the device window LM runs up to three iterations ahead of
                               the host's status reads (they find the solve terminated and change nothing); 0 = same as n_sweeps */
} bodyfit_fit_summary;
int bodyfit_solve(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                  int independent_frames, const bodyfit_fit_options* options, bodyfit_fit_summary* summaries,
                  int n_summaries);

/* ---- one window sharded over several GPUs (frames = the data-parallel axis of OptimizeMultiFrame) -----------
 * One process per GPU.  Rank r creates a bodyfit_problem over ITS contiguous frames (temporal_halo = 1 on every rank but
 * the last: the temporal pair that leaves the shard is evaluated by the shard it leaves; beta_shape > 0 on exactly one
 * rank) and calls bodyfit_solve_sharded with a communicator.  The LM state is replicated by construction (every rank
 * takes the same decisions from the same reduced scalars); the linear solve is substructured: cyclic reduction of the
 * local chain down to its two end frames, an all-gather of those 2 N interface blocks, the interface system solved by
 * every rank, local back-substitution.  The shared-shape terms [H_bb, g_beta] cross the ranks ONCE per LM iteration.
 * Per LM iteration the ranks exchange THREE all-gathers (the 2 N interface blocks with the beta terms riding on them, 225 KB
 * per rank; the beta Schur partials, 110 doubles; six scalars) and every rank sums the gathered partials in rank order, so the
 * decisions are identical everywhere without a broadcast.  Two transports:
 *   bodyfit_solve_sharded        the caller's callback on HOST buffers (device -> host -> callback -> device; torch.distributed
 *                                "gloo" in the tests, MPI in a C++ host).  Only `allgather` is called; `allreduce` may be NULL.
 *   bodyfit_solve_sharded_rccl   RCCL on the solve's device buffers and stream (ncclAllGather over xGMI): no host staging and no
 *                                stream synchronisation between the host's status reads (every fourth iteration).
 * Both callbacks / RCCL calls must be entered by every rank of the communicator the same number of times.  Failures:
 *   - a rank whose own device work fails (the first sweep, or a kernel launch / HIP call inside an LM iteration) does NOT leave on its own: it marks
 *     its scalars, keeps taking part in that iteration's exchanges, and the decision kernel ends the solve on EVERY rank in the
 *     same iteration; all ranks then return an error (the failing one its own, the others "another rank reported a device
 *     failure") after the same number of exchanges — nobody is left waiting;
 *   - a failure of the transport itself (a callback that returns non-zero, an RCCL error) returns at once on the rank that saw
 *     it; its peers may be waiting in that exchange: bodyfit_set_exchange_timeout (below) bounds that wait inside the library,
 *     whatever timeout the process group / RCCL has of its own.
 *   frame_params [F_local (+1 halo row)][76] in/out: the halo row is refreshed from the neighbour by the solve. */
/* A bound, in seconds, on every exchange and every status read of this problem's sharded solves (0, the default: none).  With it
 * a transport failure cannot strand the peers of the rank that saw it: a host callback that has not returned within the bound
 * (bodyfit_solve_sharded; the callback then runs on a helper thread, which is left behind with its own copies of the buffers) or a
 * stream that has not drained (an RCCL collective its peer never entered) ends the solve with BODYFIT_ERR_HIP on that rank too.
 * The problem's solve stream is not usable after such a return.  An abandoned callback keeps running on its helper thread: its
 * `ctx` must stay valid until it returns; and bodyfit_problem_destroy waits for the device, so with the RCCL transport abort or
 * destroy the communicator first (a collective that will never complete would hold that wait too).  Choose the bound well above
 * one LM iteration (milliseconds); it is a liveness guard, not a pacing device. */
int bodyfit_set_exchange_timeout(bodyfit_problem* p, double seconds);
typedef struct bodyfit_comm {
  int rank, size;
  void* ctx;
  int (*allreduce)(void* ctx, double* buf, int n, int op /* 0 sum, 1 max */);            /* unused since round 3; may be NULL */
  int (*allgather)(void* ctx, const double* send, double* recv /* [size][n] */, int n);
} bodyfit_comm;
int bodyfit_solve_sharded(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                          const bodyfit_comm* comm, const bodyfit_fit_options* options, bodyfit_fit_summary* summary);

/* RCCL communicator of the sharded solve.  librccl is bound at run time (dlopen), so libbodyfit.so has no link-time dependency
 * on it.  Either let the library create the communicator — rank 0 obtains the 128-byte id (ncclGetUniqueId) and ships it to the
 * other ranks by any host channel, every rank then calls bodyfit_rccl_create (ncclCommInitRank, collective) — or hand in an
 * ncclComm_t the application already has (bodyfit_rccl_wrap; not destroyed by bodyfit_rccl_destroy).                       */
typedef struct bodyfit_rccl bodyfit_rccl;
int bodyfit_rccl_unique_id(unsigned char* id128);
int bodyfit_rccl_create(const unsigned char* id128, int rank, int size, int device, bodyfit_rccl** out);
int bodyfit_rccl_wrap(void* nccl_comm, int rank, int size, bodyfit_rccl** out);
void bodyfit_rccl_destroy(bodyfit_rccl* c);
int bodyfit_solve_sharded_rccl(bodyfit_problem* p, double* frame_params, double* beta, const unsigned char* param_constant,
                               bodyfit_rccl* comm, const bodyfit_fit_options* options, bodyfit_fit_summary* summary);
/* The evaluation path's one collective (include/MultiFrameBA.h:64-68 — the shared shape block — summed over the
 * shards): ncclAllReduce(sum, f64) of the 66 doubles [cost | g_beta | upper H_bb], in place on the device buffer, on `stream`:
 * behind bodyfit_evaluate_device + bodyfit_reduce_shared_device on the same stream it needs no host synchronisation.    */
int bodyfit_allreduce_shared_rccl(bodyfit_rccl* comm, double* d_buf66, void* stream);
/* ranks of the communicator and this process's rank, as RCCL reports them (ncclCommCount, ncclCommUserRank) */
int bodyfit_rccl_count(bodyfit_rccl* comm, int* n_ranks, int* rank);
/* Measurement aid for boxes with ONE GPU: this problem's following sharded solves through a one-rank communicator run as rank
 * `rank` of `n_ranks` IDENTICAL shards — the local chain reduced with its ends pinned, the interface chain of 2 n_ranks frames,
 * every all-gather issued (on the one-rank communicator; a small kernel then fills the other n_ranks - 1 gathered slots with
 * copies of this shard's), sums over n_ranks slots, the neighbours' boundary steps.  The problem must be shaped like that rank's
 * shard (temporal_halo = 1 unless rank == n_ranks - 1).  What it times is ONE rank's critical path at that geometry with the
 * transport's latency at its lower bound; the fitted numbers belong to a window of n_ranks copies of the shard, not to the
 * caller's sequence.  n_ranks <= 1 switches it off.  bench.py's c5_strong.shard_proxy uses it. */
int bodyfit_set_shard_proxy(bodyfit_problem* p, int n_ranks, int rank);
/* all-gathers issued by the problem's last sharded solve (tests assert the number of exchanges per iteration) */
long bodyfit_last_exchange_count(const bodyfit_problem* p);

/* Normal equations of the reprojection blocks, built on the device (window solvers: bodyfit_solve's host loop,
 * the numpy cross-check tests/sharded_lm_check.py): evaluate at (frame_params, beta) and return the residual vector [total_rows], the
 * GMM components [F] (may be NULL) and, per frame, the lower triangle of J^T rho' J over its n_cols columns with the
 * gradient J^T rho' r in row n_cols, as a [F][87][88] row-major panel (HuberLoss weights rho' applied per keypoint,
 * include/MultiFrameBA.h:64,102).  Prior and temporal blocks have constant Jacobians and are left to the caller.
 * Needs <= 32 keypoints per frame.                                                                        */
int bodyfit_frame_normals(bodyfit_problem* p, const double* frame_params, const double* beta, double* residuals,
                          int* gmm_comp, double* normals);

/* ---- mesh overlay ------------------------------------------------------------------------
 * smpl::render::renderSMPLMesh(cloud, faces, img, fx, fy, cx, cy, fill, backface_cull, wireframe) of
 * include/RenderSMPLMesh.h:16-110, batched over frames with the posed vertices and the 8-bit BGR images
 * resident on the device: project (:36-46), per-face cull / flat shade / painter depth / integer corners
 * (:50-88), far-to-near order (:91-92; ties by face index, the reference's std::sort leaves them unspecified)
 * and cv::fillConvexPoly(..., LINE_AA) per triangle in that order (:95-104).  The result is, pixel for pixel,
 * what drawing the triangles one after the other gives.  wireframe != 0 adds cv::polylines in gray 40 after each
 * triangle's fill (:106-109; the reference's callers never enable it: src/main_single_frame.cpp:274,
 * src/main_multi_frame.cpp:210,223); fill == 0 and wireframe == 0 draws nothing, as there.
 *   faces [n_faces][3] vertex ids (ark::AvatarModel::mesh columns, src/main_single_frame.cpp:185-188)
 *   cloud: x, y, z per vertex (the memory order of the reference's 3xN column-major `cloud`), camera
 *   coordinates; cloud_is_f64 0: float (bodyfit_device_views.cloud of a write-back), 1: double
 *   images: n_frames images of height x width x 3 bytes, row_stride / frame_stride in bytes, modified in place
 *   (the reference draws into a clone of the video frame)                                                    */
typedef struct bodyfit_overlay bodyfit_overlay;
typedef struct bodyfit_overlay_desc {
  int device;
  int n_vertices, n_faces;
  const int32_t* faces;
  int width, height;
  int max_frames;
} bodyfit_overlay_desc;
int bodyfit_overlay_create(const bodyfit_overlay_desc* desc, bodyfit_overlay** out);
void bodyfit_overlay_destroy(bodyfit_overlay* ov);
/* device pointers; asynchronous on `stream` except for one 8-byte read-back that sizes the tile lists */
int bodyfit_overlay_render_device(bodyfit_overlay* ov, const void* d_cloud, int cloud_is_f64,
                                  size_t cloud_frame_stride_elems, int n_frames, uint8_t* d_images,
                                  size_t row_stride, size_t frame_stride, double fx, double fy, double cx,
                                  double cy, int fill, int backface_cull, int wireframe, void* stream);
/* host pointers (upload, render, download): the one-call form of the reference function */
int bodyfit_overlay_render(bodyfit_overlay* ov, const void* cloud, int cloud_is_f64, size_t cloud_frame_stride_elems,
                           int n_frames, uint8_t* images, size_t row_stride, size_t frame_stride, double fx,
                           double fy, double cx, double cy, int fill, int backface_cull, int wireframe);
/* draw list of `frame` from the latest render (far to near): face ids [n], corners [n][6] = x0 y0 x1 y1 x2 y2,
 * gray levels [n]; each array may be NULL; returns the count in *n_items                                   */
int bodyfit_overlay_drawlist(bodyfit_overlay* ov, int frame, int* n_items, int32_t* face, int32_t* corners,
                             int32_t* gray);
/* average launch durations (ms) of the overlay kernels of the latest render_device call, by HIP events:
 * [0] faces, [1] sort + rank, [2] binning (count, scan, fill), [3] tiles                                  */
int bodyfit_overlay_last_timing(bodyfit_overlay* ov, float ms[4]);

/* kernels launched by this process through the library so far (benchmarks: launches per LM iteration) */
long bodyfit_launch_count(void);
const char* bodyfit_last_error(void);
int bodyfit_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* BODYFIT_H_ */
