// solver_view.h — private view of a bodyfit_problem for the host solver (not part of the public ABI).
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
typedef struct bodyfit_solver_view {
  int n_frames, n_joints, n_shape;
  int beta_per_frame, has_gmm, temporal_halo;
  double beta_pose, beta_shape, lambda_temporal, huber_delta;
  int max_kp_per_frame;
  const int* kp_offset;     /* [F+1] host */
  const double* prec_cho;   /* [K][D][D] host, when has_gmm */
} bodyfit_solver_view;
int bodyfit_internal_solver_view(bodyfit_problem* p, bodyfit_solver_view* out);
int bodyfit_internal_fail(int code, const char* msg);
int bodyfit_internal_solve_batched_device(bodyfit_problem* p, double* frame_params, double* beta,
                                          const unsigned char* param_constant, const bodyfit_fit_options* opt,
                                          bodyfit_fit_summary* summaries, int n_summaries);   /* sets bodyfit_last_error(), returns code */
/* Device-resident LM for one shared-beta window (k_window_lm.hip). */
int bodyfit_internal_solve_window_device(bodyfit_problem* p, double* frame_params, double* beta,
                                         const unsigned char* param_constant, const bodyfit_fit_options* opt,
                                         bodyfit_fit_summary* summary, const bodyfit_comm* comm /* NULL: one GPU */);
/* Window solver: evaluate at (frame_params, beta) and return the residual vector, the GMM components and, per frame,
 * the reprojection part of the normal equations built on the device (k_frame_normal): H [F][87][88], lower triangle of
 * J^T rho' J over the frame's n columns, gradient J^T rho' r in row n.  Needs <= 32 keypoints per frame. */
int bodyfit_internal_frame_normals(bodyfit_problem* p, const double* frame_params, const double* beta, double* residuals,
                                   int* gmm_comp, double* H);
#ifdef __cplusplus
}
#endif
