// The C++ snippets of INTEGRATION.md, verbatim, inside functions that declare what the surrounding reference code would
// provide.  tests/test_docs.py compiles this file (g++ -fsyntax-only against include/, the Ceres interface double and the
// MPI declarations double) and checks that every ```cpp block of INTEGRATION.md is the text between a pair of
// [snippet:NAME] / [/snippet] markers here — so a snippet that does not compile, or that drifts from this file, fails the
// CPU suite.
#include <array>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include <mpi.h>           // tests/cpp/mpi_double
#include <bodyfit.hpp>
#include <bodyfit_ceres.h>

using namespace bodyfit;

// ---- A. entry points ---------------------------------------------------------------------------------------------------
void snippet_entry_points(const double* v_template, const double* shapedirs, const double* posedirs, const double* J_regressor,
                          const double* weights, const int* parent, std::vector<Avatar*>& anchorAv,
                          std::vector<std::vector<PixelKP>>& kps_anchor, double fx, double fy, double cx, double cy,
                          std::vector<int>& valid_ids, std::vector<FramePoseParams>& anchorPos, double betaPose, double betaShape,
                          double lambdaT, int max_iters_s1) {
// [snippet:entry_points]
// src/main_multi_frame.cpp — replace
//   #include "MultiFrameBA.h"      by   #include <bodyfit.hpp>   and   using namespace bodyfit;
// and build the model from the arrays of model.npz instead of ark::AvatarModel(smpl_path):
bodyfit_model_desc md{6890, 24, 10, 207, v_template, shapedirs, posedirs, J_regressor, weights, parent /* root = -1,
                      scripts/npz_fixer.py */, 0, nullptr};
bodyfit::AvatarModel model_av(md, /*device=*/0);
// ...
auto [ok1, rep1] = OptimizeMultiFrame(model_av, anchorAv, kps_anchor, fx, fy, cx, cy, valid_ids, anchorPos,
                                      betaPose, betaShape, lambdaT, max_iters_s1);   // unchanged call
// [/snippet]
  (void)ok1; (void)rep1;
}

// ---- B. Ceres kept as the outer loop -----------------------------------------------------------------------------------
void snippet_ceres(bodyfit_problem* bp, const int* kp_offset, std::vector<FramePoseParams>& frame_params, double* beta,
                   ceres::Solver::Options& options, ceres::Solver::Summary& summary) {
// [snippet:ceres]
// #include <bodyfit_ceres.h>
// include/MultiFrameBA.h:85-142 (the four AddResidualBlock loops) become
ceres::Problem problem;
// the reference's own parameter memory: one FramePoseParams per frame (include/MultiFrameBA.h:9-14: NOT contiguous)
const bodyfit_ceres::BlockTable blocks = bodyfit_ceres::BlocksOf(frame_params);
bodyfit_ceres::AddOptions add;
add.with_callback = true;                                  // blocks trust the sweep the callback below has cached
bodyfit_ceres::AddResidualBlocks(&problem, bp, kp_offset, blocks, beta, add);
bodyfit_ceres::SweepCallback sweep(bp, blocks, beta);      // gathers the blocks into its own packed [F][76] buffer
options.evaluation_callback = &sweep;                      // Ceres 1.14: Solver::Options; 2.x: Problem::Options
// SetParameterBlockConstant / SetParameterLowerBound etc. stay as in include/Sim3BA.h:598-611 (same blocks)
ceres::Solve(options, &problem, &summary);
// [/snippet]
}

// ---- multi-GPU: one evaluation of a frame shard, the all-reduce issued by the library ----------------------------------------
// [snippet:sharded_eval]
// one evaluation of the shard on `stream`: sweep, the 66 doubles [cost | g_beta | upper H_bb] of the local frames, their sum over
// the ranks — include/MultiFrameBA.h:64-68 (the shared shape block) seen from N GPUs; nothing synchronises the host
int evaluate_shared(bodyfit_problem* shard, bodyfit_rccl* rc, const double* d_frame_params, const double* d_beta,
                    double* d_buf66 /* armed once: bodyfit_arm_shared_reduction(shard, d_buf66) */, void* stream) {
  if (bodyfit_evaluate_device(shard, d_frame_params, d_beta, /*want_jacobian=*/1, stream)) return 1;
  if (bodyfit_reduce_shared_device(shard, d_buf66, stream)) return 1;   // (launches nothing when the sweep's own tail did it)
  return bodyfit_allreduce_shared_rccl(rc, d_buf66, stream);            // ncclAllReduce(sum, f64), in place, on `stream`
}   // before the cloud of these sweeps is consumed: bodyfit_sweep_status(shard, stream)
// [/snippet]

// ---- multi-GPU: one window sharded over the ranks, MPI as the host transport ---------------------------------------------
// [snippet:sharded_mpi]
static int ar(void*, double* buf, int n, int op) {
  return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, op ? MPI_MAX : MPI_SUM, MPI_COMM_WORLD) != MPI_SUCCESS; }
static int ag(void*, const double* send, double* recv, int n) {
  return MPI_Allgather(send, n, MPI_DOUBLE, recv, n, MPI_DOUBLE, MPI_COMM_WORLD) != MPI_SUCCESS; }
int fit_window_sharded(bodyfit_problem* problem, int rank, int size, double* my_frame_params /* [F_local (+1 halo)][76] */,
                       double* beta /* replicated, [10] */, int max_iters) {
  bodyfit_comm comm{rank, size, nullptr, ar, ag};
  bodyfit_fit_options opt{};                               // MultiFrameBA.h:146-152
  opt.max_iters = max_iters; opt.scale_lo = 0.3; opt.scale_hi = 3.0;
  bodyfit_fit_summary sum;
  return bodyfit_solve_sharded(problem, my_frame_params, beta, nullptr, &comm, &opt, &sum);
}                                                          // every rank returns the same summary and the same beta
// [/snippet]

// ---- multi-GPU: the same solve with RCCL as the transport (device buffers, the solve's stream) -------------------------------
// [snippet:sharded_rccl]
int fit_window_sharded_rccl(bodyfit_problem* problem, int rank, int size, int device, double* my_frame_params, double* beta,
                            const bodyfit_fit_options* opt, bodyfit_fit_summary* sum) {
  unsigned char id[128];                                   // ncclUniqueId: made on rank 0, carried by any host channel
  if (rank == 0 && bodyfit_rccl_unique_id(id)) return 1;
  MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
  bodyfit_rccl* rc = nullptr;                              // (an application that already has an ncclComm_t: bodyfit_rccl_wrap)
  if (bodyfit_rccl_create(id, rank, size, device, &rc)) return 1;   // ncclCommInitRank: collective
  const int status = bodyfit_solve_sharded_rccl(problem, my_frame_params, beta, nullptr, rc, opt, sum);
  bodyfit_rccl_destroy(rc);                                // (keep it for the next window instead)
  return status;
}
// [/snippet]

// ---- D. overlay ---------------------------------------------------------------------------------------------------------
struct MatLike { unsigned char* data; int rows, cols; size_t step; };   // cv::Mat's members the snippet touches
void snippet_overlay(MatLike vis, const std::vector<double>& cloud_in, const std::vector<std::array<int, 3>>& faces, double fx, double fy, double cx,
                     double cy, bodyfit_problem* problem, const double* frame_params, const double* beta, double* R0_out,
                     double* joints, float* host_or_null, double* mean_px, int device, int n_vertices, int n_faces,
                     const int32_t* faces_int32, int width, int height, int F, uint8_t* d_images_bgr, size_t row_stride,
                     size_t frame_stride, void* stream) {
// [snippet:overlay]
// #include "bodyfit.hpp"
// (1) drop-in, one frame at a time: same name and argument order; cv::Mat -> ImageView, Eigen 3xN -> its data()
//     (cv::Mat vis = img_all[i].clone();  src/main_multi_frame.cpp:209)
std::vector<double> cloud(cloud_in.data(), cloud_in.data() + cloud_in.size());   // avatars[i]->cloud.data()
smpl::render::renderSMPLMesh(cloud, faces, bodyfit::ImageView{vis.data, vis.rows, vis.cols, vis.step},
                             fx, fy, cx, cy, /*fill=*/true, /*cull=*/true, /*wire=*/false);

// (2) batched, nothing leaves the device: write-back of a whole solve, then all overlays in one call
bodyfit_writeback_batch(problem, frame_params, beta, R0_out, joints, /*cloud=*/host_or_null, mean_px);
bodyfit_device_views v; bodyfit_problem_views(problem, &v);             // v.cloud: float [F][cloud_frame_stride]
bodyfit_overlay_desc od{device, n_vertices, n_faces, faces_int32, width, height, /*max_frames=*/F};
bodyfit_overlay* ov; bodyfit_overlay_create(&od, &ov);                   // once per model / image size
bodyfit_overlay_render_device(ov, v.cloud, /*f64=*/0, v.cloud_frame_stride, F, d_images_bgr, row_stride, frame_stride,
                              fx, fy, cx, cy, 1, 1, 0, stream);
// [/snippet]
}
