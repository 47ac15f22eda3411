/* bodyfit_ceres.h — the Ceres side of the drop-in: the residual blocks of one bodyfit_problem as ceres::CostFunction objects
 * over the reference's own parameter blocks, so that "the Ceres outer loop is kept" (north_star) while every Evaluate is a
 * slice of ONE batched device sweep.
 *
 * Replaces, block for block, what the reference adds to its ceres::Problem:
 *   include/MultiFrameBA.h:85-102   DynamicAutoDiffCostFunction<ReprojCostShape> per keypoint, HuberLoss(3), blocks
 *                                   {scale, rootAA, rootT, jointAA[1..23], beta}           -> kind 0
 *   include/MultiFrameBA.h:106-111  PosePriorAAAnalytic per frame over the 23 joint blocks -> kind 1
 *   include/MultiFrameBA.h:115-118  ShapePriorL2Analytic on beta                           -> kind 2
 *   include/MultiFrameBA.h:121-142  Vec3DiffCost on rootT, rootAA and the 23 joints of consecutive frames -> kind 3
 *   include/Sim3BA.h:421-479,556-647 the single-frame variants of the same blocks
 * Header-only; needs <ceres/ceres.h> (>= 1.14 for EvaluationCallback) from the application.  Ceres is not part of this
 * repository's image: tests/test_gpu_cpp_api.py::test_ceres_adapter_blocks compiles this header against an interface double
 * (tests/cpp/ceres_double) that declares only the members used here, and tests/cpp/ceres_adapter_demo.cpp drives the blocks
 * the way ceres::Problem::Evaluate does.
 *
 * Parameter memory is the caller's and stays where the reference keeps it.  The reference's FramePoseParams
 * (include/MultiFrameBA.h:9-14) is NOT 76 contiguous doubles: scale, rootAA[3], rootT[3] are members and the joints live in a
 * std::vector<std::array<double,3>> of 24 entries whose slot 0 is unused.  A BlockTable therefore holds, per frame, the 26
 * block pointers Ceres sees (&P.scale, P.rootAA, P.rootT, P.jointAA[j].data(): include/MultiFrameBA.h:74-78,98-100);
 * BlocksOf(poses) builds it from such structs, BlocksOfPacked(x, F) from a packed [F][76] array
 * ([scale, rootAA(3), rootT(3), jointAA[1](3) ... jointAA[23](3)], the block ORDER of include/Sim3BA.h:36-40).  The sweep
 * callback gathers the blocks into a packed staging buffer of its own before each device sweep.                          */
#ifndef BODYFIT_CERES_H_
#define BODYFIT_CERES_H_

#include <ceres/ceres.h>

#include <array>
#include <utility>
#include <vector>

#include "bodyfit.h"

namespace bodyfit_ceres {

constexpr int kFrameBlocks = 26;   /* scale, rootAA, rootT, jointAA[1..23] */

/* the parameter blocks of every frame as Ceres sees them (include/Sim3BA.h:421-430) */
struct BlockTable {
  std::vector<std::array<double*, kFrameBlocks>> frame;
  int n_frames() const { return (int)frame.size(); }
  std::vector<double*> blocks(int f) const { return std::vector<double*>(frame[f].begin(), frame[f].end()); }
};
/* from the reference's per-frame structs: anything with .scale, .rootAA, .rootT and .jointAA[j].data(), j = 1..23
 * (FramePoseParams of include/MultiFrameBA.h:9-14; bodyfit::FramePoseParams of bodyfit.hpp) */
template <class FramePose>
inline BlockTable BlocksOf(std::vector<FramePose>& poses) {
  BlockTable t;
  t.frame.resize(poses.size());
  for (size_t f = 0; f < poses.size(); ++f) {
    FramePose& P = poses[f];
    t.frame[f][0] = &P.scale; t.frame[f][1] = P.rootAA; t.frame[f][2] = P.rootT;
    for (int j = 1; j < 24; ++j) t.frame[f][2 + j] = P.jointAA[j].data();
  }
  return t;
}
/* from a packed [F][76] array */
inline BlockTable BlocksOfPacked(double* frame_params, int n_frames) {
  BlockTable t;
  t.frame.resize(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    double* x = frame_params + (size_t)f * BODYFIT_FRAME_PARAMS;
    t.frame[f][0] = x; t.frame[f][1] = x + 1; t.frame[f][2] = x + 4;
    for (int j = 0; j < 23; ++j) t.frame[f][3 + j] = x + 7 + 3 * j;
  }
  return t;
}
/* kept for callers of the first version of this header */
inline std::vector<double*> FrameBlocks(double* frame_params, int f) { return BlocksOfPacked(frame_params + (size_t)f * BODYFIT_FRAME_PARAMS, 1).blocks(0); }

/* One device sweep per point: Ceres calls this before it evaluates the residual blocks of a new point
 * (Solver::Options::evaluation_callback, or Problem::Options::evaluation_callback from Ceres 2.0 on).  Without it the blocks
 * still work: bodyfit_evaluate_block re-sweeps when its parameters differ from the cached point.  The blocks are gathered
 * into a packed [F][76] staging buffer (they need not be contiguous, see above).                                        */
class SweepCallback : public ceres::EvaluationCallback {
 public:
  SweepCallback(bodyfit_problem* p, BlockTable blocks, const double* beta)
      : p_(p), t_(std::move(blocks)), beta_(beta), x_((size_t)t_.n_frames() * BODYFIT_FRAME_PARAMS) {}
  SweepCallback(bodyfit_problem* p, double* frame_params_packed, int n_frames, const double* beta)
      : SweepCallback(p, BlocksOfPacked(frame_params_packed, n_frames), beta) {}
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    if (new_evaluation_point || (evaluate_jacobians && !have_jacobian_)) {
      for (int f = 0; f < t_.n_frames(); ++f) {
        double* x = x_.data() + (size_t)f * BODYFIT_FRAME_PARAMS;
        const auto& b = t_.frame[f];
        x[0] = b[0][0];
        for (int c = 0; c < 3; ++c) { x[1 + c] = b[1][c]; x[4 + c] = b[2][c]; }
        for (int j = 0; j < 23; ++j)
          for (int c = 0; c < 3; ++c) x[7 + 3 * j + c] = b[3 + j][c];
      }
      ok_ = bodyfit_evaluate_batch(p_, x_.data(), beta_, nullptr, nullptr, nullptr, evaluate_jacobians ? 1 : 0) == BODYFIT_OK;
      have_jacobian_ = evaluate_jacobians;
    }
  }
  bool ok() const { return ok_; }

 private:
  bodyfit_problem* p_;
  BlockTable t_;
  const double* beta_;
  std::vector<double> x_;
  bool have_jacobian_ = false, ok_ = true;
};

/* ceres::CostFunction::Evaluate(parameters, residuals, jacobians) of one block = bodyfit_evaluate_block (both NULL levels
 * of `jacobians` honoured; a false return tells Ceres the step is infeasible).                                            */
class Block : public ceres::CostFunction {
 public:
  /* with_callback: a SweepCallback is registered as the solver's evaluation_callback, so the cached sweep is the point under
   * evaluation and the block is served without re-checking its parameters (bodyfit_evaluate_block_cached)              */
  Block(bodyfit_problem* p, int kind, int index, int num_residuals, const std::vector<int>& block_sizes, bool with_callback = false)
      : p_(p), kind_(kind), index_(index), cached_(with_callback) {
    set_num_residuals(num_residuals);
    for (int s : block_sizes) mutable_parameter_block_sizes()->push_back(s);
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    return (cached_ ? bodyfit_evaluate_block_cached(p_, kind_, index_, parameters, residuals, jacobians)
                    : bodyfit_evaluate_block(p_, kind_, index_, parameters, residuals, jacobians)) == BODYFIT_OK;
  }
  /* which block of the bodyfit_problem this is (bodyfit_evaluate_block's kind / index) */
  int kind() const { return kind_; }
  int index() const { return index_; }

 private:
  bodyfit_problem* p_;
  int kind_, index_;
  bool cached_;
};

struct AddOptions {
  double huber_delta = 3.0;    /* HuberLoss on the reprojection blocks (include/MultiFrameBA.h:102); <= 0: none        */
  bool beta_per_frame = false; /* 3dba_single --opt-shape: every frame its own beta[10] (beta = [F][10])              */
  bool with_callback = false;  /* a SweepCallback will be registered as evaluation_callback: blocks trust the cached sweep */
};

/* Add every residual block of `p` to `problem`, in the reference's order.  kp_offset [F + 1] is the CSR the problem was
 * created with; n_cols 76 (no shape block) or 86; prior / temporal blocks are added when the problem has them
 * (bodyfit_problem_layout).  Returns the number of residual blocks added.                                                */
inline int AddResidualBlocks(ceres::Problem* problem, bodyfit_problem* p, const int* kp_offset, const BlockTable& T,
                             double* beta, const AddOptions& opt = AddOptions()) {
  bodyfit_layout L;
  if (bodyfit_problem_layout(p, &L) != BODYFIT_OK) return -1;
  const int n_frames = T.n_frames();
  const bool with_beta = L.n_cols > BODYFIT_FRAME_PARAMS;
  const int nS = L.n_cols - BODYFIT_FRAME_PARAMS;
  int added = 0;
  std::vector<int> reproj_sizes = {1, 3, 3};
  for (int j = 0; j < 23; ++j) reproj_sizes.push_back(3);
  if (with_beta) reproj_sizes.push_back(nS);
  // Block order = the reference's: per frame its reprojection blocks and then its pose prior (include/MultiFrameBA.h:71-112,
  // include/Sim3BA.h:417-462), [a frame's own shape prior with --opt-shape batches], then the shared shape prior (:115-118),
  // then the temporal links (:121-142).  (Ceres hands contiguous runs of residual blocks to its evaluation threads: with every
  // pose prior at the end of the list — the 70 x 69 GMM Jacobian is 38 KB per block — one thread got all of them.)
  const std::vector<int> prior_sizes(23, 3);
  const int n_shape_blocks = L.shape_rows > 0 ? L.shape_rows / nS : 0;
  for (int f = 0; f < n_frames; ++f) {
    std::vector<double*> blocks = T.blocks(f);
    std::vector<double*> rb = blocks;
    if (with_beta) rb.push_back(beta + (opt.beta_per_frame ? (size_t)f * nS : 0));
    for (int k = kp_offset[f]; k < kp_offset[f + 1]; ++k) {
      ceres::LossFunction* loss = opt.huber_delta > 0.0 ? new ceres::HuberLoss(opt.huber_delta) : nullptr;
      problem->AddResidualBlock(new Block(p, 0, k, 2, reproj_sizes, opt.with_callback), loss, rb);
      ++added;
    }
    if (L.prior_rows_per_frame > 0) {
      problem->AddResidualBlock(new Block(p, 1, f, L.prior_rows_per_frame, prior_sizes, opt.with_callback), nullptr,
                                std::vector<double*>(blocks.begin() + 3, blocks.end()));
      ++added;
    }
    if (n_shape_blocks > 1 && f < n_shape_blocks) {   // beta per frame: the frame's own ShapePriorL2Analytic
      problem->AddResidualBlock(new Block(p, 2, f, nS, {nS}), nullptr, std::vector<double*>{beta + (size_t)f * nS});
      ++added;
    }
  }
  if (n_shape_blocks == 1) {
    problem->AddResidualBlock(new Block(p, 2, 0, nS, {nS}), nullptr, std::vector<double*>{beta});
    ++added;
  }
  if (L.temporal_rows > 0) {
    const int n_pairs = L.temporal_rows / 75;
    for (int pr = 0; pr < n_pairs && pr + 1 < n_frames; ++pr) {
      const auto &a = T.frame[pr], &b = T.frame[pr + 1];
      for (int slot = 0; slot < 25; ++slot) {   // rootT, rootAA, joints 1..23 (include/MultiFrameBA.h:121-142)
        const int bi = slot == 0 ? 2 : (slot == 1 ? 1 : slot + 1);
        problem->AddResidualBlock(new Block(p, 3, 25 * pr + slot, 3, {3, 3}), nullptr, std::vector<double*>{a[bi], b[bi]});
        ++added;
      }
    }
  }
  return added;
}
/* packed [F][76] parameters */
inline int AddResidualBlocks(ceres::Problem* problem, bodyfit_problem* p, int n_frames, const int* kp_offset, double* frame_params,
                             double* beta, const AddOptions& opt = AddOptions()) {
  return AddResidualBlocks(problem, p, kp_offset, BlocksOfPacked(frame_params, n_frames), beta, opt);
}

}  // namespace bodyfit_ceres

#endif  /* BODYFIT_CERES_H_ */
