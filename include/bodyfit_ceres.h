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
 * repository's image: tests/test_ceres_adapter.py compiles this header against an interface double that declares only the
 * members used here and drives the blocks the way ceres::Problem::Evaluate does.
 *
 * Parameter memory is the caller's, laid out as the reference lays it out: FramePoseParams of frame f = 76 contiguous doubles
 * [scale, rootAA(3), rootT(3), jointAA[1](3) ... jointAA[23](3)] (include/Sim3BA.h:36-40), beta = 10 doubles.             */
#ifndef BODYFIT_CERES_H_
#define BODYFIT_CERES_H_

#include <ceres/ceres.h>

#include <vector>

#include "bodyfit.h"

namespace bodyfit_ceres {

/* One device sweep per point: Ceres calls this before it evaluates the residual blocks of a new point
 * (Solver::Options::evaluation_callback, or Problem::Options::evaluation_callback from Ceres 2.0 on).  Without it the blocks
 * still work: bodyfit_evaluate_block re-sweeps when its parameters differ from the cached point.                         */
class SweepCallback : public ceres::EvaluationCallback {
 public:
  SweepCallback(bodyfit_problem* p, const double* frame_params, const double* beta) : p_(p), x_(frame_params), beta_(beta) {}
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    if (new_evaluation_point || (evaluate_jacobians && !have_jacobian_)) {
      ok_ = bodyfit_evaluate_batch(p_, x_, beta_, nullptr, nullptr, nullptr, evaluate_jacobians ? 1 : 0) == BODYFIT_OK;
      have_jacobian_ = evaluate_jacobians;
    }
  }
  bool ok() const { return ok_; }

 private:
  bodyfit_problem* p_;
  const double* x_;
  const double* beta_;
  bool have_jacobian_ = false, ok_ = true;
};

/* ceres::CostFunction::Evaluate(parameters, residuals, jacobians) of one block = bodyfit_evaluate_block (both NULL levels
 * of `jacobians` honoured; a false return tells Ceres the step is infeasible).                                            */
class Block : public ceres::CostFunction {
 public:
  Block(bodyfit_problem* p, int kind, int index, int num_residuals, const std::vector<int>& block_sizes)
      : p_(p), kind_(kind), index_(index) {
    set_num_residuals(num_residuals);
    for (int s : block_sizes) mutable_parameter_block_sizes()->push_back(s);
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    return bodyfit_evaluate_block(p_, kind_, index_, parameters, residuals, jacobians) == BODYFIT_OK;
  }

 private:
  bodyfit_problem* p_;
  int kind_, index_;
};

/* the parameter blocks of frame f as Ceres sees them (include/Sim3BA.h:421-430): scale, rootAA, rootT, 23 joints */
inline std::vector<double*> FrameBlocks(double* frame_params, int f) {
  double* x = frame_params + (size_t)f * BODYFIT_FRAME_PARAMS;
  std::vector<double*> b = {x, x + 1, x + 4};
  for (int j = 0; j < 23; ++j) b.push_back(x + 7 + 3 * j);
  return b;
}

struct AddOptions {
  double huber_delta = 3.0;    /* HuberLoss on the reprojection blocks (include/MultiFrameBA.h:102); <= 0: none        */
  bool beta_per_frame = false; /* 3dba_single --opt-shape: every frame its own beta[10] (beta = [F][10])              */
};

/* Add every residual block of `p` to `problem`, in the reference's order.  kp_offset [F + 1] is the CSR the problem was
 * created with; n_cols 76 (no shape block) or 86; prior / temporal blocks are added when the problem has them
 * (bodyfit_problem_layout).  Returns the number of residual blocks added.                                                */
inline int AddResidualBlocks(ceres::Problem* problem, bodyfit_problem* p, int n_frames, const int* kp_offset, double* frame_params,
                             double* beta, const AddOptions& opt = AddOptions()) {
  bodyfit_layout L;
  if (bodyfit_problem_layout(p, &L) != BODYFIT_OK) return -1;
  const bool with_beta = L.n_cols > BODYFIT_FRAME_PARAMS;
  const int nS = L.n_cols - BODYFIT_FRAME_PARAMS;
  int added = 0;
  std::vector<int> reproj_sizes = {1, 3, 3};
  for (int j = 0; j < 23; ++j) reproj_sizes.push_back(3);
  if (with_beta) reproj_sizes.push_back(nS);
  for (int f = 0; f < n_frames; ++f) {
    std::vector<double*> blocks = FrameBlocks(frame_params, f);
    if (with_beta) blocks.push_back(beta + (opt.beta_per_frame ? (size_t)f * nS : 0));
    for (int k = kp_offset[f]; k < kp_offset[f + 1]; ++k) {
      ceres::LossFunction* loss = opt.huber_delta > 0.0 ? new ceres::HuberLoss(opt.huber_delta) : nullptr;
      problem->AddResidualBlock(new Block(p, 0, k, 2, reproj_sizes), loss, blocks);
      ++added;
    }
  }
  if (L.prior_rows_per_frame > 0) {
    const std::vector<int> sizes(23, 3);
    for (int f = 0; f < n_frames; ++f) {
      std::vector<double*> fb = FrameBlocks(frame_params, f);
      problem->AddResidualBlock(new Block(p, 1, f, L.prior_rows_per_frame, sizes), nullptr,
                                std::vector<double*>(fb.begin() + 3, fb.end()));
      ++added;
    }
  }
  if (L.shape_rows > 0) {
    const int n_shape_blocks = L.shape_rows / nS;
    for (int i = 0; i < n_shape_blocks; ++i) {
      problem->AddResidualBlock(new Block(p, 2, i, nS, {nS}), nullptr, std::vector<double*>{beta + (size_t)i * nS});
      ++added;
    }
  }
  if (L.temporal_rows > 0) {
    const int n_pairs = L.temporal_rows / 75;
    for (int pr = 0; pr < n_pairs; ++pr) {
      std::vector<double*> a = FrameBlocks(frame_params, pr), b = FrameBlocks(frame_params, pr + 1);
      for (int slot = 0; slot < 25; ++slot) {   // rootT, rootAA, joints 1..23 (include/MultiFrameBA.h:121-142)
        const int bi = slot == 0 ? 2 : (slot == 1 ? 1 : slot + 1);
        problem->AddResidualBlock(new Block(p, 3, 25 * pr + slot, 3, {3, 3}), nullptr, std::vector<double*>{a[bi], b[bi]});
        ++added;
      }
    }
  }
  return added;
}

}  // namespace bodyfit_ceres

#endif  /* BODYFIT_CERES_H_ */
