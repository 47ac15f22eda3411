// INTERFACE DOUBLE, test infrastructure — NOT Ceres.  Declares, with the signatures of Ceres 1.14's public headers
// (include/ceres/cost_function.h, evaluation_callback.h, loss_function.h, problem.h), exactly the members that
// include/bodyfit_ceres.h uses, so that the adapter can be compiled and its blocks driven the way
// ceres::Problem::Evaluate drives them, plus the three names INTEGRATION.md's snippet mentions (Solver::Options with its
// evaluation_callback member, Solver::Summary, Solve: DECLARED only, so the snippet can be syntax-checked).  Ceres itself is
// not in this repository's image.  No solver here.
#pragma once
#include <memory>
#include <vector>

namespace ceres {

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int> parameter_block_sizes_;
  int num_residuals_;
};

class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override;   // (defined by the test driver; the adapter only constructs it)
  double a_, b_;
};

typedef struct ResidualBlockRecord* ResidualBlockId;
struct ResidualBlockRecord {
  std::unique_ptr<CostFunction> cost;
  std::unique_ptr<LossFunction> loss;
  std::vector<double*> blocks;
};

class Problem {
 public:
  ResidualBlockId AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& parameter_blocks) {
    records_.emplace_back(new ResidualBlockRecord{std::unique_ptr<CostFunction>(cost), std::unique_ptr<LossFunction>(loss),
                                                 parameter_blocks});
    return records_.back().get();
  }
  const std::vector<std::unique_ptr<ResidualBlockRecord>>& records() const { return records_; }

 private:
  std::vector<std::unique_ptr<ResidualBlockRecord>> records_;
};

class Solver {
 public:
  struct Options { EvaluationCallback* evaluation_callback = nullptr; int max_num_iterations = 50; };
  struct Summary {};
};
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);   // declared, never defined

}  // namespace ceres
