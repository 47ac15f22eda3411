// Drives include/bodyfit_ceres.h the way ceres::Problem::Evaluate would (tests/cpp/ceres_double is an interface double, not
// Ceres): every residual block of a shared-beta window is added to a Problem, the evaluation callback runs one device sweep,
// every block is evaluated with Ceres' pointer conventions, and the assembled residual vector / Jacobian is compared with
// bodyfit_evaluate_batch's.  The parameters live where the reference keeps them: one FramePoseParams per frame (scale, rootAA,
// rootT as members, the joints in a std::vector<std::array<double,3>> whose slot 0 is unused: include/MultiFrameBA.h:9-14), so
// nothing is contiguous.  Input blob: the format of tests/test_gpu_cpp_api.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bodyfit.hpp"
#include "bodyfit_ceres.h"

void ceres::HuberLoss::Evaluate(double s, double rho[3]) const {
  if (s > b_) { const double r = std::sqrt(s); rho[0] = 2 * a_ * r - b_; rho[1] = a_ / r; rho[2] = -rho[1] / (2 * s); }
  else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
}

template <typename T>
static std::vector<T> rd(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  auto hdr = rd<int>(f, 7);
  const int V = hdr[0], nJ = hdr[1], nS = hdr[2], P = hdr[3], nL = hdr[4], F = hdr[5], K = hdr[6];
  auto vt = rd<double>(f, (size_t)V * 3), sd = rd<double>(f, (size_t)V * 3 * nS), pd = rd<double>(f, (size_t)V * 3 * P),
       jr = rd<double>(f, (size_t)nJ * V), w = rd<double>(f, (size_t)V * nJ);
  auto parent = rd<int>(f, nJ), lvid = rd<int>(f, nL), koff = rd<int>(f, F + 1), kid = rd<int>(f, K);
  auto uv = rd<double>(f, (size_t)2 * K), intr = rd<double>(f, 4);
  std::fclose(f);

  bodyfit_model_desc md{V, nJ, nS, P, vt.data(), sd.data(), pd.data(), jr.data(), w.data(), parent.data(), nL, lvid.data()};
  bodyfit_model* model = nullptr;
  if (bodyfit_model_create(&md, 0, &model) != BODYFIT_OK) { std::fprintf(stderr, "%s\n", bodyfit_last_error()); return 1; }
  std::vector<double> R0((size_t)F * 9, 0.0);
  for (int i = 0; i < F; ++i) R0[i * 9] = R0[i * 9 + 4] = R0[i * 9 + 8] = -1.0;
  bodyfit_problem_desc pdsc{};
  pdsc.n_frames = F; pdsc.kp_offset = koff.data(); pdsc.kp_id = kid.data(); pdsc.kp_uv = uv.data();
  pdsc.fx = intr[0]; pdsc.fy = intr[1]; pdsc.cx = intr[2]; pdsc.cy = intr[3];
  pdsc.R0 = R0.data(); pdsc.n_cols = 86; pdsc.use_shape = 1; pdsc.pose_blend = 1;
  pdsc.beta_pose = 5.0; pdsc.beta_shape = 25.0; pdsc.lambda_temporal = 3.0; pdsc.huber_delta = 3.0;
  bodyfit_problem* bp = nullptr;
  if (bodyfit_problem_create(model, &pdsc, &bp) != BODYFIT_OK) { std::fprintf(stderr, "%s\n", bodyfit_last_error()); return 1; }
  bodyfit_layout L;
  bodyfit_problem_layout(bp, &L);

  // the parameters Ceres would own: the reference's FramePoseParams per frame (NOT contiguous), one shared beta
  std::vector<bodyfit::FramePoseParams> poses(F);
  std::vector<double> beta(10, 0.0);
  auto packed = [&](std::vector<double>& x) {     // the same point as a packed [F][76] array, for the reference evaluation
    x.assign((size_t)F * 76, 0.0);
    for (int i = 0; i < F; ++i) {
      double* xi = x.data() + (size_t)i * 76;
      xi[0] = poses[i].scale;
      for (int c = 0; c < 3; ++c) { xi[1 + c] = poses[i].rootAA[c]; xi[4 + c] = poses[i].rootT[c]; }
      for (int j = 1; j < 24; ++j)
        for (int c = 0; c < 3; ++c) xi[7 + 3 * (j - 1) + c] = poses[i].jointAA[j][c];
    }
  };
  for (int i = 0; i < F; ++i) {
    bodyfit::FramePoseParams& P = poses[i];
    P.jointAA.assign(24, {0.0, 0.0, 0.0});
    std::vector<double> xi(76, 0.0);
    xi[0] = 1.0 + 0.01 * i; xi[6] = 3.0;
    for (int c = 1; c < 76; ++c)
      if (c != 6) xi[c] += 0.05 * std::sin(0.37 * c + 1.3 * i);
    P.scale = xi[0];
    for (int c = 0; c < 3; ++c) { P.rootAA[c] = xi[1 + c]; P.rootT[c] = xi[4 + c]; }
    for (int j = 1; j < 24; ++j)
      for (int c = 0; c < 3; ++c) P.jointAA[j][c] = xi[7 + 3 * (j - 1) + c];
    P.jointAA[0] = {123.0, 456.0, 789.0};   // the unused slot must not matter
  }
  for (int k = 0; k < 10; ++k) beta[k] = 0.3 * std::cos(1.1 * k);

  ceres::Problem problem;
  const bodyfit_ceres::BlockTable table = bodyfit_ceres::BlocksOf(poses);
  const int n_blocks = bodyfit_ceres::AddResidualBlocks(&problem, bp, koff.data(), table, beta.data());
  const int expect_blocks = K + F + 1 + 25 * (F - 1);
  bodyfit_ceres::SweepCallback cb(bp, table, beta.data());
  // column of a parameter block in the dense [F x 76 | beta] layout used for the comparison
  auto column_of = [&](const double* base) -> size_t {
    if (base >= beta.data() && base < beta.data() + 10) return (size_t)F * 76 + (base - beta.data());
    for (int f2 = 0; f2 < F; ++f2)
      for (int b2 = 0; b2 < bodyfit_ceres::kFrameBlocks; ++b2)
        if (table.frame[f2][b2] == base) return (size_t)f2 * 76 + (b2 == 0 ? 0 : b2 == 1 ? 1 : b2 == 2 ? 4 : 7 + 3 * (b2 - 3));
    return (size_t)-1;
  };

  // what ceres::Problem::Evaluate does: callback, then every block with its parameter pointers and Jacobian buffers
  cb.PrepareForEvaluation(true, true);
  const int ncol = F * 76 + 10;
  // rows in the batched evaluation's layout [reprojection | pose prior | shape prior | temporal] (the blocks come in the
  // reference's order, frame by frame: each block's rows go where bodyfit_evaluate_batch has them), dense columns [F x 76 | beta]
  std::vector<double> r_all((size_t)L.total_rows, 0.0), J_all((size_t)L.total_rows * (F * 76 + 10), 0.0);
  size_t rows_seen = 0;
  int bad = 0;
  for (const auto& rec : problem.records()) {
    const ceres::CostFunction& cf = *rec->cost;
    const int nr = cf.num_residuals();
    const auto& sizes = cf.parameter_block_sizes();
    if (sizes.size() != rec->blocks.size()) ++bad;
    std::vector<double> r(nr);
    std::vector<std::vector<double>> jb(sizes.size());
    std::vector<double*> jp(sizes.size());
    for (size_t b = 0; b < sizes.size(); ++b) { jb[b].assign((size_t)nr * sizes[b], 0.0); jp[b] = jb[b].data(); }
    if (sizes.size() > 5) jp[5] = nullptr;          // a constant parameter block: jacobians[b] == NULL
    if (!cf.Evaluate(rec->blocks.data(), r.data(), jp.data())) ++bad;
    const auto* blk = dynamic_cast<const bodyfit_ceres::Block*>(rec->cost.get());
    if (!blk) { ++bad; continue; }
    const int nS_ = 10;
    const size_t row0 = blk->kind() == 0 ? (size_t)2 * blk->index()
                      : blk->kind() == 1 ? (size_t)L.reproj_rows + (size_t)blk->index() * L.prior_rows_per_frame
                      : blk->kind() == 2 ? (size_t)L.reproj_rows + (size_t)F * L.prior_rows_per_frame + (size_t)blk->index() * nS_
                                         : (size_t)L.reproj_rows + (size_t)F * L.prior_rows_per_frame + L.shape_rows + (size_t)3 * blk->index();
    for (int i = 0; i < nr; ++i) r_all[row0 + i] = r[i];
    rows_seen += nr;
    for (size_t b = 0; b < sizes.size(); ++b) {
      if (!jp[b]) continue;
      const size_t col = column_of(rec->blocks[b]);
      if (col == (size_t)-1) { ++bad; continue; }
      for (int i = 0; i < nr; ++i)
        for (int c = 0; c < sizes[b]; ++c) J_all[(row0 + i) * ncol + col + c] = jb[b][(size_t)i * sizes[b] + c];
    }
  }
  // reference: the batched evaluation of the same point
  std::vector<double> r_ref(L.total_rows), J_ref((size_t)L.reproj_rows * 86), x;
  packed(x);
  if (bodyfit_evaluate_batch(bp, x.data(), beta.data(), r_ref.data(), J_ref.data(), nullptr, 1) != BODYFIT_OK) return 1;
  double dr = 0.0, dj = 0.0;
  if ((int)rows_seen != L.total_rows) ++bad;
  for (size_t i = 0; i < r_all.size() && i < r_ref.size(); ++i) dr = std::fmax(dr, std::fabs(r_all[i] - r_ref[i]));
  for (int fr = 0; fr < F; ++fr)
    for (int k = koff[fr]; k < koff[fr + 1]; ++k)
      for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 86; ++c) {
          if (c >= 7 + 3 * 2 && c < 10 + 3 * 2) continue;   // block 5 (joint 3) was held constant above
          const size_t col = c < 76 ? (size_t)fr * 76 + c : (size_t)F * 76 + (c - 76);
          dj = std::fmax(dj, std::fabs(J_all[((size_t)2 * k + i) * ncol + col] - J_ref[((size_t)2 * k + i) * 86 + c]));
        }
  // a second call at the same point costs no sweep and gives the same numbers; a moved point is re-evaluated
  cb.PrepareForEvaluation(false, false);
  poses[0].rootT[0] += 0.01;
  packed(x);
  cb.PrepareForEvaluation(false, true);
  std::vector<double> r2(2);
  const auto& rec0 = *problem.records()[0];
  rec0.cost->Evaluate(rec0.blocks.data(), r2.data(), nullptr);
  std::vector<double> r_ref2(L.total_rows);
  bodyfit_evaluate_batch(bp, x.data(), beta.data(), r_ref2.data(), nullptr, nullptr, 0);
  const double dr2 = std::fmax(std::fabs(r2[0] - r_ref2[0]), std::fabs(r2[1] - r_ref2[1]));
  // the EvaluationCallback form of the blocks (AddOptions::with_callback: the cached sweep is trusted, no parameter check, no
  // lock): the same numbers, block for block, at the point the callback has just swept
  ceres::Problem problem_cb;
  bodyfit_ceres::AddOptions with_cb;
  with_cb.with_callback = true;
  const int n_blocks_cb = bodyfit_ceres::AddResidualBlocks(&problem_cb, bp, koff.data(), table, beta.data(), with_cb);
  cb.PrepareForEvaluation(true, true);
  double d_cb = 0.0;
  for (size_t bi = 0; bi < problem.records().size(); ++bi) {
    const auto& ra = *problem.records()[bi];
    const auto& rb = *problem_cb.records()[bi];
    const int nr = ra.cost->num_residuals();
    const auto& sizes = ra.cost->parameter_block_sizes();
    std::vector<double> r_a(nr), r_b(nr);
    std::vector<std::vector<double>> ja(sizes.size()), jbk(sizes.size());
    std::vector<double*> pa(sizes.size()), pb(sizes.size());
    for (size_t b = 0; b < sizes.size(); ++b) { ja[b].assign((size_t)nr * sizes[b], 0.0); jbk[b].assign((size_t)nr * sizes[b], 1.0); pa[b] = ja[b].data(); pb[b] = jbk[b].data(); }
    if (!ra.cost->Evaluate(ra.blocks.data(), r_a.data(), pa.data()) || !rb.cost->Evaluate(rb.blocks.data(), r_b.data(), pb.data())) ++bad;
    for (int i = 0; i < nr; ++i) d_cb = std::fmax(d_cb, std::fabs(r_a[i] - r_b[i]));
    for (size_t b = 0; b < sizes.size(); ++b)
      for (size_t i = 0; i < ja[b].size(); ++i) d_cb = std::fmax(d_cb, std::fabs(ja[b][i] - jbk[b][i]));
  }
  if (n_blocks_cb != n_blocks || d_cb != 0.0) ++bad;
  std::printf("callback-form blocks %d, max difference to the checked form %.3e\n", n_blocks_cb, d_cb);
  std::printf("blocks %d (expected %d) rows %zu (layout %d) max|dr| %.3e max|dJ| %.3e moved %.3e bad %d\n", n_blocks,
              expect_blocks, r_all.size(), L.total_rows, dr, dj, dr2, bad);
  const bool ok = n_blocks == expect_blocks && bad == 0 && dr == 0.0 && dj == 0.0 && dr2 == 0.0 && cb.ok();
  std::printf(ok ? "ceres adapter: OK\n" : "ceres adapter: MISMATCH\n");
  bodyfit_problem_destroy(bp);
  bodyfit_model_destroy(model);
  return ok ? 0 : 1;
}
