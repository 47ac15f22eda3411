// Throughput of the Ceres-kept path (north_star: "the Ceres outer loop is kept"): what a ceres::Solve over
// include/bodyfit_ceres.h pays per evaluation point — the EvaluationCallback's device sweep with its PCIe copies
// (parameters up, residuals and Jacobian down, page-locked mirrors) and every residual block's Evaluate slicing that sweep
// with Ceres' pointer conventions.  Ceres itself is not in this image: the blocks are driven through the interface double
// tests/cpp/ceres_double exactly the way ceres::Problem::Evaluate drives them (include/Sim3BA.h:263-264,420,476-479).
// usage: ceres_path_bench <blob> <mode: c3 | c4> <seconds>      (blob: the format of tests/test_gpu_cpp_api.py)
// prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
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
  if (argc < 4) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  const std::string mode = argv[2];
  const double seconds = std::atof(argv[3]);
  auto hdr = rd<int>(f, 7);
  const int V = hdr[0], nJ = hdr[1], nS = hdr[2], P = hdr[3], nL = hdr[4], F = hdr[5], K = hdr[6];
  auto vt = rd<double>(f, (size_t)V * 3), sd = rd<double>(f, (size_t)V * 3 * nS), pd = rd<double>(f, (size_t)V * 3 * P),
       jr = rd<double>(f, (size_t)nJ * V), w = rd<double>(f, (size_t)V * nJ);
  auto parent = rd<int>(f, nJ), lvid = rd<int>(f, nL), koff = rd<int>(f, F + 1), kid = rd<int>(f, K);
  auto uv = rd<double>(f, (size_t)2 * K), intr = rd<double>(f, 4);
  std::fclose(f);
  const bool c3 = mode == "c3";   // c3: independent frames, own beta each, mesh on; c4: one shared-beta window

  bodyfit_model_desc md{V, nJ, nS, P, vt.data(), sd.data(), pd.data(), jr.data(), w.data(), parent.data(), nL, lvid.data()};
  bodyfit_model* model = nullptr;
  if (bodyfit_model_create(&md, 0, &model) != BODYFIT_OK) { std::fprintf(stderr, "%s\n", bodyfit_last_error()); return 1; }
  std::vector<double> R0((size_t)F * 9, 0.0);
  for (int i = 0; i < F; ++i) R0[i * 9] = R0[i * 9 + 4] = R0[i * 9 + 8] = -1.0;
  bodyfit_problem_desc pdsc{};
  pdsc.n_frames = F; pdsc.kp_offset = koff.data(); pdsc.kp_id = kid.data(); pdsc.kp_uv = uv.data();
  pdsc.fx = intr[0]; pdsc.fy = intr[1]; pdsc.cx = intr[2]; pdsc.cy = intr[3];
  pdsc.R0 = R0.data(); pdsc.n_cols = 86; pdsc.use_shape = 1; pdsc.pose_blend = 1; pdsc.huber_delta = 3.0;
  if (c3) { pdsc.beta_per_frame = 1; pdsc.beta_pose = 20.0; pdsc.beta_shape = 30.0; pdsc.want_mesh = 1; }
  else { pdsc.beta_pose = 5.0; pdsc.beta_shape = 25.0; pdsc.lambda_temporal = 3.0; }
  bodyfit_problem* bp = nullptr;
  if (bodyfit_problem_create(model, &pdsc, &bp) != BODYFIT_OK) { std::fprintf(stderr, "%s\n", bodyfit_last_error()); return 1; }

  // the reference's own parameter memory: FramePoseParams per frame (not contiguous), beta
  std::vector<bodyfit::FramePoseParams> poses(F);
  for (int i = 0; i < F; ++i) {
    poses[i].scale = 1.0; poses[i].jointAA.assign(24, {0.0, 0.0, 0.0});
    for (int c = 0; c < 3; ++c) { poses[i].rootAA[c] = 0.01 * (c + 1); poses[i].rootT[c] = c == 2 ? 3.0 : 0.0; }
    for (int j = 1; j < 24; ++j)
      for (int c = 0; c < 3; ++c) poses[i].jointAA[j][c] = 0.05 * std::sin(0.37 * (3 * j + c) + 1.3 * i);
  }
  std::vector<double> beta((size_t)(c3 ? F : 1) * 10, 0.1);
  ceres::Problem problem;
  const bodyfit_ceres::BlockTable table = bodyfit_ceres::BlocksOf(poses);
  bodyfit_ceres::AddOptions ao;
  ao.beta_per_frame = c3;
  ao.with_callback = true;      // the sweep callback below is the solver's evaluation_callback
  const int n_blocks = bodyfit_ceres::AddResidualBlocks(&problem, bp, koff.data(), table, beta.data(), ao);
  bodyfit_ceres::SweepCallback cb(bp, table, beta.data());

  // Jacobian buffers of the largest block, one set per evaluation thread (Ceres owns such scratch per thread)
  size_t max_res = 0, max_blocks = 0;
  for (const auto& rec : problem.records()) {
    max_res = std::max<size_t>(max_res, rec->cost->num_residuals());
    max_blocks = std::max(max_blocks, rec->cost->parameter_block_sizes().size());
  }
  // ceres::Problem::Evaluate walks the residual blocks with options.num_threads threads; the reference sets 8
  // (include/Sim3BA.h:476-479, include/MultiFrameBA.h:148).  The same here: a pool of `threads` workers, each with its own
  // scratch, over contiguous ranges of the blocks (argv[4], default 8; 1 = the calling thread alone).
  const int threads = argc > 4 ? std::max(1, std::atoi(argv[4])) : 8;
  // Per-thread scratch laid out the way Ceres lays it out [recalled: scratch_evaluate_preparer.cc, ScratchEvaluatePreparer::Prepare]:
  // ONE buffer per evaluation thread, sized for the largest residual block, and for each residual block the Jacobian pointers
  // handed to Evaluate are consecutive in it — jacobians[j] = cursor; cursor += num_residuals * block_size.  (Round 4's driver
  // gave every parameter block a buffer of its own, 5.6 KB apart: the 27 blocks of a reprojection residual then touched 27
  // scattered lines and each 38 KB pose-prior block pushed them out of the L1.)
  struct Scratch { std::vector<double> r; std::vector<double> j; std::vector<double*> jp; };
  size_t max_jac = 0;
  for (const auto& rec : problem.records()) {
    size_t n = 0;
    for (int sz : rec->cost->parameter_block_sizes()) n += (size_t)rec->cost->num_residuals() * sz;
    max_jac = std::max(max_jac, n);
  }
  std::vector<Scratch> scratch(threads);
  for (auto& sc : scratch) {
    sc.r.resize(max_res);
    sc.j.resize(max_jac);
    sc.jp.resize(max_blocks);
  }
  auto prepare = [](Scratch& sc, const ceres::CostFunction& cost) {
    double* cursor = sc.j.data();
    const auto& sizes = cost.parameter_block_sizes();
    for (size_t b = 0; b < sizes.size(); ++b) { sc.jp[b] = cursor; cursor += (size_t)cost.num_residuals() * sizes[b]; }
  };
  const auto& recs = problem.records();
  const size_t nrec = recs.size();
  std::atomic<long> generation{0}, done{0};
  std::atomic<bool> failed{false}, quit{false};
  // the order the blocks are walked in: the problem's own (= the reference's: per frame its reprojection blocks, then its pose
  // prior), or — second measurement — sorted by kind (all reprojection blocks, then all priors: what bodyfit_ceres.h did before
  // round 4; contiguous thread ranges then hand every 38 KB GMM prior Jacobian to the last thread)
  std::vector<size_t> order(nrec);
  for (size_t i = 0; i < nrec; ++i) order[i] = i;
  size_t n_act = nrec;      // how many entries of `order` a pass walks (the per-kind passes walk a subset)
  auto run_range = [&](int t) {
    const size_t b0 = n_act * t / threads, b1 = n_act * (t + 1) / threads;
    Scratch& sc = scratch[t];
    for (size_t i = b0; i < b1; ++i) {
      const auto& rec = recs[order[i]];
      prepare(sc, *rec->cost);
      if (!rec->cost->Evaluate(rec->blocks.data(), sc.r.data(), sc.jp.data())) failed.store(true);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t)
    pool.emplace_back([&, t] {
      long seen = 0;
      for (;;) {
        while (generation.load(std::memory_order_acquire) == seen && !quit.load()) std::this_thread::yield();
        if (quit.load()) return;
        ++seen;
        run_range(t);
        done.fetch_add(1, std::memory_order_release);
      }
    });
  auto evaluate_blocks = [&]() -> bool {                   // what ceres::Problem::Evaluate does with every block
    done.store(0);
    generation.fetch_add(1, std::memory_order_release);
    run_range(0);
    while (done.load(std::memory_order_acquire) < threads - 1) std::this_thread::yield();
    return !failed.load();
  };

  auto one_point = [&](int it) -> bool {
    poses[it % F].rootT[0] += 1e-6;                       // a new evaluation point
    cb.PrepareForEvaluation(true, true);                  // ONE device sweep, copies included
    if (!cb.ok()) return false;
    return evaluate_blocks();
  };
  for (int it = 0; it < 3; ++it)
    if (!one_point(it)) { std::fprintf(stderr, "evaluation failed: %s\n", bodyfit_last_error()); return 1; }
  int n = 0;
  double t_sweep = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  double el = 0.0;
  while (el < seconds) {
    const auto a = std::chrono::steady_clock::now();
    poses[n % F].rootT[0] += 1e-6;
    cb.PrepareForEvaluation(true, true);
    const auto b = std::chrono::steady_clock::now();
    t_sweep += std::chrono::duration<double>(b - a).count();
    if (!evaluate_blocks()) return 1;
    ++n;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // second measurement: the same points with the blocks walked kind by kind
  auto kind_rank = [&](size_t i) {
    const int nr = recs[i]->cost->num_residuals();
    return nr == 2 ? 0 : (nr >= 69 ? 1 : (nr == 10 ? 2 : 3));
  };
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return kind_rank(a) < kind_rank(b); });
  int n2 = 0;
  double t_sweep2 = 0.0, el2 = 0.0;
  const auto t02 = std::chrono::steady_clock::now();
  while (el2 < 0.5 * seconds) {
    const auto a = std::chrono::steady_clock::now();
    poses[n2 % F].rootT[0] += 1e-6;
    cb.PrepareForEvaluation(true, true);
    const auto b = std::chrono::steady_clock::now();
    t_sweep2 += std::chrono::duration<double>(b - a).count();
    if (!evaluate_blocks()) return 1;
    ++n2;
    el2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t02).count();
  }
  // third measurement: where the block time goes — each kind of block alone, same thread pool, same cached sweep (no new sweep
  // in between: the blocks are served from the cache whatever the point); [4] = a pass over no blocks at all (the pool's
  // hand-shake alone)
  double kind_us[5] = {0, 0, 0, 0, 0};
  long kind_n[5] = {0, 0, 0, 0, 0};
  {
    const std::vector<size_t> all = order;            // sorted by kind
    for (int kd = 0; kd < 5; ++kd) {
      std::vector<size_t> sub;
      for (size_t i : all) if (kd < 4 && kind_rank(i) == kd) sub.push_back(i);
      kind_n[kd] = (long)sub.size();
      if (sub.empty() && kd < 4) continue;
      for (size_t i = 0; i < sub.size(); ++i) order[i] = sub[i];
      n_act = sub.size();
      const auto t_a = std::chrono::steady_clock::now();
      int reps = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count() < 0.1 * seconds) {
        if (!evaluate_blocks()) return 1;
        ++reps;
      }
      kind_us[kd] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count() / std::max(1, reps) * 1e6;
    }
    n_act = nrec;
  }
  quit.store(true);
  for (auto& th : pool) th.join();
  std::printf("{\"mode\": \"%s\", \"frames\": %d, \"blocks\": %d, \"block_threads\": %d, \"points\": %d, \"points_per_s\": %.1f, "
              "\"evals_per_s\": %.1f, \"blocks_per_s\": %.1f, \"sweep_with_copies_us\": %.1f, \"blocks_us_per_point\": %.1f, "
              "\"blocks_us_per_point_kind_by_kind\": %.1f, \"by_kind_us\": {\"reproj\": %.1f, \"pose_prior\": %.1f, "
              "\"shape_prior\": %.1f, \"temporal\": %.1f, \"empty_pass\": %.1f}, \"by_kind_blocks\": [%ld, %ld, %ld, %ld]}\n",
              mode.c_str(), F, n_blocks, threads, n, n / el, n / el * F, n / el * n_blocks, t_sweep / n * 1e6, (el - t_sweep) / n * 1e6,
              (el2 - t_sweep2) / std::max(1, n2) * 1e6, kind_us[0], kind_us[1], kind_us[2], kind_us[3], kind_us[4], kind_n[0], kind_n[1], kind_n[2], kind_n[3]);
  bodyfit_problem_destroy(bp);
  bodyfit_model_destroy(model);
  return 0;
}
