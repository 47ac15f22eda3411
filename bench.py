#!/usr/bin/env python3
"""bench.py — SMPL residual+Jacobian evaluations per second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames, inputs resident in HBM:
  frame role       f64 residuals + analytic Jacobian (FK joints + vertex landmarks) and the mesh operands
  mesh role        6890-vertex forward (blendshapes on MFMA + LBS)
  prior role       pose prior incl. the GMM sweep, shape prior, temporal
  The three roles are workgroups of ONE launch (k_sweep_roles: every role fits 128 VGPRs and 80 KiB of LDS, so a frame
  workgroup and a mesh workgroup share a CU and run at the same time; the mesh operands are handed over inside the launch).
  BODYFIT_ONE_LAUNCH=0 gives the two-launch form (k_frame_resjac, k_mesh_blend_lbs) for A/B runs
  [-> reduce_shared + RCCL all-reduce of 66 doubles for the shared-shape workload].
`python bench.py --gpus N` starts its N ranks itself (a child torch.distributed.run) when no launcher did; at every N the line
carries `value` = C3 (weak scaling, no collective) and `c5_strong` = configs[4] sharded over the same ranks (sweep with the
RCCL all-reduce, fit with the RCCL all-gathers).
One "eval" = all of that for one frame.

Workloads
  c3 (default)  256 independent frames per GPU, BODY_25 keypoints, --opt-shape (per-frame beta, 86
                columns), GMM prior on, mesh on.  BASELINE.json configs[2]; frames are independent, so
                N GPUs run N shards with no data-path collective ("scaling": "weak").
  c5            one multi-frame window of --window frames (default 1024) sharded over the GPUs, shared
                beta, L2 pose prior + temporal links, one all-reduce of [cost, g_beta, H_bb] per step
                ("scaling": "strong").  BASELINE.json configs[4].
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RECORD_OUT = sys.stdout   # where the JSON record goes (main() points it at a private copy of the original stdout)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0  # the guide's measured copy rate, reported beside the spec fraction
# algorithmic bytes (SURVEY.md §8d / BASELINE.md §4), f32 model tensors read once per launch
B_MODEL_MESH = 82_680 + 826_800 + 17_114_760 + 661_440      # v_template + shapedirs + posedirs + weights
B_MODEL_ALL = B_MODEL_MESH + 661_440                         # + dense J_regressor (whole pipeline figure)
B_FRAME_MESH = 82_680 + 24 * 12 * 4 + 217 * 4                # cloud out + skin transforms + blend coefficients
B_FRAME_ALL = 118_588                                        # params 608 + kps 500 + r 400 + J 34,400 + cloud 82,680


def cpu_baseline(synth, model, seq, F_sample, gmm_np, beta_pose=20.0, beta_shape=30.0):
    """The oracle (CPU restatement, kind 'port') on the host cores over a bounded sample of the same
    workload, the same residual blocks the GPU step evaluates: reference-like reprojection residual+Jacobian
    (stride-4 dual-number passes, threads over blocks), the pose prior blocks (GMM max-mixture when the workload
    has one) and the per-frame shape prior blocks (threads over blocks), plus the SMPL forward (threads over frames)."""
    from oracle import oracle
    om = oracle.OracleModel(model)
    og = oracle.OracleGmm(*gmm_np) if gmm_np is not None else None
    nthr = oracle.max_threads()
    class S: pass
    s = S()
    s.kp_offset = seq.kp_offset[:F_sample + 1]; s.kp_id = seq.kp_id; s.kp_uv = seq.kp_uv
    s.intr = seq.intr; s.R0 = seq.R0[:F_sample]
    x = seq.gt_params[:F_sample]; beta = np.tile(seq.gt_beta, (F_sample, 1))
    om.evaluate_batch(s, x[:8] if F_sample >= 8 else x, beta, 86, True, True, mode=1, nthreads=nthr)  # warm the pool
    # repeat the sample until ~10 s of CPU work have been timed (bounded: at most 40 passes)
    t_ad = t_an = t_fw = t_pr = 0.0
    passes = 0
    while passes < 40 and (t_ad + t_fw + t_pr) < 10.0:
        t0 = time.perf_counter()
        om.evaluate_batch(s, x, beta, 86, True, True, mode=1, nthreads=nthr)
        t_ad += time.perf_counter() - t0
        t0 = time.perf_counter()
        om.evaluate_batch(s, x, beta, 86, True, True, mode=0, nthreads=nthr)
        t_an += time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.priors_batch(og, beta_pose, x, beta_shape, beta, want_jac=True, nthreads=nthr)
        t_pr += time.perf_counter() - t0
        t0 = time.perf_counter()
        om.forward_batch(x, beta, s.R0, nthreads=nthr)
        t_fw += time.perf_counter() - t0
        passes += 1
    n = F_sample * passes
    # the thread counts the reference itself uses (options.num_threads = 4 / 8: include/Sim3BA.h:476,
    # include/MultiFrameBA.h:148) and one core, one pass of a smaller sample each (~1-2 s per entry)
    ladder = {}
    Fl = min(F_sample, 32)
    sl = S()
    sl.kp_offset = seq.kp_offset[:Fl + 1]; sl.kp_id = seq.kp_id; sl.kp_uv = seq.kp_uv
    sl.intr = seq.intr; sl.R0 = seq.R0[:Fl]
    for nt in (1, 4, 8):
        if nt > nthr:
            continue
        t0 = time.perf_counter()
        om.evaluate_batch(sl, x[:Fl], beta[:Fl], 86, True, True, mode=1, nthreads=nt)
        oracle.priors_batch(og, beta_pose, x[:Fl], beta_shape, beta[:Fl], want_jac=True, nthreads=nt)
        om.forward_batch(x[:Fl], beta[:Fl], sl.R0, nthreads=nt)
        ladder[str(nt)] = Fl / (time.perf_counter() - t0)
    prior_kind = "GMM max-mixture pose prior" if og is not None else "L2 pose prior"
    return {"value": n / (t_ad + t_pr + t_fw), "unit": "evals/s", "cores": nthr, "kind": "port", "evals_per_s_by_threads": ladder,
            "sample": f"{passes} x {F_sample} frames of the bench workload, the blocks the GPU step evaluates: autodiff-style "
                      f"(stride-4 dual numbers) reprojection residual+Jacobian {t_ad:.2f}s + {prior_kind} and shape prior "
                      f"blocks {t_pr:.2f}s + f64 SMPL forward {t_fw:.2f}s, OpenMP over blocks/frames",
            "evals_per_s_analytic_jacobian": n / (t_an + t_pr + t_fw)}


def ceres_path(synth, model, seq, F, resident_evals_s):
    """The Ceres-kept path (north_star: 'the Ceres outer loop is kept'): tools/ceres_path_bench.cpp drives
    include/bodyfit_ceres.h the way ceres::Problem::Evaluate does — per evaluation point ONE device sweep through the
    EvaluationCallback (parameters up, residuals + Jacobian down) and every residual block's Evaluate — against the interface
    double of Ceres (tests/cpp/ceres_double; Ceres itself is not in the image).  C3 (this bench's frames, own beta each, mesh
    on) and one 20-frame C4 window."""
    import shutil
    import struct
    import subprocess
    import tempfile
    if shutil.which("g++") is None:
        return {"skipped": "no g++ on this box"}
    tmp = tempfile.mkdtemp(prefix="bodyfit_ceres_")
    exe = os.path.join(tmp, "ceres_path_bench")
    libdir = os.path.join(ROOT, "3dbodyanimation_amd")
    cc = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), "-I",
                         os.path.join(ROOT, "tests", "cpp", "ceres_double"), os.path.join(ROOT, "tools", "ceres_path_bench.cpp"),
                         "-o", exe, "-L", libdir, "-lbodyfit", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        return {"skipped": "compile failed: " + cc.stderr[-300:]}

    def blob(path, sq, n):
        with open(path, "wb") as f:
            f.write(struct.pack("7i", model.n_verts, 24, 10, 207, len(model.landmark_vid), n, int(sq.kp_offset[n])))
            for a in (model.v_template, model.shapedirs, model.posedirs, model.j_regressor, model.weights):
                f.write(np.ascontiguousarray(a, np.float64).tobytes())
            for a in (model.parent, model.landmark_vid, sq.kp_offset[:n + 1], sq.kp_id[:sq.kp_offset[n]]):
                f.write(np.ascontiguousarray(a, np.int32).tobytes())
            f.write(np.ascontiguousarray(sq.kp_uv[:sq.kp_offset[n]], np.float64).tobytes())
            f.write(np.ascontiguousarray(sq.intr, np.float64).tobytes())

    out = {"driver": "tools/ceres_path_bench.cpp over include/bodyfit_ceres.h; Ceres = interface double; blocks evaluated by 8 threads "
                     "(the reference's options.num_threads)", "resident_evals_per_s": resident_evals_s}
    for mode, n, secs in (("c3", F, 3.0), ("c4", 20, 2.0)):
        path = os.path.join(tmp, mode + ".bin")
        blob(path, seq, n)
        res = subprocess.run([exe, path, mode, str(secs), "8"], capture_output=True, text=True, timeout=300)   # 8: the reference's options.num_threads
        try:
            out[mode if mode == "c3" else "c4_window"] = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception:
            out[mode] = {"failed": (res.stdout + res.stderr)[-300:]}
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def c5_fit(args, api, synth, model, gm, dist, rank, world, local_rank):
    """BASELINE.json configs[4] as a FIT: one `--window`-frame multi-frame window (shared beta, L2 pose prior, temporal links;
    OptimizeMultiFrame, include/MultiFrameBA.h:33-177) fitted to convergence from the reference's initial state
    (src/main_multi_frame.cpp:96-100), its frames sharded over the ranks (bodyfit_solve_sharded: cyclic reduction per shard,
    interface system all-gathered, the beta terms all-reduced once per LM iteration).  One step = one complete fit."""
    import torch
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    Fw = args.window
    full = synth.make_sequence(model, Fw, seed=0)
    shard = sharded.make_shard(Fw, world, rank)
    sl = sharded.slice_sequence(full, shard)
    prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86, use_shape=True,
                       beta_pose=5.0, beta_shape=25.0 if shard.owns_shape_prior else 0.0, lambda_temporal=3.0,
                       temporal_halo=shard.halo)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    comm = rccl = None
    if world > 1:
        if backend == "nccl":      # the exchanges as RCCL all-gathers on the solve's device buffers and stream (over xGMI)
            rccl = sharded.make_rccl(api, dist, rank, world, local_rank)
        else:                      # rehearsal on a one-GPU box: the host-callback transport over gloo
            comm = sharded.TorchComm(api, dist, rank, world, device=None)
    x0 = sharded.local_params(full.init_params, shard)
    max_iters = 1000   # the reference's max_iters_s1 (src/main_multi_frame.cpp:29)

    def fit():
        if rccl is not None:
            return prob.solve_sharded_rccl(x0, np.zeros(10), rccl, max_iters=max_iters)
        if world > 1:
            return prob.solve_sharded(x0, np.zeros(10), comm.c, max_iters=max_iters)
        x, b, s = prob.solve(x0, np.zeros(10), independent=False, max_iters=max_iters, scale_bounds=(-1e300, 1e300), solver=3)
        return x, b, s[0]

    for _ in range(args.warmup):
        fit()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, b, summ = fit()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the Jacobian sweep of the shard (the evaluator inside the fit), timed with HIP events for the roofline object
    dev = torch.device("cuda", local_rank)
    d_params = torch.from_numpy(np.ascontiguousarray(x0)).to(dev)
    d_beta = torch.zeros(10, dtype=torch.float64, device=dev)
    prof = prob.profile_sweep(d_params.data_ptr(), d_beta.data_ptr(), True, False, 50, torch.cuda.current_stream().cuda_stream)
    if rank == 0:
        F = shard.n_local
        alg = F * (608 + 80 + 500 + 400 + 34_400)
        a = alg / (prof["frame_resjac"] * 1e-3) / 1e9
        ms_fit = dt / args.steps * 1e3
        out = {
            "metric": "frames/sec to convergence (multi-frame window, shared beta)", "value": Fw * args.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_fit, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C5 fit: one {Fw}-frame multi-frame window (3dba_multi residual blocks: 25 keypoints per frame, "
                                   f"shared beta, L2 pose prior 5, shape prior 25, temporal 3) fitted to convergence, frames sharded "
                                   f"over {world} GPU(s)", "window": Fw, "frames_per_gpu": F, "keypoints_per_frame": 25, "n_cols": 86},
            "fit": {"iterations": summ.iterations, "successful": summ.n_successful, "sweeps": summ.n_sweeps,
                    "termination": summ.termination, "initial_cost": summ.initial_cost, "final_cost": summ.final_cost,
                    "ms_per_iteration": ms_fit / max(1, summ.iterations),
                    "exchanges_per_iteration": 0 if world == 1 else "3 all-gathers on device buffers: interface blocks + beta terms (225 KB per rank), beta Schur partials (110 doubles), scalars (8 doubles)",
                    "exchanges_total": 0 if world == 1 else prob.last_exchange_count(),
                    "transport": None if world == 1 else ("RCCL ncclAllGather on the solve's stream" if rccl is not None else "host callbacks over gloo (rehearsal)")},
            "roofline": {"bound": "hbm", "kernel": "frame_resjac", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": a / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": prof["frame_resjac"],
                         "note": "the Jacobian sweep of one shard inside the fit (no mesh); a fit iteration is latency-bound: "
                                 "cyclic-reduction levels of 76 x 76 f64 block factorisations"},
        }
        print(json.dumps(out), file=RECORD_OUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


def self_launch(n_gpus):
    """The N > 1 form of the driver's contract when no launcher set the rank environment: one child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`;
    rank 0's JSON line is relayed on stdout, everything else the ranks print goes to stderr, the exit code is the child's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool's hosts
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if line is not None else 1))


def c5_strong(args, api, synth, model, gm, dist, rank, world, local_rank):
    """BASELINE.json configs[4] on THIS run's ranks, reported beside the C3 value in the same line: one 1024-frame multi-frame
    window (shared beta, L2 pose prior, temporal links: OptimizeMultiFrame, include/MultiFrameBA.h:33-177) sharded over the N
    ranks by frame.
      sweep  the evaluation every LM iteration needs: the shard's residual + Jacobian + mesh sweep, the shared-beta reduction at the
             sweep's own tail (or one reduce launch for shards of more than 256 frames), and ONE ncclAllReduce(sum, f64) of the 66
             doubles [cost | g_beta | H_bb] issued by the library on the same stream (bodyfit_allreduce_shared_rccl): no host
             synchronisation and no Python hop between sweep and collective
      fit    the window fitted to convergence from the reference's initial state by bodyfit_solve_sharded_rccl (three
             ncclAllGather per LM iteration on the solve's device buffers)
    The communicator is the library's own (ncclCommInitRank from an id rank 0 draws); `rccl_ranks` is what RCCL itself reports
    for it (ncclCommCount).  Rehearsals on a one-GPU box (BENCH_BACKEND=gloo, every rank on cuda:0) use torch.distributed's
    gloo for the collective and the host-callback transport for the fit, and say so."""
    import torch
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    Fw = args.window
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank)
    full = synth.make_sequence(model, Fw, seed=0)
    shard = sharded.make_shard(Fw, world, rank)
    sl = sharded.slice_sequence(full, shard)
    kw = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0 if shard.owns_shape_prior else 0.0, lambda_temporal=3.0,
              temporal_halo=shard.halo)
    prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], want_mesh=True, **kw)
    rccl = sharded.make_rccl(api, dist, rank, world, local_rank) if backend == "nccl" else None
    rccl_ranks = rccl.count()[0] if rccl is not None else None
    d_params = torch.from_numpy(np.ascontiguousarray(full.gt_params[shard.f0:shard.f1 + (1 if shard.halo else 0)] + 0.01)).to(dev)
    d_beta = torch.from_numpy(np.ascontiguousarray(full.gt_beta + 0.01)).to(dev)
    d_red = torch.zeros(66, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    prob.arm_shared_reduction(d_red.data_ptr())

    def step():
        prob.evaluate_device(d_params.data_ptr(), d_beta.data_ptr(), True, stream)
        prob.reduce_shared_device(d_red.data_ptr(), stream)
        if rccl is not None:
            rccl.allreduce_shared(d_red.data_ptr(), stream)
        elif world > 1:
            h = d_red.cpu()
            dist.all_reduce(h)
            d_red.copy_(h)

    def bracket(fn, n):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else None)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(args.warmup):
        step()
    dt_sweep, _ = bracket(step, args.steps)
    prob.sweep_status(stream)          # (asynchronous sweeps: their in-launch waits all came through)
    total = d_red.cpu().numpy().copy()
    # the fit: a problem without the mesh (the LM never reads it), same shard
    fprob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], **kw)
    comm = sharded.TorchComm(api, dist, rank, world, device=None) if (rccl is None and world > 1) else None
    x0 = sharded.local_params(full.init_params, shard)

    def fit():
        if rccl is not None and world > 1:
            return fprob.solve_sharded_rccl(x0, np.zeros(10), rccl, max_iters=1000)
        if world > 1:
            return fprob.solve_sharded(x0, np.zeros(10), comm.c, max_iters=1000)
        x, b, sm = fprob.solve(x0, np.zeros(10), independent=False, max_iters=1000, scale_bounds=(-1e300, 1e300), solver=3)
        return x, b, sm[0]

    fit()
    n_fit = 2
    dt_fit, (_, _, summ) = bracket(fit, n_fit)
    # ---- N = 1 only: ONE rank's critical path at the N = 8 geometry, measured alone on this GPU (bodyfit_set_shard_proxy) ----
    # A 128-frame shard (rank 3 of 8: a left and a right neighbour), its chain reduced with the ends pinned (7 local levels), the
    # 16-frame interface chain solved as every rank solves it, the three all-gathers per LM iteration and the evaluation's
    # all-reduce issued on the one-rank RCCL communicator (the gathered slots of the seven other ranks are copies of this shard's:
    # one small kernel per exchange).  The transport costs nothing here that a launch does not cost, so N = 1 time / proxy time is
    # an UPPER bound on the strong-scaling speed-up at N = 8 — measured, not projected.
    proxy = None
    if world == 1 and rccl is not None and Fw % 8 == 0 and Fw >= 64:
        Np, Rp = 8, 3
        sh = sharded.make_shard(Fw, Np, Rp)
        slp = sharded.slice_sequence(full, sh)
        kwp = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=0.0, lambda_temporal=3.0, temporal_halo=True)
        pfit = api.Problem(gm, slp["kp_offset"], slp["kp_id"], slp["kp_uv"], slp["intr"], slp["R0"], **kwp)
        pfit.set_shard_proxy(Np, Rp)
        x0p = sharded.local_params(full.init_params, sh)

        def pfit_run(iters):
            _, _, sm = pfit.solve_sharded_rccl(x0p, np.zeros(10), rccl, max_iters=iters)
            return sm
        its_conv = pfit_run(400).iterations        # (the replicated window converges somewhere: both timed caps stay below it)
        it_a = max(4, its_conv // 4)
        it_b = max(it_a + 8, (3 * its_conv) // 4)
        reps = []
        for _ in range(5):
            ta, sa = bracket(lambda: pfit_run(it_a), 1)
            tb, sb = bracket(lambda: pfit_run(it_b), 1)
            reps.append(((tb - ta) / (it_b - it_a), tb, pfit.last_exchange_count()))
        reps.sort()
        us_iter_proxy, t_b, n_ex = reps[len(reps) // 2][0] * 1e6, reps[len(reps) // 2][1], reps[len(reps) // 2][2]
        # the shard's evaluation step: sweep (mesh on) + reduction at the sweep's own tail + the one-rank all-reduce
        psw = api.Problem(gm, slp["kp_offset"], slp["kp_id"], slp["kp_uv"], slp["intr"], slp["R0"], want_mesh=True, **kwp)
        dpp = torch.from_numpy(np.ascontiguousarray(full.gt_params[sh.f0:sh.f1 + 1] + 0.01)).to(dev)
        dred2 = torch.zeros(66, dtype=torch.float64, device=dev)
        psw.arm_shared_reduction(dred2.data_ptr())

        def pstep():
            psw.evaluate_device(dpp.data_ptr(), d_beta.data_ptr(), True, stream)
            psw.reduce_shared_device(dred2.data_ptr(), stream)
            rccl.allreduce_shared(dred2.data_ptr(), stream)
        for _ in range(50):
            pstep()
        ts = sorted(bracket(pstep, 200)[0] for _ in range(5))
        psw.sweep_status(stream)
        us_step_proxy = ts[2] / 200 * 1e6
        us_iter_n1 = dt_fit / n_fit * 1e6 / max(1, summ.iterations)
        us_step_n1 = dt_sweep / args.steps * 1e6
        proxy = {
            "what": f"rank {Rp} of {Np}: a {sh.n_local}-frame shard of the {Fw}-frame window run ALONE on this GPU through the sharded "
                    "code path (bodyfit_set_shard_proxy): 7 local cyclic-reduction levels with pinned ends, the 16-frame interface "
                    "chain, three ncclAllGather per LM iteration and the evaluation's ncclAllReduce on the one-rank communicator, "
                    "the other ranks' gathered slots = copies of this shard's",
            "frames": sh.n_local, "n_ranks_emulated": Np,
            "us_per_lm_iteration": us_iter_proxy, "lm_iterations_timed": [it_a, it_b],
            "how": "(t(solve capped at b iterations) - t(capped at a)) / (b - a), median of 5: start-up and read-back cancel",
            "seconds_of_the_longer_solve": t_b, "exchanges_of_the_longer_solve": n_ex, "iterations_to_convergence": its_conv,
            "us_per_sweep_step": us_step_proxy,
            "n1_us_per_lm_iteration": us_iter_n1, "n1_us_per_sweep_step": us_step_n1,
            "fit_speedup_upper_bound_at_8": us_iter_n1 / us_iter_proxy,
            "sweep_speedup_upper_bound_at_8": us_step_n1 / us_step_proxy,
            "note": "upper bounds: the transport's xGMI latency (three dependent all-gathers per iteration, one all-reduce per "
                    "evaluation) is NOT in the proxy; the N = 1 figures are this run's own c5_strong.fit / .sweep",
        }
        pfit.close(); psw.close()
    out = None
    if rank == 0:
        out = {
            "workload": f"C5: one {Fw}-frame multi-frame window (25 keypoints per frame, shared beta, L2 pose prior 5, shape prior 25, "
                        f"temporal 3, mesh on in the sweep) sharded by frame over {world} GPU(s)",
            "scaling": "strong", "window": Fw, "n_gpus": world, "frames_per_gpu": shard.n_local,
            "rccl_ranks": rccl_ranks,
            "transport": ("RCCL (library-owned communicator): ncclAllReduce for the evaluation, ncclAllGather for the fit, on the "
                          "work's own stream") if rccl is not None else "torch.distributed gloo on host buffers (one-GPU rehearsal)",
            "sweep": {"evals_per_s": Fw * args.steps / dt_sweep, "us_per_step": dt_sweep / args.steps * 1e6, "steps": args.steps,
                      "collective": "one all-reduce (sum, f64) of 66 doubles per step" if world > 1 or rccl is not None else None,
                      "reduced_cost": float(total[0])},
            "fit": {"frames_per_s": Fw * n_fit / dt_fit, "seconds": dt_fit / n_fit, "iterations": summ.iterations,
                    "successful": summ.n_successful, "termination": summ.termination, "initial_cost": summ.initial_cost,
                    "final_cost": summ.final_cost, "ms_per_iteration": dt_fit / n_fit * 1e3 / max(1, summ.iterations),
                    # (four exchanges at the start; the host looks at the device's status every fourth iteration, so up to three
                    #  iterations are launched — and exchanged — beyond the last one the solve counts)
                    "exchanges_total": fprob.last_exchange_count() if world > 1 else 0,
                    "iterations_launched": ((fprob.last_exchange_count() - 4) // 3) if world > 1 else summ.iterations,
                    "exchanges_per_iteration": ((fprob.last_exchange_count() - 4) / max(1, (fprob.last_exchange_count() - 4) // 3))
                                               if world > 1 else 0},
        }
        if proxy is not None:
            out["shard_proxy"] = proxy
    if rccl is not None:
        rccl.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"])
    ap.add_argument("--frames-per-gpu", type=int, default=256)
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--prewarm", type=int, default=1000,
                    help="untimed sweeps in FRONT of the --warmup steps: a fresh process's first ~500 launches run 8-15 %% slower "
                         "(clock ramp, first-touch of the model: 23-25 us per step against 21 once warm, profiles/r4_y_variant_ab.txt); "
                         "the driver's --warmup 5 alone leaves the timed steps on that ramp.  Reported as prewarm_steps")
    ap.add_argument("--prewarm-rest-ms", type=float, default=0.0,
                    help="idle time in front of each timed bracket's warm-up steps (round 4 tuned 2 ms against the clock governor; "
                         "the median over --repeats brackets makes it unnecessary: default 0)")
    ap.add_argument("--repeats", type=int, default=9,
                    help="how many times the contract's bracket (barrier + synchronize, K steps, synchronize + barrier) is timed; "
                         "ms_per_step = the median bracket, min / max are reported beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="(kept for old command lines: the host-pointer rate is always reported)")
    ap.add_argument("--no-ceres-path", action="store_true", help="skip the Ceres-kept-path record (needs g++ on the box)")
    ap.add_argument("--cpu-sample-frames", type=int, default=0)
    ap.add_argument("--no-fit", action="store_true", help="skip the frames/sec-to-convergence record")
    ap.add_argument("--no-c5-strong", action="store_true",
                    help="skip the c5_strong object (configs[4] sharded over this run's ranks: sweep with the RCCL all-reduce, fit)")
    ap.add_argument("--fit", action="store_true",
                    help="with --workload c5: a step is one complete LM fit of the window (frames/sec to convergence), sharded "
                         "over the GPUs by bodyfit_solve_sharded; defaults then to --steps 5 --warmup 1")
    args = ap.parse_args()
    if args.fit and args.workload == "c5":
        if "--steps" not in sys.argv:
            args.steps = 5
        if "--warmup" not in sys.argv:
            args.warmup = 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one process per
        # GPU, RCCL over xGMI) and relay its line.  Nothing in this process has touched the GPU yet (not even `import torch`),
        # and it never does: a process that has initialised the GPU must not be replaced or forked into ranks.
        return self_launch(args.gpus)

    # stdout carries ONE line, the JSON record.  Libraries write there too (RCCL prints a version banner on stdout when a
    # communicator is created): from here on file descriptor 1 points at stderr and the record goes out through a private copy
    # of the original stdout.
    global RECORD_OUT
    sys.stdout.flush()
    RECORD_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without a launcher at all: bench.py then starts its ranks itself)")
    # rehearsal switches (tests only): BENCH_SHARE_DEVICE0=1 puts every rank on cuda:0 and BENCH_BACKEND=gloo replaces
    # RCCL, so the N > 1 code path can be exercised on a one-GPU box; the driver's runs use neither
    if os.environ.get("BENCH_SHARE_DEVICE0"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    model = synth.make_model(0)
    gm = api.Model(model, device=local_rank)

    if args.workload == "c5" and args.fit:
        return c5_fit(args, api, synth, model, gm, dist, rank, world, local_rank)

    if args.workload == "c3":
        F = args.frames_per_gpu
        seq = synth.make_sequence(model, F, seed=rank)
        w, mu, cov = synth.make_gmm(0)
        gmm = api.Gmm(w, mu, cov, device=local_rank)
        prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True,
                                         beta_pose=20.0, gmm=gmm, beta_shape=30.0, want_mesh=True)
        beta_h = np.tile(seq.gt_beta, (F, 1)) + 0.01
        params_h = seq.gt_params + 0.01
        total_frames = F * world
        scaling, with_reduce = "weak", False
        wl_name = (f"C3: {F} independent frames per GPU, BODY_25 (14 FK joints + 11 vertex landmarks), "
                   "opt-shape (per-frame beta, 86 cols), GMM prior on, 6890-vertex mesh on")
    else:
        total_frames = args.window
        sharded = importlib.import_module("3dbodyanimation_amd.sharded")
        shard = sharded.make_shard(total_frames, world, rank)
        f0, f1, halo = shard.f0, shard.f1, shard.halo
        full = synth.make_sequence(model, total_frames, seed=0)
        F = f1 - f0
        sl = sharded.slice_sequence(full, shard)
        prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86,
                           use_shape=True, beta_per_frame=False, pose_blend=True, beta_pose=5.0,
                           beta_shape=25.0 if shard.owns_shape_prior else 0.0, lambda_temporal=3.0,
                           temporal_halo=halo, want_mesh=True)
        seq = full
        params_h = full.gt_params[f0:f1 + (1 if halo else 0)] + 0.01
        beta_h = full.gt_beta + 0.01
        scaling, with_reduce = "strong", True
        wl_name = (f"C5: one {total_frames}-frame multi-frame window sharded over {world} GPU(s), shared beta, "
                   "L2 pose prior + temporal, mesh on, all-reduce of [cost,g_beta,H_bb] (66 f64) per step")

    dev = torch.device("cuda", local_rank)
    d_params = torch.from_numpy(np.ascontiguousarray(params_h)).to(dev)
    d_beta = torch.from_numpy(np.ascontiguousarray(beta_h)).to(dev)
    d_red = torch.zeros(66, dtype=torch.float64, device=dev)
    # a non-default stream for the sweep, the reduction and torch's collective
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream

    rccl = None
    if with_reduce:
        prob.arm_shared_reduction(d_red.data_ptr())   # shards of <= 256 frames: the reduction rides on the sweep's tail
        if world > 1 and os.environ.get("BENCH_BACKEND", "nccl") == "nccl":
            rccl = sharded.make_rccl(api, dist, rank, world, local_rank)

    def step():
        prob.evaluate_device(d_params.data_ptr(), d_beta.data_ptr(), True, stream)
        if with_reduce:
            prob.reduce_shared_device(d_red.data_ptr(), stream)
            if rccl is not None:      # ncclAllReduce issued by the library on the sweep's stream (bodyfit_allreduce_shared_rccl)
                rccl.allreduce_shared(d_red.data_ptr(), stream)
            elif world > 1:           # one-GPU rehearsal over gloo
                h = d_red.cpu()
                dist.all_reduce(h)
                d_red.copy_(h)

    def bracket(n):
        """the contract's timed region: barrier + synchronize, EXACTLY n steps, synchronize + barrier; MAX over ranks"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # (1) the contract read literally, on a process that has done nothing yet: W warm-up steps, K timed steps.  Reported as
    # `cold_start` beside the headline (a fresh process's first ~500 launches run on the clock ramp: 23-25 us per step)
    for _ in range(args.warmup):
        step()
    dt_cold = bracket(args.steps)
    # (2) untimed sweeps until the clock (and the caches' view of the model) have settled ...
    for i in range(args.prewarm):
        prob.evaluate_device(d_params.data_ptr(), d_beta.data_ptr(), True, stream)
        if i % 256 == 255:
            torch.cuda.synchronize()      # (keeps the queue short; nothing is timed here)
    torch.cuda.synchronize()
    # (3) ... then the contract's bracket `--repeats` times (W warm-up steps in front of each); `ms_per_step` is the MEDIAN bracket,
    # the spread goes into the line.  No governor-tuned rest any more: a 0.4 ms timed region read once depends on where the
    # clock governor happens to be (one run in four read 23 us in round 4), the median of seven does not
    dts = []
    for _ in range(max(1, args.repeats)):
        if args.prewarm_rest_ms > 0:
            time.sleep(args.prewarm_rest_ms * 1e-3)
        for _ in range(args.warmup):
            step()
        dts.append(bracket(args.steps))
    dts_sorted = sorted(dts)
    dt = dts_sorted[len(dts_sorted) // 2] if len(dts_sorted) % 2 else 0.5 * (dts_sorted[len(dts_sorted) // 2 - 1] + dts_sorted[len(dts_sorted) // 2])
    # every timed sweep was an asynchronous one-launch sweep: did all their in-launch waits come through?  (a timed-out hand-off
    # leaves tiles of the cloud unwritten in a FASTER step.)  The run fails when one did not; the count goes into the line.
    prob.sweep_status(stream)
    sweep_timeouts = prob.sweep_timeouts()

    # per-kernel durations with HIP events on the launch stream (same inputs, same stream), right behind the timed steps — the
    # same warm device — and over at least 100 launches
    prof = prob.profile_sweep(d_params.data_ptr(), d_beta.data_ptr(), True, with_reduce, max(100, min(400, args.steps)), stream)
    # ... and the kernel's launch PERIOD, by two HIP events on the same stream around a run of back-to-back sweeps: what a launch
    # costs in steady state (the dispatch's own begin -> end, which profile_sweep reads, plus the dependent-launch boundary of
    # ~1.2 us).  This is the figure all three clocks reproduce — the K-step timing above, these events, and rocprofv3's kernel
    # trace of the same command (profiles/<tag>_kernel_stats_post_prewarm.json), whose per-dispatch durations under the tracer
    # read as the period, not as the bare begin -> end (r5_00: 20.23 us traced against 18.86 us begin -> end and a 20.03 us
    # period untraced) — so `roofline` is computed from it; the bare figure stays beside it as `kernel_only`.
    n_ev = max(100, min(400, args.steps))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        step()
    ev0.record(work_stream)
    for _ in range(n_ev):
        step()
    ev1.record(work_stream)
    torch.cuda.synchronize()
    period_ms = ev0.elapsed_time(ev1) / n_ev
    prob.sweep_status(stream)
    sweep_timeouts = prob.sweep_timeouts()

    # configs[4] on the same ranks, in the same line (default workload only: `value` stays the C3 figure at every N)
    strong = None
    if args.workload == "c3" and not args.no_c5_strong:
        strong = c5_strong(args, api, synth, model, gm, dist, rank, world, local_rank)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        evals_s = total_frames * args.steps / dt
        # algorithmic bytes per launch: what each kernel must read and write once
        alg = {"mesh_blend_lbs": B_MODEL_MESH + F * B_FRAME_MESH,
               "frame_resjac": F * (608 + 80 + 500 + 400 + 34_400 + 24 * 12 * 4 + 217 * 4)}
        pmc_name = {"mesh_blend_lbs": "k_mesh_blend_lbs", "frame_resjac": "k_frame_resjac", "sweep_roles": "k_sweep_roles"}
        fused = prof.get("sweep_roles", 0.0) > 0.0
        if fused:
            # the sweep was ONE launch (frame, mesh and prior roles side by side): the kernel IS the pipeline, so its algorithmic
            # bytes are SURVEY.md 8d's B(F) = B_model + F B_frame exactly (49,705,648 B at 256 frames) — the same figure
            # `pipeline` divides by the step time.  (Rounds 2-3 summed the two launches' own figures here, which count the
            # in-launch hand-off — transforms + coefficients, 2,020 B per frame — on both sides and leave the dense J_regressor out:
            # +0.8 %.)
            alg = {"sweep_roles": B_MODEL_ALL + F * B_FRAME_ALL}
        # HBM bytes per launch: rocprofv3 cannot run inside this process, so `traffic` is what the committed --pmc passes
        # of THIS command measured (tools/profile_round.sh writes profiles/rN_xx_pmc_traffic.json: FETCH_SIZE doubled per the
        # gfx950 note + WRITE_SIZE); `traffic_source` names that file.  Null when no committed pass matches the workload
        # and size, or its kernel names are not the ones this build launches.
        pm = pm_src = None
        try:
            import glob
            pm_src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
            pm = json.load(open(pm_src))
            if not (args.workload == pm["workload"] and F == pm["frames_per_gpu"]):
                pm = None
        except Exception:
            pm = None
        kernels = {}
        for k in alg:
            # one launch per step (and no reduce launch): the launch period of the step's events IS this kernel's
            dur = period_ms if (fused and not with_reduce) else prof[k]
            a = alg[k] / (dur * 1e-3) / 1e9
            ko = alg[k] / (prof[k] * 1e-3) / 1e9
            kernels[k] = {"achieved": a, "frac": a / HBM_PEAK_GBS, "frac_of_measured_copy_rate": a / HBM_COPY_GBS,
                          "algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": dur,
                          "avg_launch_ms_definition": ("launch period: two HIP events around %d back-to-back launches on the launch "
                                                       "stream" % n_ev) if dur is period_ms else "dispatch begin -> end (HIP events)",
                          "kernel_only": {"avg_launch_ms": prof[k], "achieved": ko, "frac": ko / HBM_PEAK_GBS,
                                          "definition": "the dispatch's own begin -> end timestamps (hipExtLaunchKernelGGL events), "
                                                        "without the dependent-launch boundary"},
                          "traffic": pm["kernels"].get(pmc_name[k], {}).get("hbm_bytes") if pm else None}
        # matrix-pipe view of the mesh kernel: the blend contraction is 2 x 20670 x 217 flop per frame; it is
        # executed as three bf16 products per k-step on v_mfma_f32_32x32x16_bf16 (216 tiles x 14 k-steps x 9 MFMAs per 32 frames)
        alg_flop = 2.0 * 20670 * 217 * F
        exe_flop = 216 * ((F + 31) // 32) * 14 * 9 * 2.0 * 32 * 32 * 16
        mk = "sweep_roles" if fused else "mesh_blend_lbs"
        kernels[mk]["mfma"] = {
            "algorithmic_TFLOPs": alg_flop / (prof[mk] * 1e-3) / 1e12, "f32_mfma_peak_TFLOPs": 157.3,
            "executed_bf16_TFLOPs": exe_flop / (prof[mk] * 1e-3) / 1e12, "bf16_dense_peak_TFLOPs": 2500.0}
        dom = max(alg, key=lambda k: prof[k])
        bytes_launch, ach, traffic = alg[dom], kernels[dom]["achieved"], kernels[dom]["traffic"]
        whole = (B_MODEL_ALL + F * B_FRAME_ALL) / (ms_step * 1e-3) / 1e9
        out = {
            "metric": "SMPL residual+Jacobian evals/sec (6890v, 10 beta, 24 joints)",
            "value": evals_s, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            # how ms_per_step was taken: the contract's bracket (barrier + synchronize, K steps, synchronize + barrier; MAX over
            # ranks) `repeats` times behind `prewarm_steps` untimed sweeps, each with its W warm-up steps in front; ms_per_step and
            # value are the MEDIAN bracket.  cold_start: the same bracket as the process's very first work (W warm-up steps only)
            "timing": {"repeats": len(dts), "prewarm_steps": args.prewarm, "rest_ms_before_each_bracket": args.prewarm_rest_ms,
                       "ms_per_step_min": dts_sorted[0] / args.steps * 1e3, "ms_per_step_median": ms_step,
                       "ms_per_step_max": dts_sorted[-1] / args.steps * 1e3,
                       "ms_per_step_all": [d / args.steps * 1e3 for d in dts],
                       "cold_start": {"ms_per_step": dt_cold / args.steps * 1e3, "value": total_frames * args.steps / dt_cold,
                                      "note": "W warm-up + K timed steps as the first GPU work of the process (clock ramp, cold caches)"}},
            "sweep_timeouts": sweep_timeouts,
            "dtype": "f64 residual/Jacobian; f32 mesh (f32 + split-bf16 MFMA blend)", "data": "synthetic",
            "config": {"workload": wl_name, "frames_per_gpu": F, "keypoints_per_frame": 25, "n_cols": 86},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "frac_of_measured_copy_rate": ach / HBM_COPY_GBS, "traffic": traffic,
                         "traffic_source": (os.path.relpath(pm_src, ROOT) + " (rocprofv3 --pmc passes of this command, "
                                            "committed; not measured in this run)") if pm else None,
                         "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                         "avg_launch_ms_definition": kernels[dom]["avg_launch_ms_definition"],
                         "kernel_only": kernels[dom]["kernel_only"],
                         "algorithmic_bytes_definition": ("SURVEY.md 8d: B(F) = 19,347,120 (f32 model tensors, read once per launch) + "
                                                         "F x 118,588 (params, keypoints, r, J, posed vertices)") if fused else
                                                        "per kernel: the model tensors it reads + its per-frame inputs and outputs",
                         "kernels": kernels},
            "kernel_ms": prof,
            "pipeline": {"algorithmic_bytes_per_step": B_MODEL_ALL + F * B_FRAME_ALL, "achieved_GBps": whole,
                         "frac_of_hbm_peak": whole / HBM_PEAK_GBS, "frac_of_measured_copy_rate": whole / HBM_COPY_GBS},
        }
        if strong is not None:
            out["c5_strong"] = strong
        if world == 1:
            # the rate a kept ceres::Solve would see: host parameters up, one sweep, residuals + Jacobian down (page-locked
            # mirrors, the problem's own stream); never `value`
            # (cache_sweep: what bodyfit_ceres::SweepCallback::PrepareForEvaluation does — the sweep stays in the problem's
            #  page-locked cache for the blocks' Evaluate calls, the Jacobian goes down as its non-zero column blocks;
            #  ..._dense: the same with the dense [2K][86] panel copied out into the caller's numpy arrays)
            for _ in range(3):
                prob.cache_sweep(params_h, beta_h)
            t1 = time.perf_counter()
            for _ in range(50):
                prob.cache_sweep(params_h, beta_h)
            out["pcie_inclusive_evals_per_s"] = F * 50 / (time.perf_counter() - t1)
            for _ in range(3):
                prob.evaluate(params_h, beta_h, True)
            t1 = time.perf_counter()
            for _ in range(20):
                prob.evaluate(params_h, beta_h, True)
            out["pcie_inclusive_dense_evals_per_s"] = F * 20 / (time.perf_counter() - t1)
            if args.workload == "c3" and not args.no_ceres_path:
                out["ceres_path"] = ceres_path(synth, model, seq, F, evals_s)
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            n_cpu = args.cpu_sample_frames or min(F, 256)
            cseq = seq if args.workload == "c3" else synth.make_sequence(model, n_cpu, seed=0)
            out["cpu_baseline"] = cpu_baseline(synth, model, cseq, n_cpu, (w, mu, cov) if args.workload == "c3" else None,
                                               beta_pose=20.0 if args.workload == "c3" else 5.0,
                                               beta_shape=30.0 if args.workload == "c3" else 0.0)
        if not args.no_fit and world == 1 and args.workload == "c3":
            # the second half of BASELINE.json's metric: frames/sec to convergence of the product's LM on c2 / c3 / c4
            # (tools/fit_bench.py; reference hooks src/main_single_frame.cpp:234-249,265-269), and the checker's own fits
            # (oracle evaluator under the dense numpy LM) on a bounded sample beside them
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import fit_bench
            # every record: median of its repeats, final robustified cost per stage, launches and microseconds per LM iteration;
            # c4 and c5_staged both go through drivers.run_multi (the reference's staging incl. the write-back after every
            # solve); c5_window is configs[4] as ONE 1024-frame window (what --workload c5 --fit shards over the ranks)
            fit = {"unit": "frames/s", "c2": fit_bench.fit_c2(api, synth, model, gm),
                   "c3": fit_bench.fit_c3(api, synth, model, gm), "c4": fit_bench.fit_c4(api, synth, model, gm),
                   "window_20": fit_bench.fit_window(api, synth, model, gm, 20, 60),
                   "c5_staged": fit_bench.fit_c5(api, synth, model, gm), "c5_window": fit_bench.fit_c5_window(api, synth, model, gm)}
            if not args.no_cpu_baseline:
                # eight threads: what the reference itself configures (options.num_threads = 8, include/MultiFrameBA.h:148;
                # 4 in include/Sim3BA.h:476) — and what these small problems can use: 25 to 500 residual blocks per evaluation
                # on all 256 hardware threads of the box is ten times SLOWER (fork/join of the OpenMP pool per evaluation)
                fit["cpu_baseline"] = fit_bench.cpu_fit_baseline(synth, model, threads=min(8, os.cpu_count() or 8))
            out["fit"] = fit
        print(json.dumps(out), file=RECORD_OUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
