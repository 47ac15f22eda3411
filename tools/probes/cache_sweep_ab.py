"""PCIe-inclusive sweep (bodyfit_evaluate_batch into the problem's page-locked cache, what bodyfit_ceres::SweepCallback does per
evaluation point) with the parameters read by the sweep straight from the pinned mirrors (default) against copied up front
(BODYFIT_HOST_PARAMS=0): one child process per variant, interleaved."""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
    m = synth.make_model(0); gm = api.Model(m); F = 256
    seq = synth.make_sequence(m, F, seed=0); gmm = api.Gmm(*synth.make_gmm(0))
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True, beta_pose=20.0, gmm=gmm,
                                     beta_shape=30.0, want_mesh=True)
    x = seq.gt_params + 0.01; b = np.tile(seq.gt_beta, (F, 1)) + 0.01
    for _ in range(20): prob.cache_sweep(x, b)
    ts = []
    for rep in range(7):
        t0 = time.perf_counter()
        for _ in range(100): prob.cache_sweep(x, b)
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
    print(f"{np.median(ts):.1f}")
    sys.exit(0)
for rnd in range(3):
    for hp in ("1", "0"):
        env = dict(os.environ, BODYFIT_HOST_PARAMS=hp)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print(f"round {rnd} BODYFIT_HOST_PARAMS={hp}: {out.stdout.strip()} us per cached sweep of 256 frames {out.stderr[-200:] if out.returncode else ''}", flush=True)
