"""Per-iteration wall time of the window LM: host loop (solver 1) vs device-resident loop (solver 3)."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import fit_bench
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
model = synth.make_model(0)
gm = api.Model(model)
out = {}
for F, iters in ((20, 60), (103, 30), (256, 20), (1024, 10)):
    seq = synth.make_sequence(model, F, seed=5)
    rec = {}
    for solver in (1, 3):
        prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
        prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=2, scale_bounds=(-1e300, 1e300), solver=solver)
        t0 = time.perf_counter()
        x, b, s = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=iters, scale_bounds=(-1e300, 1e300),
                             solver=solver)
        dt = time.perf_counter() - t0
        rec["host" if solver == 1 else "device"] = dict(seconds=dt, iterations=s[0].iterations, ms_per_iteration=1e3 * dt / max(1, s[0].iterations),
                                                        final_cost=s[0].final_cost)
    out[f"F{F}"] = rec
out["c4_staged"] = fit_bench.fit_c4(api, synth, model, gm)
print(json.dumps(out, indent=1))
