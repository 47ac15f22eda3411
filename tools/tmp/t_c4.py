import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import numpy as np
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
fb = importlib.import_module("tools.fit_bench")
m = synth.make_model(0); gm = api.Model(m)
seq = synth.make_sequence(m, 128, seed=2)
ids = list(range(0, 20))
sw = fb.sub_sequence(seq, ids)
for rep in range(3):
    t0 = time.perf_counter()
    pw = api.Problem.from_sequence(gm, sw, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=1e5, lambda_temporal=3.0)
    t1 = time.perf_counter()
    xw, bw, s2 = pw.solve(seq.init_params[ids], np.zeros(10), independent=False, max_iters=60, scale_bounds=(-1e300, 1e300))
    t2 = time.perf_counter()
    xw, bw, s3 = pw.solve(seq.init_params[ids], np.zeros(10), independent=False, max_iters=60, scale_bounds=(-1e300, 1e300))
    t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms  first solve {1e3*(t2-t1):.2f} ms ({s2[0].iterations} it)  second solve {1e3*(t3-t2):.2f} ms  per-iteration {1e3*(t3-t2)/s3[0].iterations:.3f} ms")
