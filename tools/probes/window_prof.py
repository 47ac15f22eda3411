import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = synth.make_model(0)
gm = api.Model(model)
seq = synth.make_sequence(model, F, seed=5)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
for _ in range(2):
    x, b, s = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=int(os.environ.get("WP_ITERS", "60")), scale_bounds=(-1e300, 1e300), solver=3)
print(s[0].iterations)
