"""Quick timing of the LM kernels' consumers: C3 fit (256 independent frames), one 20-frame window, the 1024-frame window."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import fit_bench
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
model = synth.make_model(0)
gm = api.Model(model)
pick = lambda d: {k: d[k] for k in ("frames", "seconds", "frames_per_s", "iterations", "us_per_iteration", "launches_per_iteration") if k in d}
print(json.dumps({"c3": pick(fit_bench.fit_c3(api, synth, model, gm)),
                  "window_20": pick(fit_bench.fit_window(api, synth, model, gm, 20, 60)),
                  "c5_window": pick(fit_bench.fit_c5_window(api, synth, model, gm))}))
