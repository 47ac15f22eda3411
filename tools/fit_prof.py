"""One fit of the kind named on the command line, for rocprofv3 --kernel-trace --stats (tools/profile_fits.sh):
  c3        256 independent single-frame fits (own beta, GMM prior), device batched LM
  c4        drivers.run_multi over 128 frames (103-anchor-style stage 1 at 13 anchors, windows of 20)
  window N  one shared-beta window of N frames, device window LM (N = 20: a C4 window; N = 545: the C5 window of one GPU)"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import fit_bench

api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
model = synth.make_model(0)
gm = api.Model(model)
kind = sys.argv[1]
if kind == "c3":
    rec = fit_bench.fit_c3(api, synth, model, gm, repeats=1)
elif kind == "c4":
    rec = fit_bench.fit_c4(api, synth, model, gm, repeats=1)
else:
    F = int(sys.argv[2])
    rec = fit_bench.fit_window(api, synth, model, gm, F, 60 if F <= 64 else 30, repeats=1)
print({k: v for k, v in rec.items() if k != "note"})
