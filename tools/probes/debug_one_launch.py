"""Diagnostic: where the one-launch sweep's cloud differs from the two-launch sweep's (frames, vertex tiles), over several
back-to-back launches with changing parameters."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = synth.make_model(0)
gm = api.Model(m)
seq = synth.make_sequence(m, F, seed=F)
gmm = api.Gmm(*synth.make_gmm(0))
kw = dict(n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0, gmm=gmm, beta_shape=30.0, want_mesh=True)
os.environ["BODYFIT_ONE_LAUNCH"] = "1"
pf = api.Problem.from_sequence(gm, seq, **kw)
os.environ["BODYFIT_ONE_LAUNCH"] = "0"
pt = api.Problem.from_sequence(gm, seq, **kw)
rng = np.random.default_rng(F)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    x = seq.gt_params + 0.05 * rng.standard_normal(seq.gt_params.shape)
    beta = np.tile(seq.gt_beta, (F, 1)) + 0.3 * rng.standard_normal((F, 10))
    pf.evaluate(x, beta, True)
    _, cf = pf.forward(x, beta)
    _, ct = pt.forward(x, beta)
    d = np.abs(cf - ct).max(axis=2)          # [F][V]
    bad = d > 2e-6
    frames = np.where(bad.any(axis=1))[0]
    tiles = np.unique(np.where(bad.any(axis=0))[0] // 32)
    print(f"launch pair {it}: max |diff| {d.max():.2e}; frames with a difference: {frames[:40].tolist()} ({len(frames)});"
          f" vertex tiles: {tiles[:40].tolist()} ({len(tiles)}); bad vertices per bad frame: "
          f"{[int(bad[f].sum()) for f in frames[:10]]}")
