"""A/B of the one-launch sweep (k_sweep_roles) against the two-launch sweep on the C3 problem: per-dispatch durations
from the dispatches' own timestamps (bodyfit_profile_sweep) and wall time per step of back-to-back sweeps."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")


def run(F, one, iters=200):
    os.environ["BODYFIT_ONE_LAUNCH"] = "1" if one else "0"
    m = synth.make_model(0)
    gm = api.Model(m)
    seq = synth.make_sequence(m, F, seed=0)
    gmm = api.Gmm(*synth.make_gmm(0))
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True,
                                     beta_pose=20.0, gmm=gmm, beta_shape=30.0, want_mesh=True)
    dev = torch.device("cuda", 0)
    dx = torch.from_numpy(seq.gt_params + 0.01).to(dev)
    db = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    prof = prob.profile_sweep(dx.data_ptr(), db.data_ptr(), True, False, iters, st)
    for _ in range(20):
        prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e6
    return dict(F=F, one_launch=one, us_per_step=round(wall, 2), **{k: round(v * 1e3, 2) for k, v in prof.items()})


if __name__ == "__main__":
    Fs = [int(a) for a in sys.argv[1:]] or [256]
    for F in Fs:
        for one in (False, True, False, True):
            print(json.dumps(run(F, one)), flush=True)
