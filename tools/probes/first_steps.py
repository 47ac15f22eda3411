"""Are the first launches behind a device synchronisation slower?  Average dispatch duration (the dispatches' own timestamps) of the
first N sweeps after an idle gap, N = 5, 20, 200."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
m = synth.make_model(0); gm = api.Model(m); seq = synth.make_sequence(m, 256, seed=0); gmm = api.Gmm(*synth.make_gmm(0))
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True, beta_pose=20.0, gmm=gmm,
                                 beta_shape=30.0, want_mesh=True)
dx = torch.from_numpy(seq.gt_params + 0.01).cuda(); db = torch.from_numpy(np.tile(seq.gt_beta, (256, 1))).cuda()
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws); st = ws.cuda_stream
for _ in range(1500): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
torch.cuda.synchronize()
for gap_ms in (0.0, 0.1, 1.0, 10.0):
    out = []
    for N in (5, 20, 200):
        v = []
        for rep in range(7):
            for _ in range(300): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
            torch.cuda.synchronize()
            if gap_ms: time.sleep(gap_ms * 1e-3)
            v.append(prob.profile_sweep(dx.data_ptr(), db.data_ptr(), True, False, N, st)["sweep_roles"] * 1e3)
        out.append(f"first {N:3d}: {np.median(v):6.2f} us")
    print(f"idle gap {gap_ms:5.1f} ms | " + " | ".join(out), flush=True)
