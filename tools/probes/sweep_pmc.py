"""Driver for rocprofv3 --pmc passes over the sweep's kernels at a batch of argv[1] frames, in the one-launch (argv[2] = one) or
the two-launch (two) form: a few sweeps, nothing else.  tools/probes/sweep_pmc.sh runs the passes."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
F = int(sys.argv[1]); form = sys.argv[2]
os.environ["BODYFIT_ONE_LAUNCH"] = "0" if form == "two" else "1"
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
m = synth.make_model(0); gm = api.Model(m); seq = synth.make_sequence(m, F, seed=0); gmm = api.Gmm(*synth.make_gmm(0))
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True, beta_pose=20.0, gmm=gmm,
                                 beta_shape=30.0, want_mesh=True)
dx = torch.from_numpy(seq.gt_params + 0.01).cuda(); db = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
st = torch.cuda.current_stream().cuda_stream
for _ in range(6): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
torch.cuda.synchronize()
