"""Probe: C3's 256 independent frames as ONE 256-frame problem on one stream against TWO 128-frame problems on two streams
(two sweeps in flight: the latency chain at the head of one overlaps the store-bound tail of the other)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
m = synth.make_model(0); gm = api.Model(m); gmm = api.Gmm(*synth.make_gmm(0))
def mk(F, seed):
    seq = synth.make_sequence(m, F, seed=seed)
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True,
                                     beta_pose=20.0, gmm=gmm, beta_shape=30.0, want_mesh=True)
    dx = torch.from_numpy(seq.gt_params + 0.01).cuda(); db = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
    return prob, dx, db
def run(parts, iters=400):
    sts = [torch.cuda.Stream() for _ in parts]
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            for (prob, dx, db), st in zip(parts, sts):
                prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st.cuda_stream)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / iters * 1e6)
    return res
one = [mk(256, 0)]
two = [mk(128, 0), mk(128, 1)]
four = [mk(64, i) for i in range(4)]
for name, parts in (("1 x 256", one), ("2 x 128", two), ("4 x 64", four), ("1 x 256", one), ("2 x 128", two)):
    r = run(parts)
    print(f"{name}: us per 256 frames {['%.2f' % x for x in r]}", flush=True)
