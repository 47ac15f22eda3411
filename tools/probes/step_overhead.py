"""How much of a K-step timing (synchronise, K back-to-back sweeps, synchronise) is fixed cost: t(K) = a + b K fitted over K = 1 ... 200."""
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
Ks = [1, 2, 5, 10, 20, 50, 100, 200]
res = {K: [] for K in Ks}
for rep in range(15):
    for K in Ks:
        torch.cuda.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
        torch.cuda.synchronize()
        res[K].append((time.perf_counter() - t0) * 1e6)
med = np.array([np.median(res[K]) for K in Ks]); mn = np.array([np.min(res[K]) for K in Ks])
for K, a, b in zip(Ks, med, mn): print(f"K={K:4d}: median {a:8.1f} us ({a / K:6.2f} per step)   min {b:8.1f} us ({b / K:6.2f} per step)")
A = np.vstack([np.ones(len(Ks)), Ks]).T
(a, b), *_ = np.linalg.lstsq(A, med, rcond=None)
print(f"fit of the medians: t(K) = {a:.1f} us + {b:.2f} us x K")
