"""frames/sec to convergence (the second half of BASELINE.json's metric) with the product's LM (bodyfit_solve):
  c2: 1 frame, pose + Sim3 only, L2 prior            (BASELINE configs[1])
  c3: 256 independent frames, --opt-shape, GMM on    (configs[2]), one batched solve
  c4: 128-frame sequence staged like src/main_multi_frame.cpp: anchors every 10th frame (shared beta), then
      windows of 20 / overlap 5 with the beta lock 1e5, 60 iterations (configs[3])
  c5: 1024-frame sequence through drivers.run_multi (103 anchors, 69 windows) on one GPU (configs[4] at N=1)"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
model = synth.make_model(0)
gm = api.Model(model)
out = {}

seq = synth.make_sequence(model, 1, seed=0, beta_fixed=True)
const = np.zeros(76, np.uint8)
for j in (10, 11, 22, 23):
    const[7 + 3 * (j - 1):10 + 3 * (j - 1)] = 1
prob = api.Problem.from_sequence(gm, seq, n_cols=76, use_shape=False, beta_pose=20.0)
prob.solve(seq.init_params, None, constant=const, independent=True, max_iters=100)
t0 = time.perf_counter()
x, _, s = prob.solve(seq.init_params, None, constant=const, independent=True, max_iters=100)
dt = time.perf_counter() - t0
out["c2"] = dict(frames=1, seconds=dt, frames_per_s=1 / dt, iterations=s[0].iterations, sweeps=s[0].n_sweeps)

F = 256
seq = synth.make_sequence(model, F, seed=1)
w, mu, cov = synth.make_gmm(0)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                 gmm=api.Gmm(w, mu, cov), beta_shape=30.0)
t0 = time.perf_counter()
x, b, s = prob.solve(seq.init_params, np.zeros((F, 10)), independent=True, max_iters=100)
dt = time.perf_counter() - t0
r, _, _ = prob.evaluate(x, b, False)
K = prob.layout.n_keypoints
out["c3"] = dict(frames=F, seconds=dt, frames_per_s=F / dt, max_iterations=max(q.iterations for q in s),
                 sweeps=s[0].n_sweeps, converged=sum(q.termination == 0 for q in s),
                 mean_px=float(np.sqrt((r[:2 * K].reshape(K, 2) ** 2).sum(1)).mean()))

F = 128
seq = synth.make_sequence(model, F, seed=2)
t0 = time.perf_counter()
anchors = list(range(0, F, 10))
class S: pass
def sub(ids):
    s_ = S(); offs = [0]; kid = []; uv = []
    for f in ids:
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        kid.append(seq.kp_id[k0:k1]); uv.append(seq.kp_uv[k0:k1]); offs.append(offs[-1] + k1 - k0)
    s_.kp_offset = np.array(offs, np.int32); s_.kp_id = np.concatenate(kid); s_.kp_uv = np.concatenate(uv)
    s_.intr = seq.intr; s_.R0 = seq.R0[ids]
    return s_
sa = sub(anchors)
pa = api.Problem.from_sequence(gm, sa, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
xa, beta, s1 = pa.solve(seq.init_params[anchors], np.zeros(10), independent=False, max_iters=1000,
                        scale_bounds=(-1e300, 1e300))
poses = seq.init_params.copy()
n_win = 0
for s0 in range(0, F, 15):
    e = min(s0 + 20, F)
    ids = list(range(s0, e))
    sw = sub(ids)
    pw = api.Problem.from_sequence(gm, sw, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=1e5, lambda_temporal=3.0)
    bw = beta.copy()
    xw, bw, s2 = pw.solve(poses[ids], bw, independent=False, max_iters=60, scale_bounds=(-1e300, 1e300))
    poses[ids] = xw
    n_win += 1
dt = time.perf_counter() - t0
out["c4"] = dict(frames=F, seconds=dt, frames_per_s=F / dt, anchors=len(anchors), stage1_iterations=s1[0].iterations,
                 windows=n_win)

if "--c5" in sys.argv:
    drivers = importlib.import_module("3dbodyanimation_amd.drivers")
    F = 1024
    seq = synth.make_sequence(model, F, seed=3)
    ks = drivers.KeypointSequence(seq.kp_offset, seq.kp_id, seq.kp_uv, [f"{i:06d}.json" for i in range(F)])
    t0 = time.perf_counter()
    res = drivers.run_multi(gm, ks, seq.intr)
    dt = time.perf_counter() - t0
    px = np.array([r[1] for r in res["log"]])
    out["c5"] = dict(frames=F, seconds=dt, frames_per_s=F / dt, stage1_iterations=res["stage1"].iterations,
                     stage1_sweeps=res["stage1"].n_sweeps, mean_px_fk=float(px[F // 10 + 1:].mean()))
print(json.dumps(out))
