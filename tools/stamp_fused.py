"""Diagnostic: timeline of the one-launch sweep (k_sweep_fused) from in-kernel s_memrealtime stamps (10 ns ticks).
Build: make -C 3dbodyanimation_amd/csrc stamps ; run with BODYFIT_LIB=.../libbodyfit_stamps.so"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = synth.make_model(0)
seq = synth.make_sequence(model, F, seed=0)
w, mu, cov = synth.make_gmm(0)
gm = api.Model(model)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True, beta_pose=20.0,
                                 gmm=api.Gmm(w, mu, cov), beta_shape=30.0, want_mesh=True)
lib = api.load_library()
nblk = (6890 + 31) // 32
BASE, FBASE = 1 << 20, (1 << 20) + (1 << 16)
buf = torch.zeros(FBASE + 512 * 8, dtype=torch.int64, device="cuda")
lib.bodyfit_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.bodyfit_debug_set_stamp_buffer(prob.h, buf.data_ptr())
x = torch.from_numpy(seq.gt_params + 0.01).cuda()
b = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
for _ in range(5):
    prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
G = max(F, nblk + (F + 15) // 16)
fs = raw[FBASE:FBASE + G * 8].reshape(G, 8).astype(np.float64)
fr = raw[:F * 8 * 16].reshape(F, 8, 16).astype(np.float64)
ms = raw[BASE:BASE + nblk * 8 * 16].reshape(nblk, 8, 16).astype(np.float64)
t0 = fs[:, 0].min()
us = lambda a: (a - t0) / 100.0
def q(a):
    a = np.asarray(a).ravel()
    return f"min {a.min():6.2f}  med {np.median(a):6.2f}  max {a.max():6.2f}"
print("all times in us after the first workgroup's entry")
print("workgroup entry               ", q(us(fs[:, 0])))
print("frame part: first stamp       ", q(us(fr[:, 0, 10])))
print("frame part: hand-off published", q(us(fr[:, 0, 12])))
print("frame part: end               ", q(us(fr[:, :, 11].max(1))))
print("after frame part (+barrier)   ", q(us(fs[:F, 1])))
tiles = min(nblk, G)
print("counter complete (wait ends)  ", q(us(fs[:tiles, 2])))
print("mesh part entry               ", q(us(ms[:, :, 0])))
print("mesh: operands requested      ", q(us(ms[:, :, 1])))
print("mesh: blend done   waves 0-3  ", q(us(ms[:, :4, 2])), "  waves 4-7", q(us(ms[:, 4:, 2])))
print("mesh: skinning done waves 0-3 ", q(us(ms[:, :4, 3])), "  waves 4-7", q(us(ms[:, 4:, 3])))
print("workgroup end                 ", q(us(fs[:tiles, 3])))
cyc = np.diff(fr[:, :, :9], axis=2)
names = ["A tables", "B rodrigues/offsets", "C chain walks / landmark items", "C barrier", "D", "E", "hand-off + F1", "F2 sweep"]
print("frame part phases, shader cycles (median over frames of the slowest wave | per wave 0..7):")
for i, n in enumerate(names):
    print(f"  {n:32s} {int(np.median(cyc[:, :, i].max(1))):6d} |", np.median(cyc[:, :, i], axis=0).astype(int))
print("frame part total cycles (median):", int(np.median(fr[:, :, 8].max(1) - fr[:, :, 0].min(1))))
