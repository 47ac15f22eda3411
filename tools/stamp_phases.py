"""Diagnostic: per-phase cycle shares of k_frame_resjac from in-kernel s_memtime stamps.
Build first:  make -C 3dbodyanimation_amd/csrc stamps ;  run with BODYFIT_LIB=.../libbodyfit_stamps.so"""
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
gm = api.Model(model)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, want_mesh=os.environ.get("STAMP_NO_MESH") is None)
lib = api.load_library()
buf = torch.zeros((1 << 20) + 216 * 8 * 16, dtype=torch.int64, device="cuda")   # the mesh kernel stamps behind 1 << 20
lib.bodyfit_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.bodyfit_debug_set_stamp_buffer(prob.h, buf.data_ptr())
x = torch.from_numpy(seq.gt_params + 0.01).cuda()
b = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
for _ in range(5):
    prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()[:F * 8 * 16].reshape(F, 8, 16)
t = raw[:, :, :9].astype(np.float64)
names = ["A tables", "B rodrigues/offsets", "C feat+chain walks", "C2 landmark rows", "D W/lmLBS/cam",
         "E mesh ops + lm jac terms", "F1 kp stage", "F2 jacobian sweep"]
d = np.diff(t, axis=2)  # [F,8,8]
print("phase durations in shader cycles (median over blocks; per wave 0..7):")
for i, n in enumerate(names):
    print(f"  {n:28s}", np.median(d[:, :, i], axis=0).astype(int), " max-wave median:", int(np.median(d[:, :, i].max(1))))
tot = t[:, :, 8].max(1) - t[:, :, 0].min(1)
print("block total (median):", int(np.median(tot)), "cycles")
start = t[:, :, 0].min(1); end = t[:, :, 8].max(1)
print("block duration cycles: min/median/p90/max", int(tot.min()), int(np.median(tot)), int(np.percentile(tot, 90)), int(tot.max()))
print("start spread (cycles, relative to first block): median/max", int(np.median(start - start.min())), int((start - start.min()).max()))
print("kernel span (first start -> last end):", int(end.max() - start.min()), "cycles")
real = (raw[:, 0, 11] - raw[:, 0, 10]).astype(np.float64)   # 100 MHz ticks
clk = (t[:, 0, 8] - t[:, 0, 0]) / real * 100e6
print("in-kernel clock (GHz): median", round(float(np.median(clk)) / 1e9, 3), " block wall (us): median", round(float(np.median(real)) / 100, 2))
rs = raw[:, 0, 10].astype(np.float64); re_ = raw[:, 0, 11].astype(np.float64)
print("kernel span by s_memrealtime (us):", (re_.max() - rs.min()) / 100, " start spread (us):", (rs.max() - rs.min()) / 100)
