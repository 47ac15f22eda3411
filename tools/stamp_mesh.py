"""Diagnostic: where a k_mesh_blend_lbs wave spends its time (s_memrealtime stamps, 10 ns ticks).
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
gm = api.Model(model)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, want_mesh=True)
lib = api.load_library()
nblk = (6890 + 31) // 32
BASE = 1 << 20
buf = torch.zeros(BASE + nblk * 8 * 16, dtype=torch.int64, device="cuda")
lib.bodyfit_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.bodyfit_debug_set_stamp_buffer(prob.h, buf.data_ptr())
x = torch.from_numpy(seq.gt_params + 0.01).cuda()
b = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
for _ in range(5):
    prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()[BASE:BASE + nblk * 8 * 16].reshape(nblk, 8, 16).astype(np.float64)
nft = (F + 31) // 32
raw = raw[:, :min(8, nft), :]
d = np.diff(raw[:, :, :4], axis=2) / 100.0   # us
cyc = np.diff(raw[:, :, 8:12], axis=2)
names = ["B staging (HBM->LDS) + barrier", "blend phase (14 k-steps) [last unit]", "skinning phase (16 rows) [last unit]"]
print("per-wave segment times, us (median over blocks and waves / max) and shader-clock cycles:")
for i, n in enumerate(names):
    print(f"  {n:40s} {np.median(d[:, :, i]):7.2f} {d[:, :, i].max():7.2f}   {np.median(cyc[:, :, i]):9.0f} cyc")
if raw.shape[1] == 8:
    for nm, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        t0 = raw[:, :1, 1]
        print(f"  {nm}: blend ends {np.median(raw[:, sl, 2] - t0) / 100:6.2f} us, skinning ends {np.median(raw[:, sl, 3] - t0) / 100:6.2f} us after the barrier")
span = raw[:, :, 3].max() - raw[:, :, 0].min()
print("kernel span (us):", span / 100.0, " block wall median (us):", np.median(raw[:, :, 3].max(1) - raw[:, :, 0].min(1)) / 100.0)
print("block start spread (us):", (raw[:, 0, 0].max() - raw[:, 0, 0].min()) / 100.0)
