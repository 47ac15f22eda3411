"""Diagnostic: where k_lm_step spends its time (s_memtime stamps, diagnostic build).
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
seq = synth.make_sequence(model, F, seed=1)
gm = api.Model(model)
w, mu, cov = synth.make_gmm(0)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0, gmm=api.Gmm(w, mu, cov),
                                 beta_shape=30.0)
lib = api.load_library()
buf = torch.zeros(F * 16, dtype=torch.int64, device="cuda")
lib.bodyfit_debug_set_lm_stamp_buffer.argtypes = [C.c_void_p]
assert lib.bodyfit_debug_set_lm_stamp_buffer(buf.data_ptr()) == 0
x, b, s = prob.solve(seq.init_params, np.zeros((F, 10)), independent=True, max_iters=6)

torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(F, 16).astype(np.float64)
ok = raw[:, 7] > raw[:, 0]
d = np.diff(raw[ok, :8], axis=1)
names = ["Jhat staging (L2 -> LDS)", "Gram on MFMA", "priors (GMM block)", "gradient, scaling, damped system", "blocked Cholesky",
         "backward substitution", "step, model change, candidate"]
print(f"k_lm_step phases, shader cycles (median over {ok.sum()} frames of the last iteration):")
for i, n in enumerate(names):
    print(f"  {n:36s} {np.median(d[:, i]):9.0f}")
print("  total", np.median(raw[ok, 7] - raw[ok, 0]))
print("  staging: operands arrived + first LDS pass", np.median(raw[ok, 11] - raw[ok, 0]), " Gram: wave 0's own tiles",
      np.median(raw[ok, 13] - raw[ok, 1]), " gradient + scaling up to the tolerance test", np.median(raw[ok, 12] - raw[ok, 3]))
print("  Cholesky parts summed over the solve's iterations / iterations: (a) diagonal blocks, (b) panel solves, (c) trailing updates:",
      np.median(raw[ok, 8:11], axis=0) / max(1, int(max(q.iterations for q in s))))
