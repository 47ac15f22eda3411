"""Diagnostic: duration of the GMM prior workgroups vs the frame workgroups inside one k_frame_resjac launch."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
F = 256
model = synth.make_model(0); seq = synth.make_sequence(model, F, seed=0); gm = api.Model(model)
w, mu, cov = synth.make_gmm(0)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                 gmm=api.Gmm(w, mu, cov), beta_shape=30.0)
lib = api.load_library()
buf = torch.zeros((F * 8 + 64) * 16, dtype=torch.int64, device="cuda")
lib.bodyfit_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.bodyfit_debug_set_stamp_buffer(prob.h, buf.data_ptr())
x = torch.from_numpy(seq.gt_params + 0.01).cuda(); b = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
for _ in range(5): prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
fr = raw[:F * 8].reshape(F, 8, 16); pr = raw[F * 8:F * 8 + 16]
f_start, f_end = fr[:, 0, 10], fr[:, 0, 11]
t0 = min(f_start.min(), pr[:, 0].min())
print("frame workgroups: start %.2f..%.2f us, end %.2f..%.2f us" % ((f_start.min()-t0)/100, (f_start.max()-t0)/100, (f_end.min()-t0)/100, (f_end.max()-t0)/100))
print("prior workgroups: start %.2f..%.2f us, end %.2f..%.2f us, duration median %.2f us" % ((pr[:,0].min()-t0)/100, (pr[:,0].max()-t0)/100, (pr[:,1].min()-t0)/100, (pr[:,1].max()-t0)/100, np.median(pr[:,1]-pr[:,0])/100))
