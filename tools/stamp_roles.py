"""Diagnostic: timeline of the one-launch sweep (k_sweep_roles) from in-kernel s_memrealtime stamps (10 ns ticks).
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
nvt = (6890 + 31) // 32
nG = (F + 255) // 256
BASE = 1 << 20
CYC = BASE + (1 << 19)
buf = torch.zeros(CYC + nG * nvt * 8 * 40 + 64, dtype=torch.int64, device="cuda")
lib.bodyfit_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.bodyfit_debug_set_stamp_buffer(prob.h, buf.data_ptr())
x = torch.from_numpy(seq.gt_params + 0.01).cuda()
b = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).cuda()
for _ in range(5):
    prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
fr = raw[:F * 8 * 16].reshape(F, 8, 16).astype(np.float64)
ms = raw[BASE:BASE + nG * nvt * 8 * 16].reshape(nG * nvt, 8, 16).astype(np.float64)
nC = 2 * ((F + 31) // 32)
cf = raw[BASE - 8192:BASE - 8192 + nC * 16].reshape(nC, 16).astype(np.float64)
t0 = min(fr[:, 0, 10].min(), ms[:, :, 0].min())
us = lambda a: (a - t0) / 100.0
def q(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return f"min {a.min():6.2f}  med {np.median(a):6.2f}  max {a.max():6.2f}"
print(f"F = {F}; all times in us after the first workgroup's entry")
print("frame role: entry              ", q(us(fr[:, 0, 10])))
print("frame role: hand-off published ", q(us(fr[:, 7, 12])))
w7 = fr[:, 7, :]
c0 = w7[:, 2]   # start of phase C (shader cycles)
print("frame role wave 7, shader cycles into phase C: walks done", q(w7[:, 13] - c0), "| wave 6 + 0 seen", q(w7[:, 14] - c0),
      "| operands stored", q(w7[:, 15] - c0), "| end of C", q(w7[:, 3] - c0))
print("                   own stores drained (us)", q(us(w7[:, 9])), "| signalled (us)", q(us(w7[:, 12])), "| waiting for wave 0's drain", q((w7[:, 12] - w7[:, 9]) / 100))
print("frame role: blend coefficients drained, about to be signalled (wave 5, top of phase C)", q(us(fr[:, 5, 14])), "| slowest frame of each unit:", np.round([us(fr[u * 32:(u + 1) * 32, 5, 14]).max() for u in range((F + 31) // 32)][:8], 2))
print("frame role: transforms published (wave 7), slowest frame of each unit:", np.round([us(fr[u * 32:(u + 1) * 32, 7, 12]).max() for u in range((F + 31) // 32)][:8], 2))
pub = us(fr[:, 7, 12])
late = np.argsort(pub)[-24:]
print("frame role: latest 24 hand-offs (frame: us):", " ".join(f"{int(i)}:{pub[i]:.1f}" for i in late))
print("frame role: latest 8 hand-offs: frame (XCD) entry us | wave 7 cycles entry->C, C->walks, ->seen, ->stored | hand-off us | us per 1000 cycles")
for i in late[-8:]:
    cyc_all = w7[i, 15] - w7[i, 0]
    print(f"    {int(i):4d} ({int(fr[i, 0, 9])})  {us(fr[i, 0, 10]):5.2f} | {int(w7[i, 2] - w7[i, 0]):6d} {int(w7[i, 13] - w7[i, 2]):6d} {int(w7[i, 14] - w7[i, 13]):6d} {int(w7[i, 15] - w7[i, 14]):6d}"
          f" | {pub[i]:5.2f} | {(pub[i] - us(fr[i, 0, 10])) / cyc_all * 1000:.3f}")
med = np.argsort(pub)[F // 2 - 4:F // 2 + 4]
print("frame role: 8 median hand-offs, same columns")
for i in med:
    cyc_all = w7[i, 15] - w7[i, 0]
    print(f"    {int(i):4d} ({int(fr[i, 0, 9])})  {us(fr[i, 0, 10]):5.2f} | {int(w7[i, 2] - w7[i, 0]):6d} {int(w7[i, 13] - w7[i, 2]):6d} {int(w7[i, 14] - w7[i, 13]):6d} {int(w7[i, 15] - w7[i, 14]):6d}"
          f" | {pub[i]:5.2f} | {(pub[i] - us(fr[i, 0, 10])) / cyc_all * 1000:.3f}")
print("frame role: hand-off percentiles 50/90/95/99/100:", np.percentile(pub, [50, 90, 95, 99, 100]).round(2))
xcc = fr[:, 0, 9].astype(int)
print("frame role: median hand-off by XCD:", " ".join(f"{x}:{np.median(pub[xcc == x]):.2f}/{pub[xcc == x].max():.2f}(n={int((xcc == x).sum())})" for x in sorted(set(xcc))))
print("frame role: XCD of block b (first 16 blocks):", xcc[:16])
print("frame role: end                ", q(us(fr[:, :, 11].max(1))))
print("mesh role: entry               ", q(us(ms[:, :, 0])))
print("mesh role: own unit seen complete (per wave)", q(us(ms[:, :, 1])))
print("mesh role: blend starts        ", q(us(ms[:, :, 2])))
print("mesh role: blend done          ", q(us(ms[:, :, 3])))
print("mesh role: skinning done       ", q(us(ms[:, :, 4])))
print("mesh role durations: wait", q((ms[:, :, 1] - ms[:, :, 0]) / 100), "| prologue", q((ms[:, :, 2] - ms[:, :, 1]) / 100))
print("                     blend", q((ms[:, :, 3] - ms[:, :, 2]) / 100), "| skin", q((ms[:, :, 4] - ms[:, :, 3]) / 100))
cyc = np.diff(fr[:, :, :9], axis=2)
names = ["A tables", "B rodrigues/offsets", "C chain walks / landmark items", "C barrier", "D", "E", "hand-off + F1", "F2 sweep"]
print("frame role phases, shader cycles (median over frames of the slowest wave | per wave 0..7):")
for i, n in enumerate(names):
    print(f"  {n:32s} {int(np.median(cyc[:, :, i].max(1))):6d} |", np.median(cyc[:, :, i], axis=0).astype(int))
is_late = pub > np.percentile(pub, 85)
print("late frames (top 15 % hand-off) vs the rest, slowest-wave cycles per phase:")
for i, n in enumerate(names):
    print(f"  {n:32s} late {int(np.median(cyc[is_late][:, :, i].max(1))):6d}   rest {int(np.median(cyc[~is_late][:, :, i].max(1))):6d}")
print("frame role total cycles (median):", int(np.median(fr[:, :, 8].max(1) - fr[:, :, 0].min(1))))

cy = raw[CYC:CYC + nG * nvt * 8 * 40].reshape(nG * nvt, 8, 40).astype(np.float64)
steps = np.diff(cy[:, :, 0:15], axis=2)
rows = np.diff(cy[:, :, 16:33], axis=2)
print("mesh role blend k-steps, shader cycles (median over tiles; waves 0-3 | waves 4-7):")
print("   ", np.median(steps[:, :4], axis=(0, 1)).astype(int), "|", np.median(steps[:, 4:], axis=(0, 1)).astype(int))
print("mesh role skinning rows, shader cycles (median over tiles and waves):")
print("   ", np.median(rows, axis=(0, 1)).astype(int))
print("blend total cycles (median)", int(np.median(cy[:, :, 14] - cy[:, :, 0])), " skin total", int(np.median(cy[:, :, 32] - cy[:, :, 16])))
# ---- the launch's tail: when do the workgroups end, and which end last -------------------------------------------------
fe = us(fr[:, :, 11]).max(axis=1)            # frame workgroup f: its last wave's end
me = us(ms[:, :, 4]).max(axis=1)             # mesh workgroup (tile): its last wave's end
pc = [50, 75, 90, 95, 99, 100]
print("tail: frame workgroups end, percentiles", pc, np.round(np.percentile(fe, pc), 2))
print("tail: mesh workgroups end,  percentiles", pc, np.round(np.percentile(me, pc), 2))
xcd_f = fr[:, 0, 9].astype(int)
print("tail: frame workgroups' end by XCD (median / max):", " ".join(f"{x}:{np.median(fe[xcd_f == x]):.1f}/{fe[xcd_f == x].max():.1f}" for x in range(8)))
nv = ms.shape[0]
xcd_m = (np.arange(nv) + F) % 8 if nG == 1 else None   # block order: frames, then mesh tiles (one group): XCD = block % 8
if xcd_m is not None:
    print("tail: mesh workgroups' end by XCD (median / max): ", " ".join(f"{x}:{np.median(me[xcd_m == x]):.1f}/{me[xcd_m == x].max():.1f}" for x in range(8)))
    late = np.argsort(me)[-12:]
    print("tail: latest 12 mesh workgroups (tile: end us | blend start of its waves min..max | slowest wave):",
          " ".join(f"{int(t)}:{me[t]:.1f}|{us(ms[t, :, 2]).min():.1f}..{us(ms[t, :, 2]).max():.1f}|w{int(np.argmax(ms[t, :, 4]))}" for t in late))
    w_end = us(ms[:, :, 4])
    print("tail: mesh waves' end by wave number (median):", np.round(np.median(w_end, axis=0), 2))
    print("tail: mesh waves' own-unit-seen by wave number (median):", np.round(np.median(us(ms[:, :, 1]), axis=0), 2))
