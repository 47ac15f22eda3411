"""Where the wall time of drivers.run_multi (the reference's staged 3dba_multi run, C4: 128 frames) goes: problem creation,
the solves, write-back problems, the rest (host bookkeeping)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
api = importlib.import_module("3dbodyanimation_amd.api")
synth = importlib.import_module("3dbodyanimation_amd.synth")
drivers = importlib.import_module("3dbodyanimation_amd.drivers")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = synth.make_model(0); gm = api.Model(model)
seq = synth.make_sequence(model, F, seed=2)
ks = drivers.KeypointSequence(seq.kp_offset, seq.kp_id, seq.kp_uv, [f"{i:06d}.json" for i in range(F)])
acc = {}
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            d_ = time.perf_counter() - t0
            acc.setdefault(name, [0.0, 0, []]); acc[name][0] += d_; acc[name][1] += 1; acc[name][2].append(round(d_ * 1e3, 2))
    setattr(cls, name, f)
for n in ("__init__", "solve", "writeback", "close", "__del__"):
    wrap(api.Problem, n)
drivers.run_multi(gm, ks, seq.intr)
for rep in range(3):
    acc.clear()
    t0 = time.perf_counter()
    res = drivers.run_multi(gm, ks, seq.intr)
    dt = time.perf_counter() - t0
    it = res["stage1"].iterations + sum(q.iterations for q in res["stage2"])
    print(f"run_multi {F} frames: {dt*1e3:.1f} ms, {it} LM iterations; " +
          "  ".join(f"{k}: {v[0]*1e3:.1f} ms / {v[1]}" for k, v in acc.items()) +
          f"  other: {(dt - sum(v[0] for k, v in acc.items() if k != '__del__'))*1e3:.1f} ms", flush=True)
    print("   __init__ per call (ms):", acc["__init__"][2], flush=True)
