"""A/B of diagnostic builds of the sweep (tools/build_variant.sh): every named libbodyfit_NAME.so runs the C3 problem in a
child process of its own (BODYFIT_LIB is read at import), interleaved over `rounds` passes so that box drift falls on all of
them alike; each child checks its cloud / residuals / Jacobian against the shipped library's (bit for bit: the variants only
move instructions) and prints wall time per step of back-to-back sweeps and the dispatch's own duration.
usage: python tools/variant_ab.py [-F 256] [-r 3] name1 name2 ...   (name "ship" = libbodyfit.so)"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(F, iters):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    m = synth.make_model(0)
    gm = api.Model(m)
    seq = synth.make_sequence(m, F, seed=0)
    gmm = api.Gmm(*synth.make_gmm(0))
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True,
                                     beta_pose=20.0, gmm=gmm, beta_shape=30.0, want_mesh=True)
    dev = torch.device("cuda", 0)
    dx = torch.from_numpy(seq.gt_params + 0.01).to(dev)
    db = torch.from_numpy(np.tile(seq.gt_beta, (F, 1))).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(30):
        prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
    torch.cuda.synchronize()
    walls = []
    for _ in range(6):
        t0 = time.perf_counter()
        for _ in range(iters):
            prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / iters * 1e6)
    prof = prob.profile_sweep(dx.data_ptr(), db.data_ptr(), True, False, iters, st)
    status = prob.sweep_status(st)
    r, J, _ = prob.evaluate(seq.gt_params + 0.01, np.tile(seq.gt_beta, (F, 1)), True)
    _, c = prob.forward(seq.gt_params + 0.01, np.tile(seq.gt_beta, (F, 1)))
    h = hashlib.sha256(c.tobytes() + r.tobytes() + J.tobytes()).hexdigest()[:16]
    print(json.dumps(dict(us_per_step=[round(w, 2) for w in walls], kernel_us=round(prof["sweep_roles"] * 1e3, 2),
                          status=status, hash=h, finite=bool(np.isfinite(c).all()))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("-F", type=int, default=256)
    ap.add_argument("-r", "--rounds", type=int, default=2)
    ap.add_argument("-i", "--iters", type=int, default=300)
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    res = {n: [] for n in a.names}
    for rnd in range(a.rounds):
        for n in a.names:
            base, _, kv = n.partition("@")     # NAME@TUNE0=1,TUNE1=2 -> BODYFIT_TUNE0=1 BODYFIT_TUNE1=2 in the child
            lib = os.path.join(ROOT, "3dbodyanimation_amd", "libbodyfit.so" if base == "ship" else f"libbodyfit_{base}.so")
            env = dict(os.environ, BODYFIT_LIB=lib)
            for item in filter(None, kv.split(",")):
                k, _, v = item.partition("=")
                env[k[4:] if k.startswith("ENV:") else "BODYFIT_" + k] = v     # (ENV:NAME=value: any variable, as it stands)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(a.F), str(a.iters)], env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", out.stderr[-2000:], flush=True)
                continue
            d = json.loads(line[-1])
            res[n].append(d)
            print(f"F={a.F} {n:24s} round {rnd}: {d}", flush=True)
    ref = None
    for n in a.names:
        if not res[n]:
            continue
        best = min(min(d["us_per_step"]) for d in res[n])
        med = sorted(w for d in res[n] for w in d["us_per_step"][2:])   # (the first windows of a child run on a cold clock)
        kern = min(d["kernel_us"] for d in res[n])
        hs = {d["hash"] for d in res[n]}
        ref = ref or hs
        print(f"SUMMARY F={a.F} {n:24s} best {best:6.2f}  median {med[len(med) // 2]:6.2f} us/step  kernel(min of averages) {kern:6.2f} us"
              f"  outputs {'== first' if hs == ref else 'DIFFER from first'}", flush=True)
