"""What torch.cuda.synchronize() itself costs on an idle device, and what a K-step bracket reads, by the stream the sweeps run on:
torch's default stream, a torch.cuda.Stream() (which makes torch create its pool of streams), one hipStream created through the
HIP runtime and wrapped (torch.cuda.ExternalStream).  One child process per variant."""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    mode = sys.argv[1]
    sys.path.insert(0, ROOT)
    import torch
    api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
    m = synth.make_model(0); gm = api.Model(m); seq = synth.make_sequence(m, 256, seed=0); gmm = api.Gmm(*synth.make_gmm(0))
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, pose_blend=True, beta_pose=20.0, gmm=gmm,
                                     beta_shape=30.0, want_mesh=True)
    dx = torch.from_numpy(seq.gt_params + 0.01).cuda(); db = torch.from_numpy(np.tile(seq.gt_beta, (256, 1))).cuda()
    if mode == "default":
        st = torch.cuda.current_stream().cuda_stream
    elif mode == "pool":
        ws = torch.cuda.Stream(); torch.cuda.set_stream(ws); st = ws.cuda_stream
    else:
        import ctypes
        hip = ctypes.CDLL(None)
        h = ctypes.c_void_p()
        fn = getattr(hip, "hipStreamCreateWithFlags", None)
        if fn is None:
            hip = ctypes.CDLL("libamdhip64.so"); fn = hip.hipStreamCreateWithFlags
        assert fn(ctypes.byref(h), 1) == 0           # hipStreamNonBlocking
        ws = torch.cuda.ExternalStream(h.value); torch.cuda.set_stream(ws); st = ws.cuda_stream
    for _ in range(1500): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
    torch.cuda.synchronize()
    idle = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); idle.append((time.perf_counter() - t0) * 1e6)
    res = {}
    for K in (1, 20, 200):
        v = []
        for rep in range(21):
            for _ in range(5): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
            torch.cuda.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K): prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, st)
            torch.cuda.synchronize()
            v.append((time.perf_counter() - t0) * 1e6)
        res[K] = np.median(v)
    print(f"{mode:8s}: idle torch.cuda.synchronize() {np.median(idle):6.1f} us | K=1 {res[1]:7.1f} us | K=20 {res[20]:7.1f} us = {res[20] / 20:6.2f} per step | "
          f"K=200 {res[200] / 200:6.2f} per step | fixed cost (K=20 - 20 x K=200 rate) {res[20] - 20 * res[200] / 200:6.1f} us")
    sys.exit(0)
for rnd in range(2):
    for mode in ("default", "pool", "external"):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-300:], flush=True)
