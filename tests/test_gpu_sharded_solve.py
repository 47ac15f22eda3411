"""bodyfit_solve_sharded (C ABI): one window fitted by several ranks, one bodyfit_problem per shard, collectives through a
bodyfit_comm (torch.distributed, gloo).  The ranks share the box's one GPU here; on a node every rank has its own.  The
result must be the unsharded device window LM's (same iterates up to rounding: the elimination order differs)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, F, iters, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    seq = synth.make_sequence(model, F, seed=6)
    gm = api.Model(model, device=0)
    x, b, summ, shard = sharded.solve_window_sharded(api, gm, seq, F, dist, rank, world, seq.init_params, np.zeros(10),
                                                     max_iters=iters)
    # three all-gathers per launched iteration (interface blocks + beta terms, Schur partials, scalars) and four at the start
    # (boundary rows, initial cost, the first iteration's beta terms and scaling rows); the host looks at the status every
    # fourth iteration, so up to three iterations are launched beyond the last counted one
    n_loop, rem = divmod(summ.exchanges - 4, 3)
    assert rem == 0 and summ.iterations <= n_loop <= min(iters, summ.iterations + 3), (summ.exchanges, summ.iterations)
    parts = [None] * world
    dist.all_gather_object(parts, (shard.f0, shard.f1, x, b, summ.iterations, summ.n_successful, summ.final_cost,
                                   summ.initial_cost))
    if rank == 0:
        xs = np.concatenate([p[2] for p in sorted(parts)], axis=0)
        # every rank holds the same beta and took the same decisions
        for p in parts:
            assert np.array_equal(p[3], parts[0][3]) and p[4:] == parts[0][4:]
        prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
        x2, b2, s2 = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=iters,
                                scale_bounds=(-1e300, 1e300), solver=3)
        np.savez(out_path, x=xs, b=parts[0][3], it=parts[0][4], ok=parts[0][5], cost=parts[0][6], c0=parts[0][7],
                 x2=x2, b2=b2, it2=s2[0].iterations, ok2=s2[0].n_successful, cost2=s2[0].final_cost, c02=s2[0].initial_cost)
    dist.barrier()
    dist.destroy_process_group()


# (2, 1024, 6): BASELINE configs[4] at its real size, two shards of 512 frames
@pytest.mark.parametrize("world,F,iters", [(2, 9, 12), (3, 14, 12), (2, 40, 8), (2, 1024, 6), (5, 43, 6)])   # (5 ranks + this process: the box admits six on its GPU)
def test_sharded_solve_equals_unsharded(tmp_path, world, F, iters):
    out = str(tmp_path / "res.npz")
    port = 29700 + (os.getpid() % 1500) + world * 11 + F
    mp.spawn(_worker, args=(world, port, F, iters, out), nprocs=world, join=True)
    g = np.load(out)
    assert (int(g["it"]), int(g["ok"])) == (int(g["it2"]), int(g["ok2"]))
    assert abs(float(g["c0"]) - float(g["c02"])) <= 1e-12 * float(g["c02"])
    assert abs(float(g["cost"]) - float(g["cost2"])) <= 1e-9 * float(g["cost2"])
    assert np.abs(g["x"][:, 1:] - g["x2"][:, 1:]).max() < 1e-7 and np.abs(g["b"] - g["b2"]).max() < 1e-7


def test_rccl_transport_single_rank(monkeypatch):
    """bodyfit_solve_sharded_rccl with the exchanges as real RCCL all-gathers on the device buffers and the solve's stream.
    A test box has ONE GPU, so the communicator has one rank; BODYFIT_FORCE_SHARDED=1 makes that rank still take the sharded
    code path (its two end frames as the interface system, every exchange issued through ncclAllGather).  The result must be
    the unsharded device window LM's."""
    import torch  # noqa: F401  (one HIP runtime / one librccl in the process)
    sys.path.insert(0, ROOT)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    F, iters = 24, 10
    seq = synth.make_sequence(model, F, seed=6)
    gm = api.Model(model, device=0)
    comm = api.Rccl.create(api.Rccl.unique_id(), 0, 1, 0)
    monkeypatch.setenv("BODYFIT_FORCE_SHARDED", "1")
    x, b, summ, shard = sharded.solve_window_sharded(api, gm, seq, F, None, 0, 1, seq.init_params, np.zeros(10), max_iters=iters,
                                                     rccl=comm)
    monkeypatch.delenv("BODYFIT_FORCE_SHARDED")
    comm.close()
    n_loop, rem = divmod(summ.exchanges - 4, 3)
    assert rem == 0 and summ.iterations <= n_loop <= iters
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
    x2, b2, s2 = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=iters, scale_bounds=(-1e300, 1e300), solver=3)
    assert (summ.iterations, summ.n_successful) == (s2[0].iterations, s2[0].n_successful)
    assert abs(summ.final_cost - s2[0].final_cost) <= 1e-9 * s2[0].final_cost
    assert np.abs(x[:, 1:] - x2[:, 1:]).max() < 1e-7 and np.abs(b - b2).max() < 1e-7


def test_shard_proxy_runs_one_ranks_work_at_the_eight_gpu_geometry():
    """bodyfit_set_shard_proxy (the measurement aid behind bench.py's c5_strong.shard_proxy): through a ONE-rank RCCL communicator
    the problem runs as rank 3 of 8 identical shards — the sharded code path with a 16-frame interface chain, three all-gathers
    per LM iteration, sums over 8 gathered slots.  There is no other implementation of "8 copies of a shard, every copy in the
    middle" to compare the numbers with; what is checked is that the LM works on that system (cost falls by orders of
    magnitude, steps accepted, no failed factorisation), that every exchange of the geometry is issued, that the run is
    deterministic, and that switching the aid off restores the plain one-rank solve."""
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    Fw, N, R, iters = 256, 8, 3, 24
    seq = synth.make_sequence(model, Fw, seed=6)
    gm = api.Model(model, device=0)
    sh = sharded.make_shard(Fw, N, R)
    assert sh.n_local == 32 and sh.halo
    sl = sharded.slice_sequence(seq, sh)
    prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86, use_shape=True,
                       beta_pose=5.0, beta_shape=0.0, lambda_temporal=3.0, temporal_halo=True)
    comm = api.Rccl.create(api.Rccl.unique_id(), 0, 1, 0)
    x0 = sharded.local_params(seq.init_params, sh)
    prob.set_shard_proxy(N, R)
    x, b, s1 = prob.solve_sharded_rccl(x0, np.zeros(10), comm, max_iters=iters)
    n_ex = prob.last_exchange_count()
    n_loop, rem = divmod(n_ex - 4, 3)
    assert rem == 0 and s1.iterations <= n_loop <= iters
    assert s1.termination in (0, 1) and s1.usable and s1.n_successful >= 10
    assert np.isfinite(x).all() and np.isfinite(b).all() and s1.final_cost < 0.02 * s1.initial_cost
    x2, b2, s2 = prob.solve_sharded_rccl(x0, np.zeros(10), comm, max_iters=iters)
    assert np.array_equal(x, x2) and np.array_equal(b, b2) and s1.final_cost == s2.final_cost
    # a shard with a halo row is not a window of its own: without the aid the one-rank solve refuses it
    prob.set_shard_proxy(0, 0)
    with pytest.raises(api.BodyfitError):
        prob.solve_sharded_rccl(x0, np.zeros(10), comm, max_iters=3)
    comm.close()


def test_rccl_allreduce_of_the_shared_reduction_single_rank():
    """The evaluation path's one collective issued BY THE LIBRARY (bodyfit_allreduce_shared_rccl: ncclAllReduce(sum, f64) of the
    66 doubles, in place, on the sweep's stream — SURVEY 8e; the shared shape block of include/MultiFrameBA.h:64-68 summed over
    the shards).  One GPU, so a communicator of one rank: the call must go through RCCL (ncclCommCount reports the rank count),
    run stream-ordered behind the sweep + the reduction at the sweep's own tail with no host synchronisation in between, and
    leave the sum of one shard = that shard's numbers."""
    import torch
    sys.path.insert(0, ROOT)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    model = synth.make_model(0)
    F = 48
    seq = synth.make_sequence(model, F, seed=4)
    gm = api.Model(model, device=0)
    prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0,
                                     want_mesh=True)
    comm = api.Rccl.create(api.Rccl.unique_id(), 0, 1, 0)
    assert comm.count() == (1, 0)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(seq.gt_params + 0.02).to(dev)
    b = torch.from_numpy(seq.gt_beta + 0.1).to(dev)
    ref = torch.zeros(66, dtype=torch.float64, device=dev)
    out = torch.zeros(66, dtype=torch.float64, device=dev)
    work = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(work):
        st = work.cuda_stream
        prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, st)
        prob.reduce_shared_device(ref.data_ptr(), st)              # a reduce launch: the reference numbers
        prob.arm_shared_reduction(out.data_ptr())
        for _ in range(3):                                         # sweep -> (tail reduction) -> all-reduce, back to back
            prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, st)
            prob.reduce_shared_device(out.data_ptr(), st)
            comm.allreduce_shared(out.data_ptr(), st)
        prob.sweep_status(st)
    assert torch.equal(out, ref) and float(ref[0]) > 0
    comm.close()


def _poison_worker(rank, world, port, F, iters, fail_rank, fail_iter, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    seq = synth.make_sequence(model, F, seed=6)
    gm = api.Model(model, device=0)
    shard = sharded.make_shard(F, world, rank)
    sl = sharded.slice_sequence(seq, shard)
    prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86, use_shape=True,
                       beta_pose=5.0, beta_shape=25.0 if shard.owns_shape_prior else 0.0, lambda_temporal=3.0,
                       temporal_halo=shard.halo)
    comm = sharded.TorchComm(api, dist, rank, world, device=None)
    import ctypes as C
    lib = api.load_library()
    lib.bodyfit_internal_set_test_poison.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert lib.bodyfit_internal_set_test_poison(prob.h, fail_rank, fail_iter) == 0
    msg = ""
    try:
        prob.solve_sharded(sharded.local_params(seq.init_params, shard), np.zeros(10), comm.c, max_iters=iters)
    except api.BodyfitError as e:
        msg = str(e)
    n_bad = prob.last_exchange_count()
    # ... and the failure does not stick to the problem (the device's status record lives in the problem's pool): the same
    # solve without the hook, on the same problem, runs to its end on both ranks
    assert lib.bodyfit_internal_set_test_poison(prob.h, -1, -1) == 0
    msg2, its2, cost2 = "", -1, float("nan")
    try:
        _, _, s2 = prob.solve_sharded(sharded.local_params(seq.init_params, shard), np.zeros(10), comm.c, max_iters=iters)
        its2, cost2 = s2.iterations, s2.final_cost
    except api.BodyfitError as e:
        msg2 = str(e)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write(f"{n_bad}\n{msg}\n{msg2}\n{its2}\n{cost2!r}\n")
    dist.barrier()               # both ranks are still in step: a rank stuck in an exchange would hang here (60 s bound)
    dist.destroy_process_group()


def test_a_failing_rank_takes_every_rank_out_at_the_same_exchange(tmp_path):
    """Cross-rank failure propagation of bodyfit_solve_sharded (include/bodyfit.h): rank 1's sweep 'fails' in LM iteration 5
    (test hook).  It must not simply return — rank 0 would wait for it in the next all-gather for ever; it marks its scalars,
    both ranks' decision kernels end the solve in that same iteration, and BOTH calls return an error after the SAME number of
    exchanges (no hang: the process group's 60 s timeout is never reached)."""
    world, F, iters = 2, 40, 30
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_poison_worker, args=(world, port, F, iters, 1, 5, str(tmp_path)), nprocs=world, join=True)
    res = [open(tmp_path / f"rank{r}.txt").read().splitlines() for r in range(world)]
    n0, n1 = int(res[0][0]), int(res[1][0])
    assert n0 == n1 and n0 >= 4 + 3 * 6                      # iterations 0 .. 5 were exchanged completely
    assert n0 <= 4 + 3 * (5 + 4)                             # ... and at most three more before the status read
    assert "another rank reported a device failure" in res[0][1]
    assert "this rank failed" in res[1][1] and "test hook" in res[1][1]
    # the second, un-poisoned solve on the SAME problems succeeded on both ranks, with identical decisions
    for r in range(world):
        assert res[r][2] == "", res[r][2]
    assert int(res[0][3]) == int(res[1][3]) > 6
    assert float(res[0][4]) == float(res[1][4]) and np.isfinite(float(res[0][4]))


def _transport_failure_worker(rank, world, port, F, fail_call, bound, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    import time
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    seq = synth.make_sequence(model, F, seed=6)
    gm = api.Model(model, device=0)
    shard = sharded.make_shard(F, world, rank)
    sl = sharded.slice_sequence(seq, shard)
    prob = api.Problem(gm, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86, use_shape=True,
                       beta_pose=5.0, beta_shape=25.0 if shard.owns_shape_prior else 0.0, lambda_temporal=3.0,
                       temporal_halo=shard.halo)
    tc = sharded.TorchComm(api, dist, rank, world, device=None)
    inner = tc._cbs[1]
    calls = [0]

    def allgather(ctx, send, recv, n):
        calls[0] += 1
        if rank == 1 and calls[0] == fail_call:
            return 1                      # rank 1's transport fails: it never enters this collective
        return inner(ctx, send, recv, n)

    cb = api._ALLGATHER_CB(allgather)
    comm = api.Comm(rank, world, None, tc._cbs[0], cb)
    prob.set_exchange_timeout(bound)
    dist.barrier()
    t0 = time.perf_counter()
    msg = ""
    try:
        prob.solve_sharded(sharded.local_params(seq.init_params, shard), np.zeros(10), comm, max_iters=40)
    except api.BodyfitError as e:
        msg = str(e)
    dt = time.perf_counter() - t0
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write(f"{dt}\n{calls[0]}\n{msg}\n")
        fh.flush()
    if rank == 1:
        # stay alive (and out of the collective) until rank 0 is back: a rank that DIES closes its sockets, which gloo notices
        # at once — the case the bound is for is the peer that is alive and simply never arrives
        t_end = time.perf_counter() + 90.0
        while not os.path.exists(os.path.join(out_dir, "rank0.txt")) and time.perf_counter() < t_end:
            time.sleep(0.05)
    # rank 0's helper thread still sits in gloo's all_gather (until the process group's own 120 s): leave without tearing
    # the group down in an orderly way — what an application would do after such an error
    os._exit(0)


def test_a_transport_failure_strands_nobody_with_an_exchange_timeout(tmp_path):
    """include/bodyfit.h, failures of the transport itself: rank 1's all-gather callback returns non-zero in its 12th exchange
    (iteration 2) and rank 1 leaves at once.  Rank 0 is then alone in a gloo collective whose own timeout is 120 s; with
    bodyfit_set_exchange_timeout(4 s) its solve returns an error after ~4 s — BOTH ranks are back within the bound."""
    world, F, bound = 2, 40, 4.0
    port = 29900 + ((os.getpid() + 311) % 1000)
    mp.spawn(_transport_failure_worker, args=(world, port, F, 12, bound, str(tmp_path)), nprocs=world, join=True)
    res = [open(tmp_path / f"rank{r}.txt").read().splitlines() for r in range(world)]
    t0, t1 = float(res[0][0]), float(res[1][0])
    assert "allgather callback failed" in res[1][2] and int(res[1][1]) == 12
    assert t1 < bound                                         # the rank that saw the failure: at once
    assert "exchange timeout" in res[0][2], res[0][2]
    assert bound <= t0 < bound + 20.0                         # its peer: after the bound, not after gloo's 120 s
    assert int(res[0][1]) == 12                               # ... and in that same exchange
