"""Frame sharding + the shared-shape all-reduce.  CPU: world_size-2 gloo processes with the
oracle as the local evaluator; the all-reduced [cost, g_beta, H_bb] must equal the single-process window."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shared_from_oracle(om, oracle, seq_slice, x_local, beta, n_local, halo, beta_pose, beta_shape, lam, huber=3.0):
    """66-vector of one shard from the CPU restatement (what the HIP reduce kernel produces on a GPU)."""
    class S: pass
    s = S(); s.__dict__.update(seq_slice)
    r, J = om.evaluate_batch(s, x_local[:n_local], beta, 86, True, True, mode=0)
    K = len(r) // 2
    rk = r.reshape(K, 2); Jb = J[:, 76:].reshape(K, 2, 10)
    rho = np.array([oracle.huber(huber, v) for v in (rk ** 2).sum(1)])
    cost = 0.5 * rho[:, 0].sum()
    g = np.einsum("k,kri,kr->i", rho[:, 1], Jb, rk)
    H = np.einsum("k,kri,krj->ij", rho[:, 1], Jb, Jb)
    for f in range(n_local):
        cost += 0.5 * ((beta_pose * x_local[f, 7:]) ** 2).sum()
    if beta_shape > 0:
        cost += 0.5 * ((beta_shape * beta) ** 2).sum()
        g = g + beta_shape ** 2 * beta
        H = H + beta_shape ** 2 * np.eye(10)
    src = np.concatenate([np.arange(4, 7), np.arange(1, 4), np.arange(7, 76)])
    for f in range(n_local - 1 + (1 if halo else 0)):
        cost += 0.5 * ((lam * (x_local[f, src] - x_local[f + 1, src])) ** 2).sum()
    return np.concatenate([[cost], g, H[np.triu_indices(10)]])


def _worker(rank, world, port, F, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    from oracle import oracle
    model = synth.make_model(0, n_verts=1200)
    seq = synth.make_sequence(model, F, seed=4)
    om = oracle.OracleModel(model)

    class OracleLocal:
        def __init__(self, shard, sl, kw):
            self.shard, self.sl, self.kw = shard, sl, kw

        def evaluate_shared(self, x_local, beta):
            v = _shared_from_oracle(om, oracle, self.sl, x_local, beta, self.shard.n_local, self.shard.halo,
                                    self.kw["beta_pose"], self.kw["beta_shape"], self.kw["lambda_temporal"])
            return torch.from_numpy(v)

    win = sharded.ShardedWindow(seq, F, lambda sh, sl, kw: OracleLocal(sh, sl, kw))
    x = seq.gt_params + 0.02
    beta = seq.gt_beta + 0.1
    cost, g, H = win.evaluate_shared(x, beta)
    if rank == 0:
        np.savez(out_path, cost=cost, g=g, H=H)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 9), (2, 2), (3, 2)])
def test_sharded_window_equals_single_process(tmp_path, world, F):
    out = str(tmp_path / "shared.npz")
    port = 29500 + (os.getpid() % 2000) + world * 7 + F
    mp.spawn(_worker, args=(world, port, F, out), nprocs=world, join=True)
    got = np.load(out)
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    from oracle import oracle
    model = synth.make_model(0, n_verts=1200)
    seq = synth.make_sequence(model, F, seed=4)
    om = oracle.OracleModel(model)
    whole = sharded.make_shard(F, 1, 0)
    ref = _shared_from_oracle(om, oracle, sharded.slice_sequence(seq, whole), seq.gt_params + 0.02, seq.gt_beta + 0.1,
                              F, False, 5.0, 25.0, 3.0)
    assert abs(got["cost"] - ref[0]) < 1e-9 * abs(ref[0])
    assert np.abs(got["g"] - ref[1:11]).max() < 1e-9 * np.abs(ref[1:11]).max()
    assert np.abs(got["H"][np.triu_indices(10)] - ref[11:]).max() < 1e-9 * np.abs(ref[11:]).max()


def test_shard_ranges_cover_and_balance():
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    for F in [1, 2, 7, 103, 1024]:
        for world in [1, 2, 3, 4, 8]:
            rs = [sharded.shard_range(F, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == F
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
            shards = [sharded.make_shard(F, world, r) for r in range(world)]
            assert sum(s.owns_shape_prior for s in shards) == 1
            # every temporal pair (f, f+1) is owned exactly once: inside a shard, or through its halo
            owned = sum(max(0, s.n_local - 1) + (1 if s.halo else 0) for s in shards)
            assert owned == F - 1


@pytest.mark.gpu
def test_sharded_window_on_gpu_matches_unsharded(tmp_path):
    """Three gloo ranks sharing the one GPU of the box: HIP sweep + device reduce per shard, all-reduce, vs one
    unsharded problem."""
    out = str(tmp_path / "gpu_shared.npz")
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_gpu_worker, args=(3, port, 10, out), nprocs=3, join=True)
    got = np.load(out)
    assert abs(got["cost"] - got["cost_ref"]) < 1e-9 * abs(got["cost_ref"])
    assert np.abs(got["g"] - got["g_ref"]).max() < 1e-9 * np.abs(got["g_ref"]).max()
    assert np.abs(got["H"] - got["H_ref"]).max() < 1e-9 * np.abs(got["H_ref"]).max()


def _gpu_worker(rank, world, port, F, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    model = synth.make_model(0)
    seq = synth.make_sequence(model, F, seed=4)
    gm = api.Model(model, device=0)

    class CpuBuf(sharded.HipLocal):  # gloo reduces host tensors
        def evaluate_shared(self, x_local, beta):
            return super().evaluate_shared(x_local, beta).cpu()

    win = sharded.ShardedWindow(seq, F, lambda sh, sl, kw: CpuBuf(api, gm, sh, sl, kw, torch.device("cuda", 0)))
    x = seq.gt_params + 0.02
    beta = seq.gt_beta + 0.1
    cost, g, H = win.evaluate_shared(x, beta)
    if rank == 0:
        one = sharded.ShardedWindow(seq, F, lambda sh, sl, kw: CpuBuf(api, gm, sh, sl, kw, torch.device("cuda", 0)),
                                    rank=0, world=1)
        c1, g1, H1 = one.evaluate_shared(x, beta)
        np.savez(out_path, cost=cost, g=g, H=H, cost_ref=c1, g_ref=g1, H_ref=H1)
    dist.barrier()
    dist.destroy_process_group()
