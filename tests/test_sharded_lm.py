"""Frame-sharded LM (tests/sharded_lm_check.py): world sizes 1, 2 and 3 over gloo on the CPU with the oracle as the
local evaluator must reproduce the dense single-process LM (oracle/lm_dense.py); on the GPU the same class with the HIP
evaluator must reproduce bodyfit_solve."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRI = dict(beta_pose=5.0, beta_shape=25.0, lambda_t=3.0)


class OracleNormals:
    """normals / cost of a shard's frames from the CPU restatement, in k_frame_normal's panel layout."""

    def __init__(self, om, oracle, seq_slice):
        class S: pass
        self.s = S(); self.s.__dict__.update(seq_slice)
        self.om, self.oracle = om, oracle

    def _eval(self, x, beta, want_jac):
        r, J = self.om.evaluate_batch(self.s, x, beta, 86, True, True, mode=0, want_jac=want_jac)
        K = len(r) // 2
        rk = r.reshape(K, 2)
        rho = np.array([self.oracle.huber(3.0, v) for v in (rk ** 2).sum(1)]).reshape(K, 3)
        return rk, J, rho

    def cost(self, x, beta):
        _, _, rho = self._eval(x, beta, False)
        return 0.5 * rho[:, 0].sum()

    def normals(self, x, beta):
        rk, J, rho = self._eval(x, beta, True)
        n = x.shape[0]
        P = np.zeros((n, 87, 88))
        for l in range(n):
            k0, k1 = self.s.kp_offset[l], self.s.kp_offset[l + 1]
            w = np.repeat(rho[k0:k1, 1], 2)
            Jh = np.concatenate([J[2 * k0:2 * k1], rk[k0:k1].reshape(-1, 1)], axis=1)      # [rows, 86 + 1]
            P[l, :, :87] = np.tril((Jh * w[:, None]).T @ Jh)
        return 0.5 * rho[:, 0].sum(), P


def _worker(rank, world, port, F, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    import sharded_lm_check as slm
    from oracle import oracle
    model = synth.make_model(0, n_verts=1200)
    seq = synth.make_sequence(model, F, seed=4)
    om = oracle.OracleModel(model)
    shard = sharded.make_shard(F, world, rank)
    local = OracleNormals(om, oracle, sharded.slice_sequence(seq, shard))
    lm = slm.ShardedLM(F, local, dist=dist if world > 1 else None, rank=rank, world=world, max_iters=12, **PRI)
    x, b, info = lm.solve(seq.init_params, np.zeros(10))
    if rank == 0:
        np.savez(out_path, x=x, b=b, cost=info["final_cost"], it=info["iterations"], ok=info["n_ok"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(1, 5), (2, 5), (3, 7), (3, 3), (8, 19)])   # (8: the node the north star names)
def test_sharded_lm_equals_dense_lm(tmp_path, world, F):
    out = str(tmp_path / "res.npz")
    port = 29600 + world * 7 + F
    if world == 1:
        _worker(0, 1, port, F, out)
    else:
        mp.spawn(_worker, args=(world, port, F, out), nprocs=world, join=True)
    got = np.load(out)
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    from oracle import lm_dense, oracle
    model = synth.make_model(0, n_verts=1200)
    seq = synth.make_sequence(model, F, seed=4)
    om = oracle.OracleModel(model)
    xd, bd, info = lm_dense.solve(om, seq, seq.init_params, np.zeros(10), beta_pose=PRI["beta_pose"], beta_shape=PRI["beta_shape"],
                                  lam=PRI["lambda_t"], max_iters=12, scale_bounds=(-1e300, 1e300))
    assert int(got["it"]) == info["iterations"] and int(got["ok"]) == info["n_ok"]
    assert abs(float(got["cost"]) - info["final_cost"]) < 1e-8 * info["final_cost"]
    assert np.abs(got["x"] - xd).max() < 1e-6 and np.abs(got["b"] - bd).max() < 1e-6


@pytest.mark.gpu
def test_sharded_lm_with_hip_evaluator_matches_bodyfit_solve(api, synth, model, gpu_model):
    import sharded_lm_check as slm
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    F = 10
    seq = synth.make_sequence(model, F, seed=6)
    local = slm.HipNormals(api, gpu_model, sharded.slice_sequence(seq, sharded.make_shard(F, 1, 0)))
    lm = slm.ShardedLM(F, local, max_iters=20, **PRI)
    x, b, info = lm.solve(seq.init_params, np.zeros(10))
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=PRI["beta_pose"],
                                     beta_shape=PRI["beta_shape"], lambda_temporal=PRI["lambda_t"])
    x2, b2, s2 = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=20, scale_bounds=(-1e300, 1e300))
    assert info["iterations"] == s2[0].iterations
    assert abs(info["final_cost"] - s2[0].final_cost) < 1e-8 * s2[0].final_cost
    assert np.abs(x[:, 1:] - x2[:, 1:]).max() < 1e-6 and np.abs(b - b2).max() < 1e-6


def _gpu_worker(rank, world, port, F, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = importlib.import_module("3dbodyanimation_amd.api")
    synth = importlib.import_module("3dbodyanimation_amd.synth")
    sharded = importlib.import_module("3dbodyanimation_amd.sharded")
    import sharded_lm_check as slm
    model = synth.make_model(0)
    seq = synth.make_sequence(model, F, seed=6)
    gm = api.Model(model, device=0)
    local = slm.HipNormals(api, gm, sharded.slice_sequence(seq, sharded.make_shard(F, world, rank)))
    lm = slm.ShardedLM(F, local, dist=dist, rank=rank, world=world, max_iters=15, **PRI)
    x, b, info = lm.solve(seq.init_params, np.zeros(10))
    if rank == 0:
        prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_pose=PRI["beta_pose"],
                                         beta_shape=PRI["beta_shape"], lambda_temporal=PRI["lambda_t"])
        x2, b2, s2 = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=15, scale_bounds=(-1e300, 1e300))
        np.savez(out_path, x=x, b=b, cost=info["final_cost"], it=info["iterations"], x2=x2, b2=b2, cost2=s2[0].final_cost,
                 it2=s2[0].iterations)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_lm_three_ranks_sharing_the_gpu(tmp_path):
    """Three gloo ranks, each with its shard's frames as its own HIP problem on the box's one GPU, against the unsharded
    bodyfit_solve."""
    out = str(tmp_path / "gpu_lm.npz")
    port = 29500 + (os.getpid() % 2000) + 37
    mp.spawn(_gpu_worker, args=(3, port, 11, out), nprocs=3, join=True)
    g = np.load(out)
    assert int(g["it"]) == int(g["it2"])
    assert abs(float(g["cost"]) - float(g["cost2"])) < 1e-8 * float(g["cost2"])
    assert np.abs(g["x"][:, 1:] - g["x2"][:, 1:]).max() < 1e-6 and np.abs(g["b"] - g["b2"]).max() < 1e-6
