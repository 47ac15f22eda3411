"""Frame sharding of one multi-frame window across the GPUs of a node.

OptimizeMultiFrame (include/MultiFrameBA.h:33-177) couples frames only through (a) the shared shape
block beta (:67-68,95,100), (b) temporal links between adjacent frames (:121-142) and (c) one shape prior
(:115-118).  So frames shard into contiguous ranges, one per rank:
  * rank r owns frames [f0, f1) and the temporal pairs (f, f+1) for f in [f0, f1); the pair that crosses
    into the next shard needs that shard's first frame — a 608-byte halo taken from the (replicated)
    parameter vector, no GPU-to-GPU exchange;
  * the shape prior is evaluated on rank 0 only;
  * the only collective of an evaluation is ONE all-reduce (sum, f64) of the fused 66-double buffer
    [cost, g_beta(10), upper(H_beta_beta)(55)] — RCCL over xGMI with backend "nccl", gloo on CPU.
The local evaluator is injected, so the same class drives the HIP path (api.Problem) on GPUs and is
exercised by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

N_SHARED = 66  # 1 + 10 + 55


def shard_range(n_frames: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced: the first (n_frames % world) ranks get one extra frame."""
    base, extra = divmod(n_frames, world)
    f0 = rank * base + min(rank, extra)
    return f0, f0 + base + (1 if rank < extra else 0)


@dataclass
class Shard:
    rank: int
    world: int
    f0: int
    f1: int
    halo: bool          # a temporal link leaves this shard (f1 < n_frames)
    owns_shape_prior: bool

    @property
    def n_local(self):
        return self.f1 - self.f0


def make_shard(n_frames: int, world: int, rank: int) -> Shard:
    f0, f1 = shard_range(n_frames, world, rank)
    return Shard(rank, world, f0, f1, f1 < n_frames and f1 > f0, rank == 0)


def slice_sequence(seq, shard: Shard):
    """Keypoint CSR / R0 of the shard's frames (offsets rebased to 0)."""
    k0, k1 = int(seq.kp_offset[shard.f0]), int(seq.kp_offset[shard.f1])
    return dict(kp_offset=(seq.kp_offset[shard.f0:shard.f1 + 1] - k0).astype(np.int32),
                kp_id=seq.kp_id[k0:k1], kp_uv=seq.kp_uv[k0:k1], intr=seq.intr, R0=seq.R0[shard.f0:shard.f1])


def local_params(params_full: np.ndarray, shard: Shard) -> np.ndarray:
    """The shard's parameter rows plus the halo row (the next shard's first frame) when it has one."""
    return np.ascontiguousarray(params_full[shard.f0:shard.f1 + (1 if shard.halo else 0)])


def upper_to_full(h55: np.ndarray) -> np.ndarray:
    H = np.zeros((10, 10))
    H[np.triu_indices(10)] = h55
    return H + np.triu(H, 1).T


class ShardedWindow:
    """One multi-frame window evaluated by `world` ranks.

    make_local(shard, seq_slice, prior_kwargs) -> object with
        evaluate_shared(local_params, beta) -> 66-vector (torch tensor on the collective's device)
    """

    def __init__(self, seq, n_frames, make_local, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0,
                 rank=None, world=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.shard = make_shard(n_frames, self.world, self.rank)
        kw = dict(beta_pose=beta_pose, beta_shape=beta_shape if self.shard.owns_shape_prior else 0.0,
                  lambda_temporal=lambda_temporal, temporal_halo=self.shard.halo)
        self.local = make_local(self.shard, slice_sequence(seq, self.shard), kw) if self.shard.n_local > 0 else None

    def evaluate_shared(self, params_full: np.ndarray, beta: np.ndarray):
        """Local sweep + the single all-reduce.  Returns (cost, g_beta[10], H_bb[10,10]) on every rank."""
        import torch
        if self.local is not None:
            buf = self.local.evaluate_shared(local_params(params_full, self.shard), beta)
        else:
            buf = torch.zeros(N_SHARED, dtype=torch.float64)
        if self.world > 1:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        out = buf.detach().cpu().numpy()
        return float(out[0]), out[1:11].copy(), upper_to_full(out[11:])


class HipLocal:
    """Local evaluator on one MI355X: api.Problem sweep + device reduction into a torch tensor."""

    def __init__(self, api, gpu_model, shard, seq_slice, prior_kw, device):
        import torch
        self.torch = torch
        self.api = api
        self.device = device
        self.prob = api.Problem(gpu_model, seq_slice["kp_offset"], seq_slice["kp_id"], seq_slice["kp_uv"],
                                seq_slice["intr"], seq_slice["R0"], n_cols=86, use_shape=True, **prior_kw)
        self.buf = torch.zeros(N_SHARED, dtype=torch.float64, device=device)
        self.prob.arm_shared_reduction(self.buf.data_ptr())   # (one-launch sweeps reduce at their own tail)

    def evaluate_shared(self, local_params, beta):
        torch = self.torch
        x = torch.from_numpy(np.ascontiguousarray(local_params)).to(self.device)
        b = torch.from_numpy(np.ascontiguousarray(beta)).to(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.prob.evaluate_device(x.data_ptr(), b.data_ptr(), True, stream)
        self.prob.reduce_shared_device(self.buf.data_ptr(), stream)
        # waits for the stream and reports a one-launch sweep whose in-launch wait ran out (bodyfit_sweep_status; the 66 doubles
        # do not depend on such a wait, the cloud does: a caller that also consumes the cloud must not go on)
        self.prob.sweep_status(stream)
        return self.buf


class TorchComm:
    """bodyfit_comm over torch.distributed for bodyfit_solve_sharded (include/bodyfit.h): the two collectives the sharded LM
    needs, on small host buffers.  backend "gloo": CPU tensors as they are; backend "nccl" (= RCCL over xGMI): through a
    device staging tensor, since RCCL reduces device memory."""

    def __init__(self, api, dist, rank: int, world: int, device=None):
        import ctypes as C
        import torch
        self.dist, self.rank, self.world, self.device = dist, rank, world, device

        def _tensor(ptr, n):
            a = np.ctypeslib.as_array(ptr, shape=(n,))
            return a, torch.from_numpy(a)

        def allreduce(ctx, buf, n, op):
            try:
                a, t = _tensor(buf, n)
                rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
                if self.device is not None:
                    g = t.to(self.device)
                    dist.all_reduce(g, op=rop)
                    t.copy_(g.cpu())
                else:
                    dist.all_reduce(t, op=rop)
                return 0
            except Exception:   # never raise through the C ABI
                return 1

        def allgather(ctx, send, recv, n):
            try:
                _, ts = _tensor(send, n)
                _, tr = _tensor(recv, n * world)
                if self.device is not None:
                    out = torch.empty(n * world, dtype=torch.float64, device=self.device)
                    dist.all_gather_into_tensor(out, ts.to(self.device))
                    tr.copy_(out.cpu())
                else:
                    parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, ts.clone())
                    tr.copy_(torch.cat(parts))
                return 0
            except Exception:
                return 1

        self._cbs = (api._ALLREDUCE_CB(allreduce), api._ALLGATHER_CB(allgather))   # keep the thunks alive
        self.c = api.Comm(rank, world, None, self._cbs[0], self._cbs[1])


def solve_window_sharded(api, gpu_model, seq, n_frames, dist, rank, world, init_params, beta0, beta_pose=5.0, beta_shape=25.0,
                         lambda_temporal=3.0, max_iters=100, device=None, constant=None, rccl=None):
    """One window of `n_frames` frames fitted by `world` ranks (one process per GPU): every rank builds the problem of its
    shard and calls bodyfit_solve_sharded (host callbacks over torch.distributed) or, with rccl = an api.Rccl communicator,
    bodyfit_solve_sharded_rccl.  Returns (x_local [f0:f1], beta, summary, shard)."""
    shard = make_shard(n_frames, world, rank)
    sl = slice_sequence(seq, shard)
    prob = api.Problem(gpu_model, sl["kp_offset"], sl["kp_id"], sl["kp_uv"], sl["intr"], sl["R0"], n_cols=86, use_shape=True,
                       beta_pose=beta_pose, beta_shape=beta_shape if shard.owns_shape_prior else 0.0,
                       lambda_temporal=lambda_temporal, temporal_halo=shard.halo)
    if rccl is not None:      # RCCL on the solve's device buffers and stream: nothing is staged through the host
        x, b, summ = prob.solve_sharded_rccl(local_params(init_params, shard), beta0, rccl, constant=constant, max_iters=max_iters)
    else:
        comm = TorchComm(api, dist, rank, world, device=device)
        x, b, summ = prob.solve_sharded(local_params(init_params, shard), beta0, comm.c, constant=constant, max_iters=max_iters)
    summ.exchanges = prob.last_exchange_count()     # all-gathers this solve issued (3 per LM iteration + 4 at the start)
    return x, b, summ, shard


def make_rccl(api, dist, rank: int, world: int, device: int):
    """An api.Rccl communicator for the ranks of a torch.distributed group: rank 0 draws the id (ncclGetUniqueId), the group's
    own host channel ships it, every rank joins (ncclCommInitRank)."""
    box = [api.Rccl.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return api.Rccl.create(box[0], rank, world, device)
