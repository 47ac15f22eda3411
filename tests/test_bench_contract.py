"""bench.py keeps the driver's contract: one JSON line with the agreed keys at N = 1, and the N > 1 launch form
(torch.distributed.run, one rank per GPU) runs end to end -- rehearsed on the one-GPU box with both ranks on cuda:0 and
gloo in place of RCCL (BENCH_SHARE_DEVICE0 / BENCH_BACKEND, switches the driver never sets)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert len([l for l in out.stdout.splitlines() if l.strip()]) == 1, out.stdout[:600]   # ONE line (RCCL's banner goes to stderr)
    d = _last_json(out.stdout)
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["vs_baseline"] is None
    assert d["unit"] == "evals/s" and d["value"] > 1e5 and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - d["config"]["frames_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "traffic_source" in r
    # how ms_per_step was taken: the median of >= 7 brackets of K steps behind the prewarm, with the spread and the cold-start
    # bracket beside it; and no in-launch wait of any timed sweep ran out
    tm = d["timing"]
    assert tm["repeats"] >= 7 and len(tm["ms_per_step_all"]) == tm["repeats"]
    assert tm["ms_per_step_min"] <= tm["ms_per_step_median"] <= tm["ms_per_step_max"] and tm["ms_per_step_median"] == d["ms_per_step"]
    assert tm["cold_start"]["ms_per_step"] > 0 and d["sweep_timeouts"] == 0
    # the second half of the metric: frames/sec to convergence of c2 / c3 / c4, with iteration and sweep counts
    fit = d["fit"]
    assert fit["unit"] == "frames/s"
    for k in ("c2", "c3", "c4", "c5_staged", "c5_window"):
        assert fit[k]["frames_per_s"] > 0 and fit[k]["frames"] >= 1
    assert fit["c2"]["iterations"] >= 1 and fit["c2"]["sweeps"] >= 1 and fit["c3"]["converged"] >= 250
    assert fit["c4"]["windows"] == 9 and fit["c4"]["anchors"] == 13
    for k in ("c2", "c3", "c4", "window_20", "c5_staged", "c5_window"):   # reproducible form: median, cost, launches, us per iteration
        assert fit[k]["launches_per_iteration"] > 0 and fit[k]["us_per_iteration"] > 0 and len(fit[k]["seconds_all"]) >= 1
    assert fit["c4"]["stage2_final_cost"] < fit["c4"]["stage2_initial_cost"]
    # the Ceres-kept path next to the resident rate: sweeps with their PCIe copies + every block's Evaluate
    assert d["pcie_inclusive_evals_per_s"] > 0
    cp = d["ceres_path"]
    assert cp["c3"]["points_per_s"] > 0 and cp["c3"]["blocks"] == 256 * 25 + 2 * 256 and cp["c4_window"]["blocks_per_s"] > 0
    # configs[4] in the same line: at N = 1 the whole 1024-frame window on the one GPU, through a ONE-rank RCCL communicator of the
    # library (the all-reduce really is issued: rccl_ranks is what ncclCommCount reports)
    cs = d["c5_strong"]
    assert cs["n_gpus"] == 1 and cs["window"] == 1024 and cs["frames_per_gpu"] == 1024 and cs["rccl_ranks"] == 1
    assert cs["sweep"]["evals_per_s"] > 1e6 and cs["fit"]["frames_per_s"] > 100 and cs["fit"]["termination"] == 0
    assert cs["fit"]["final_cost"] < 0.05 * cs["fit"]["initial_cost"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (no WORLD_SIZE): bench.py starts its ranks itself as a child torch.distributed.run
    and relays rank 0's line — the form a driver that just runs `bench.py --gpus N` gets.  Rehearsed with both ranks on cuda:0
    over gloo.  The line keeps C3 as `value` (weak scaling) and carries configs[4] sharded over the two ranks as `c5_strong`:
    the sweep with its all-reduce, the fit with three exchanges per LM iteration."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_SHARE_DEVICE0="1", BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--window", "64",
                          "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # ONE JSON line on stdout, whatever the ranks printed
    d = json.loads(lines[0])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1e5 and d["steps"] == 5
    assert abs(d["value"] - 2 * d["config"]["frames_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    cs = d["c5_strong"]
    assert cs["n_gpus"] == 2 and cs["window"] == 64 and cs["frames_per_gpu"] == 32 and cs["scaling"] == "strong"
    assert cs["sweep"]["evals_per_s"] > 0 and cs["sweep"]["collective"]
    assert cs["fit"]["termination"] == 0 and cs["fit"]["final_cost"] < 0.05 * cs["fit"]["initial_cost"]
    f = cs["fit"]
    assert f["exchanges_per_iteration"] == 3 and (f["exchanges_total"] - 4) % 3 == 0
    assert f["iterations"] <= f["iterations_launched"] <= f["iterations"] + 3


@pytest.mark.gpu
@pytest.mark.parametrize("workload,scaling", [("c3", "weak"), ("c5", "strong")])
def test_bench_two_ranks_rehearsal(workload, scaling):
    env = dict(os.environ, BENCH_SHARE_DEVICE0="1", BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", workload,
           "--no-cpu-baseline"] + (["--window", "128"] if workload == "c5" else [])
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_bench_c5_fit_line(world):
    """--workload c5 --fit: a step is one complete fit of the window; N = 2 rehearsed with both ranks on cuda:0 over gloo."""
    env = dict(os.environ, BENCH_SHARE_DEVICE0="1", BENCH_BACKEND="gloo")
    base = ["bench.py", "--gpus", str(world), "--workload", "c5", "--fit", "--window", "64", "--steps", "2", "--warmup", "1"]
    if world == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", "29547"] + base
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["unit"] == "frames/s" and d["value"] > 0
    assert d["fit"]["termination"] == 0 and d["fit"]["final_cost"] < 0.05 * d["fit"]["initial_cost"]
