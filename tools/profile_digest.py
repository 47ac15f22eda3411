"""Digest gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small files kept under profiles/:
  <tag>_kernel_stats.csv     rocprofv3 --stats kernel table of the default bench.py run
  <tag>_bench.json           the bench line printed by that run
  <tag>_kernel_stats_post_prewarm.json   the sweep kernel's launch durations behind the cold bracket and the prewarm sweeps
  <tag>_pmc_traffic.json     HBM bytes per launch and kernel from the FETCH_SIZE / WRITE_SIZE passes"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
st = glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], f"profiles/{tag}_kernel_stats.csv")
line = open(src + "/bench.json").read().strip()
bench = json.loads(line)
json.dump(bench, open(f"profiles/{tag}_bench.json", "w"), indent=1)
# The --stats table averages over EVERY launch of the run, the clock-ramp launches of the cold bracket and the untimed prewarm
# sweeps included.  The figure roofline.avg_launch_ms must be reproducible from is the kernel's duration over the launches
# BEHIND them (the timed brackets and the in-bench HIP-event sample): taken here from the kernel trace itself, in dispatch order.
tr = glob.glob(src + "/stats/**/*kernel_trace.csv", recursive=True)
if tr:
    rows = list(csv.DictReader(open(tr[0])))
    tm = bench.get("timing", {})
    skip = int(bench.get("warmup", 0)) + int(bench.get("steps", 0)) + int(tm.get("prewarm_steps", 0))   # cold bracket + prewarm
    by = defaultdict(list)
    for r in rows:
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if m:
            by[m.group(0)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    post = {}
    for k, v in by.items():
        v.sort()
        d = sorted(x[1] for x in v[skip:]) if k in ("k_sweep_roles", "k_frame_resjac", "k_mesh_blend_lbs") else sorted(x[1] for x in v)
        if not d:
            continue
        q = lambda f: d[min(len(d) - 1, int(f * len(d)))]
        post[k] = {"launches": len(d), "skipped_leading_launches": skip if len(d) != len(v) else 0, "avg_ns": round(sum(d) / len(d), 1),
                   "median_ns": q(0.5), "p10_ns": q(0.1), "p90_ns": q(0.9), "min_ns": d[0], "max_ns": d[-1]}
    json.dump({"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-fit --no-ceres-path --no-c5-strong (the kernel "
                         "trace of the same run as <tag>_kernel_stats.csv), durations = End - Start per dispatch, dispatch order",
               "note": "sweep kernels: the first warmup + steps + prewarm_steps launches (cold bracket, untimed prewarm) are dropped; "
                       "what remains are the timed brackets' launches and the in-bench HIP-event sample behind them",
               "bench_roofline_avg_launch_ms": bench.get("roofline", {}).get("avg_launch_ms"),
               "bench_ms_per_step_median": bench.get("ms_per_step"), "kernels": post},
              open(f"profiles/{tag}_kernel_stats_post_prewarm.json", "w"), indent=1)
kern = defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = defaultdict(list)
    for path in glob.glob(f"{src}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            m = re.search(r"k_[a-z0-9_]+", row["Kernel_Name"])
            if m and row["Counter_Name"] == c:
                vals[m.group(0)].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        kern[k][c + "_KiB"] = round(sum(v) / len(v), 1)
for k, d in kern.items():
    d["hbm_bytes"] = int((2 * d.get("FETCH_SIZE_KiB", 0.0) + d.get("WRITE_SIZE_KiB", 0.0)) * 1024)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2",
    "workload": "c3", "frames_per_gpu": bench["config"]["frames_per_gpu"],
    "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (average over launches); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
             "(gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section)",
    "kernels": kern,
}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes"] for k, v in kern.items()}))
# SQ counter passes -> mean per launch and kernel
sq = defaultdict(dict)
for path in glob.glob(f"{src}/pmc_sq*/**/*counter_collection.csv", recursive=True):
    vals = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        m = re.search(r"k_[a-z0-9_]+", row["Kernel_Name"])
        if m:
            vals[m.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in vals.items():
        for c, v in cs.items():
            sq[k][c] = round(sum(v) / len(v), 1)
if sq:
    for k, d in sq.items():
        if d.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy_over_sq_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"], 4)
        if d.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in d:
            d["lds_conflict_share"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
    json.dump({"source": "rocprofv3 --pmc <three SQ counters per pass> -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2",
               "note": "sums over all shader engines / CUs per launch, as rocprofv3 reports them", "kernels": sq},
              open(f"profiles/{tag}_pmc_sq.json", "w"), indent=1)
