"""Digest gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small files kept under profiles/:
  <tag>_kernel_stats.csv     rocprofv3 --stats kernel table of the default bench.py run
  <tag>_bench.json           the bench line printed by that run
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
