cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/wprof -o w --output-format csv -- python3 tools/tmp/c3prof.py > gpurun_out/wprof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/wprof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
print("total kernel ms", tot/1e6)
PY
tail -2 gpurun_out/wprof.log
find gpurun_out/wprof -name "*.csv" -size +2M -delete
