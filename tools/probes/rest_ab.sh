cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
for r in 0 0.3 1 2 3 6; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --prewarm-rest-ms $r --no-cpu-baseline --no-fit --no-c5-strong --no-ceres-path > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("/tmp/b.json")); t=d["timing"]
print("round $rnd rest $r ms: median %.2f us  min %.2f  max %.2f  cold %.2f | period %.2f kernel-only %.2f" % (d["ms_per_step"]*1e3, t["ms_per_step_min"]*1e3, t["ms_per_step_max"]*1e3, t["cold_start"]["ms_per_step"]*1e3, d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["kernel_only"]["avg_launch_ms"]*1e3), flush=True)
PY
done; done
