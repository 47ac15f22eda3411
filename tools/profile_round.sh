#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench.py run
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (the TCC cannot hold both)
# Outputs under gpurun_out/prof_<tag>/; tools/profile_digest.py turns them into the files kept in profiles/.
set -e
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py > $out/bench_stats.log 2>&1
grep "^{\"metric\"" $out/bench_stats.log | tail -1 > $out/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/pmc_$c.log 2>&1
done
python3 tools/profile_digest.py $tag
