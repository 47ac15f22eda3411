#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench.py run
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (the TCC cannot hold both)
#   3. SQ counters in three more --pmc passes
# Outputs under gpurun_out/prof_<tag>/; tools/profile_digest.py turns them into the files kept in profiles/.
set -e
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --no-fit --no-ceres-path --no-c5-strong > $out/bench_stats.log 2>&1
grep "^{\"metric\"" $out/bench_stats.log | tail -1 > $out/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-fit --no-ceres-path --no-c5-strong --prewarm 0 --repeats 1 --steps 5 --warmup 2 > $out/pmc_$c.log 2>&1
done
# 3. SQ counters (matrix-pipe busy cycles, LDS activity and bank conflicts), three counters per pass
i=0
for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_sq$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-fit --no-ceres-path --no-c5-strong --prewarm 0 --repeats 1 --steps 5 --warmup 2 > $out/pmc_sq$i.log 2>&1
done
python3 tools/profile_digest.py $tag
# the digested files travel back through gpurun_out/ (profiles/ on the box is not merged; gpurun_out/ is capped at 64 MiB,
# so the raw traces stay behind)
mkdir -p gpurun_out/profiles_$tag
cp profiles/${tag}_* gpurun_out/profiles_$tag/
find $out -name "*.csv" -size +1M -delete   # (after the digest: the kernel trace is read there)
