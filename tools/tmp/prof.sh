cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r2_02 > gpurun_out/prof_r2_02.log 2>&1
timeout -k 10 500 python3 bench.py > gpurun_out/bench_r2_02_untraced.log 2>&1
grep '^{"metric"' gpurun_out/bench_r2_02_untraced.log | tail -1 > gpurun_out/profiles_r2_02/r2_02_bench_untraced.json
tail -3 gpurun_out/prof_r2_02.log
ls gpurun_out/profiles_r2_02
