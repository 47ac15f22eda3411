# Run through gpurun from the repo root: all rocprofv3 passes of a round (tools/profile_round.sh) plus an untraced bench line.
TAG=${1:-r2_02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1
timeout -k 10 500 python3 bench.py > gpurun_out/bench_${TAG}_untraced.log 2>&1
grep '^{"metric"' gpurun_out/bench_${TAG}_untraced.log | tail -1 > gpurun_out/profiles_${TAG}/${TAG}_bench_untraced.json
tail -3 gpurun_out/prof_${TAG}.log
ls gpurun_out/profiles_${TAG}
