#!/bin/bash
# rocprofv3 kernel statistics of the LM fits (run through gpurun from the repo root): the C3 fit, the C4 staged run, one
# 20-frame C4 window and the 545-frame C5 window of one GPU.  The digested tables land in gpurun_out/profiles_<tag>/.
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_$tag
for spec in "c3" "c4" "window 20" "window 545"; do
  name=$(echo $spec | tr ' ' '_')
  out=gpurun_out/prof_fit_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o f --output-format csv -- python3 tools/fit_prof.py $spec > $out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/profiles_$tag/${tag}_fit_${name}_kernel_stats.csv
  grep "^{'frames" $out.log | tail -1 > gpurun_out/profiles_$tag/${tag}_fit_${name}.txt
  find $out -name "*.csv" -size +1M -delete
done
ls gpurun_out/profiles_$tag
