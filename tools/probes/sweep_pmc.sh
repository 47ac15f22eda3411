# rocprofv3 --pmc passes over the sweep's kernels at 4,096 frames (both forms) and 256 frames (one launch); run through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4096 two" "4096 one" "256 one"; do
  set -- $cfg; tag=${1}_${2}; i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/sweep_pmc/${tag}_$i -o p --output-format csv -- python3 tools/probes/sweep_pmc.py $1 $2 > gpurun_out/sweep_pmc_${tag}_$i.log 2>&1
  done
  python3 tools/pmc_summary.py gpurun_out/sweep_pmc/${tag}_1 gpurun_out/sweep_pmc/${tag}_2 gpurun_out/sweep_pmc/${tag}_3 gpurun_out/sweep_pmc/${tag}_4 > gpurun_out/r5_sweep_pmc_${tag}.txt 2>&1
done
find gpurun_out/sweep_pmc -name "*.csv" -size +1M -delete
