#!/bin/bash
# SQ / GRBM counters of the CURRENT kernels (run through gpurun):  bash tools/pmc_sq_counters.sh <tag> [workload ...]
# One rocprofv3 --kernel-trace --pmc pass per counter set (never combined with other trace domains), the default bench command
# of each workload with --no-cpu; tools/summarize_sq.py turns the csv files into profiles/<tag>_<workload>_sq_counters.json.
TAG=${1:-r04_a}; shift
WL=${@:-headline}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in $WL; do
  O=$R/gpurun_out/$TAG/sq_$w
  mkdir -p $O
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
             "SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/bench.py --no-cpu --no-ref-width --workload $w --steps 3 --warmup 1 > $O/log$i.txt 2>&1 </dev/null
  done
  python $R/tools/summarize_sq.py $O $R/gpurun_out/$TAG/${TAG}_${w}_sq_counters.json "$w"
done
