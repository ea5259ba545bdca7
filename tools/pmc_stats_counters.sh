# rocprofv3 PMC passes over the statistics kernels of the whole-genome workload (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc6
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc6/p$i -- python $R/bench.py --no-cpu --no-ref-width --workload c3 --steps 2 --warmup 1 > $R/gpurun_out/pmc6/log$i.txt 2>&1 </dev/null
done
ls $R/gpurun_out/pmc6/*/*/ | head
