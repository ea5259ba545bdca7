# Run config C4 once; if the eval shows the slow mode (time the caller's E_step spends outside the engine's host phase and the
# device intervals > 0.12 ms - VERDICT r05 item 4), run the matrix of suspects on THIS box.   bash tools/c4_watch.sh <tag>
TAG=${1:-r06_c4w}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['split_ms']
print('%-22s %7.1f evals/s  %.3f ms | caller %s | host %.3f device %.3f unaccounted %.3f | %s' % (sys.argv[1], d['value'], d['ms_per_step'],
      {k: round(v, 3) for k, v in (d.get('caller_ms') or {}).items() if k != 'note'}, s['host_prep_ms'], s['device_total_ms'], d.get('unaccounted_ms') or 0,
      d.get('gpu_power_state')))" "$1"; }
run() { name=$1; shift; "$@" python bench.py --workload c4 --no-cpu --no-ref-width --steps 40 --warmup 10 2> $O/c4_$name.err | line $name | tee -a $O/c4_watch.txt; }
run default env A=1
GAP=$(tail -1 $O/c4_watch.txt | sed 's/.*unaccounted \([0-9.]*\).*/\1/')
if python -c "import sys; sys.exit(0 if float('$GAP') > 0.12 else 1)"; then
  echo "slow mode on this box (gap $GAP ms): suspects" | tee -a $O/c4_watch.txt
  run default_again env A=1
  run threads1 env SMCPP_BENCH_THREADS=1
  run threads4 env SMCPP_BENCH_THREADS=4
  run blocktime0 env SMCPP_OMP_BLOCKTIME=0
  run nopoll env SMCPP_POLL=0
  run hsa_no_interrupt env HSA_ENABLE_INTERRUPT=0
  run node0_16 taskset -c 0-15
  run node1_16 taskset -c 64-79
  run hwq8 env GPU_MAX_HW_QUEUES=8
  run default_third env A=1
  SMCPP_HOST_TRACE=1 python bench.py --workload c4 --no-cpu --no-ref-width --steps 6 --warmup 3 2>&1 | grep -a host-trace | tail -60 > $O/c4_host_trace.log
  rocm-smi --showperflevel --showclocks > $O/smi.txt 2>&1
fi
