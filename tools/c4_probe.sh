# Where does an eval of config C4 (two populations) spend the time that neither the engine's host phase nor its device intervals
# account for, and why does it differ between boxes?   bash tools/c4_probe.sh <tag>   (through gpurun)
TAG=${1:-r06_c4}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
{
  echo "nproc $(nproc)   cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  grep Cpus_allowed_list /proc/self/status
  lscpu | grep -i 'model name\|thread(s) per core\|socket\|numa\|mhz'
  rocm-smi --showtoponuma 2>/dev/null | grep -i numa
  cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
} > $O/box.txt 2>&1
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['split_ms']
print('%-28s %7.1f evals/s  %.3f ms | caller %s | host_prep %.3f device %.3f | throttle %s' % (sys.argv[1], d['value'], d['ms_per_step'],
      {k: round(v, 3) for k, v in (d.get('caller_ms') or {}).items() if k != 'note'}, s['host_prep_ms'], s['device_total_ms'], d.get('cpu_throttle_in_timed_region')))" "$1"; }
run() { name=$1; shift; env "$@" python bench.py --workload c4 --no-cpu --no-ref-width --steps 40 --warmup 10 2> $O/c4_$name.err | line $name; }
python bench.py --no-cpu --no-ref-width 2>/dev/null | line headline
run default             A=1
run default_again       A=1
run blocktime0          SMCPP_OMP_BLOCKTIME=0
run blocktime200        SMCPP_OMP_BLOCKTIME=200
run threads4            SMCPP_BENCH_THREADS=4
run threads8            SMCPP_BENCH_THREADS=8
run threads1            SMCPP_BENCH_THREADS=1
run nopoll              SMCPP_POLL=0
run bind_close          OMP_PROC_BIND=close OMP_PLACES=cores
run default_third       A=1
SMCPP_HOST_TRACE=1 python bench.py --workload c4 --no-cpu --no-ref-width --steps 6 --warmup 3 2>&1 | grep -a host-trace | tail -80 > $O/c4_host_trace.log
SMCPP_HOST_TIMING=1 python bench.py --workload c4 --no-cpu --no-ref-width --steps 8 --warmup 3 2>&1 | grep -a "prep2\|jcsfs" | tail -4 > $O/c4_host_timing.log
