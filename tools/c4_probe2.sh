# NUMA hypothesis for the two speeds of config C4 (tools/c4_probe.sh found no gap on a fast box): the same evals with the process
# confined to the GPU's NUMA node, to the other node, and left free.   bash tools/c4_probe2.sh <tag>   (through gpurun)
TAG=${1:-r06_c4b}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
GN=$(rocm-smi --showtoponuma 2>/dev/null | grep -i "Numa Node" | head -1 | sed 's/.*: //')
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "gpu numa node: $GN   node0: $N0   node1: $N1" > $O/box.txt
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['split_ms']
print('%-28s %7.1f evals/s  %.3f ms | caller %s | host_prep %.3f device %.3f' % (sys.argv[1], d['value'], d['ms_per_step'],
      {k: round(v, 3) for k, v in (d.get('caller_ms') or {}).items() if k != 'note'}, s['host_prep_ms'], s['device_total_ms']))" "$1"; }
run() { name=$1; w=$2; shift; shift; "$@" python bench.py --workload $w --no-cpu --no-ref-width --steps 40 --warmup 10 2> $O/${w}_$name.err | line ${w}_$name; }
for w in c4 headline; do
  run free $w env A=1
  run node0 $w taskset -c $N0
  run node1 $w taskset -c $N1
  run node0_first16 $w taskset -c 0-15
  run node1_first16 $w taskset -c 64-79
  run free_again $w env A=1
done
