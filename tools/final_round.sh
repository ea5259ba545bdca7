# Round profile (run through gpurun):  bash tools/final_round.sh <tag>     e.g. r03_a
TAG=${1:-r03_a}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG > /dev/null 2>&1
O=gpurun_out/$TAG
for w in c2 c3 c4 c5 posterior; do python bench.py --workload $w > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log | cut -c1-200; done
python bench.py --no-cpu --warm > $O/bench_warm.log 2>&1; tail -1 $O/bench_warm.log | python -c "import sys,json; print('warm', json.loads(sys.stdin.read()).get('warm_start'))"
python bench.py --gpus 2 > $O/bench_gpus2.log 2>&1; tail -1 $O/bench_gpus2.log | cut -c1-200
python bench.py --gpus 2 --workload c3 > $O/bench_c3_gpus2.log 2>&1; tail -1 $O/bench_c3_gpus2.log | cut -c1-200
SHARD_RANKS=2 SHARD_MODES=ss,coop python tools/shard_probe.py > $O/shard_probe.log 2>&1; tail -4 $O/shard_probe.log | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c5 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c5 --steps 5 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c3 --steps 5 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_posterior -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload posterior --steps 5 > /dev/null 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/dpp_lab.hip -o /tmp/dpp_lab && /tmp/dpp_lab > $GRAFT_REPO_ROOT/$O/dpp_lab.log 2>&1
tail -1 $GRAFT_REPO_ROOT/$O/bench_default.log | cut -c1-300
