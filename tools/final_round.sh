# Round profile (run through gpurun):  bash tools/final_round.sh <tag>     e.g. r04_k
# default bench (CPU leg included) + rocprofv3 kernel stats + FETCH / WRITE PMC passes (tools/profile_round.sh), SQ counters of the
# headline and the whole genome, every workload's bench line, Q-with-gradient, warm start, the N > 1 path on one device (2 ranks;
# 8 ranks with --check for c3 and c4), one rank's shard of the 8-GPU genome run, kernel stats of c3 / c5 / posterior / qgrad,
# the DPP issue-cost lab, the stream-hop cost lab, a host trace of the headline's E-step.
TAG=${1:-r05_b}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG > /dev/null 2>&1
O=gpurun_out/$TAG
bash tools/pmc_sq_counters.sh $TAG headline c3 c5 > $O/sq.log 2>&1
for w in c2 c3 c4 c5 posterior posterior64 posterior128 pbinned; do python bench.py --workload $w > $O/bench_$w.log 2>&1; grep '^{"metric"' $O/bench_$w.log | tail -1 | cut -c1-200; done
python bench.py --workload qgrad > $O/bench_qgrad.log 2>&1; grep '^{"metric"' $O/bench_qgrad.log | tail -1 | cut -c1-260
SMCPP_BENCH_THREADS=1 python bench.py --no-cpu > $O/bench_default_1thread.log 2>&1; grep '^{"metric"' $O/bench_default_1thread.log | tail -1 | cut -c1-200
python bench.py --no-cpu --warm > $O/bench_warm.log 2>&1; grep '^{"metric"' $O/bench_warm.log | tail -1 | python -c "import sys,json; print('warm', json.loads(sys.stdin.read()).get('warm_start'))"
python bench.py --gpus 2 --no-cpu > $O/bench_gpus2.log 2>&1; grep '^{"metric"' $O/bench_gpus2.log | tail -1 | cut -c1-200
python bench.py --gpus 8 --workload c3 --no-cpu --check --steps 10 > $O/bench_c3_gpus8_check.log 2>&1; grep '^{"metric"' $O/bench_c3_gpus8_check.log | tail -1 | cut -c1-200
python bench.py --gpus 8 --workload c4 --no-cpu --check --steps 10 > $O/bench_c4_gpus8_check.log 2>&1; grep '^{"metric"' $O/bench_c4_gpus8_check.log | tail -1 | cut -c1-200
SHARD_RANKS=2 SHARD_MODES=ss python tools/shard_probe.py > $O/shard_probe.log 2>&1; tail -2 $O/shard_probe.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
# what the E-step's exchange adds to an eval (one-rank RCCL group: host wait / stream-ordered through torch / issued by the engine)
timeout 300 python -m pytest tests/test_gpu_multi.py -m gpu -q -s -k world_of_one 2>&1 | grep -a "exchange cost" | tail -1 > $O/exchange_cost.log; cat $O/exchange_cost.log
cd /tmp && export TMPDIR=/tmp
for w in c3 c5 posterior posterior64 posterior128 qgrad; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ref-width --workload $w --steps 5 > /dev/null 2>&1
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/dpp_lab.hip -o /tmp/dpp_lab && /tmp/dpp_lab > $GRAFT_REPO_ROOT/$O/dpp_lab.log 2>&1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/sync_lab.hip -o /tmp/sync_lab && /tmp/sync_lab > $GRAFT_REPO_ROOT/$O/sync_lab.log 2>&1
SMCPP_HOST_TRACE=1 python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 4 --warmup 2 2>&1 | grep host-trace | sed -n 40,60p > $GRAFT_REPO_ROOT/$O/host_trace.log
# two-population preparation: where the host phase of config C4 goes (transition / joint CSFS / assembly; inside the joint CSFS)
SMCPP_HOST_TIMING=1 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c4 --steps 8 --warmup 3 2>&1 | grep -a "prep2\|jcsfs" | tail -8 > $GRAFT_REPO_ROOT/$O/c4_host_timing.log
# the statistics phase of the headline and of config C5, kernel by kernel (tools/stats_timeline.py)
for w in headline c5; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ref-width --workload $w --steps 6 --warmup 3 > /dev/null 2>&1 </dev/null
  python $GRAFT_REPO_ROOT/tools/stats_timeline.py /tmp/tl_$w > $GRAFT_REPO_ROOT/$O/stats_timeline_$w.txt 2>&1
done
grep '^{"metric"' $GRAFT_REPO_ROOT/$O/bench_default.log | tail -1 | cut -c1-300
