cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t13
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','fwd_passes')})"; }
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --workload c3 > gpurun_out/t13/bench_c3.log 2>&1; pj gpurun_out/t13/bench_c3.log
python bench.py --no-cpu --gpus 2 --workload c3 > gpurun_out/t13/bench_c3_gpus2.log 2>&1; pj gpurun_out/t13/bench_c3_gpus2.log
SHARD_RANKS=2 SHARD_MODES=ss,coop python tools/shard_probe.py > gpurun_out/t13/shard_probe.log 2>&1; tail -4 gpurun_out/t13/shard_probe.log | cut -c1-160
python bench.py --no-cpu > gpurun_out/t13/b.log 2>&1; pj gpurun_out/t13/b.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t13/stats_c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c3 --steps 5 > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/t13/stats_c3/*/*kernel_stats.csv | cut -c1-130
