cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t26
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','cold_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms','staging_ms','eigensystems_ms')})"; }
for i in 1 2; do python bench.py --no-cpu --workload c5 --steps 20 > gpurun_out/t26/c5_$i.log 2>&1; pj gpurun_out/t26/c5_$i.log; done
python bench.py --no-cpu > gpurun_out/t26/b.log 2>&1; pj gpurun_out/t26/b.log
timeout 600 python -m pytest tests/test_gpu_ss.py tests/test_gpu_parity.py -x -q -m gpu -k "ss or c5 or golden_stats" 2>&1 | tail -2
