cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t23
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms','fwd_passes')})"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ss.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do python bench.py --no-cpu --workload posterior > gpurun_out/t23/post_$i.log 2>&1; pj gpurun_out/t23/post_$i.log; done
python bench.py --no-cpu > gpurun_out/t23/b.log 2>&1; pj gpurun_out/t23/b.log
