cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t11
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','cold_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms')}, (j.get('warm_start') or {}).get('evals_per_s'))"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ss.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do python bench.py --no-cpu > gpurun_out/t11/b_$i.log 2>&1; pj gpurun_out/t11/b_$i.log; done
for w in c2 c4 c5 c3; do python bench.py --no-cpu --workload $w --steps 15 > gpurun_out/t11/b_$w.log 2>&1; pj gpurun_out/t11/b_$w.log; done
