cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t15
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','fwd_passes')})"; }
python bench.py --no-cpu --workload c3 > gpurun_out/t15/c3.log 2>&1; pj gpurun_out/t15/c3.log
for i in 1 2; do python bench.py --no-cpu > gpurun_out/t15/b$i.log 2>&1; pj gpurun_out/t15/b$i.log; done
for w in c2 c4 c5 posterior; do python bench.py --no-cpu --workload $w --steps 15 > gpurun_out/t15/b_$w.log 2>&1; pj gpurun_out/t15/b_$w.log; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or span1 or gamma_sums" 2>&1 | tail -3
