cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t16
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms')})"; }
for v in 1 0 1 0; do SMCPP_POLL=$v python bench.py --no-cpu > gpurun_out/t16/b_p$v.log 2>&1; pj gpurun_out/t16/b_p$v.log; done
python bench.py --no-cpu --workload c2 > gpurun_out/t16/c2.log 2>&1; pj gpurun_out/t16/c2.log
python bench.py --no-cpu --workload c3 --steps 15 > gpurun_out/t16/c3.log 2>&1; pj gpurun_out/t16/c3.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
