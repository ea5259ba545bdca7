cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t25
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms','fwd_passes')})"; }
python bench.py --no-cpu --workload posterior --steps 20 > gpurun_out/t25/p_11.log 2>&1; pj gpurun_out/t25/p_11.log
SMCPP_GAMMA_SIDE=0 python bench.py --no-cpu --workload posterior --steps 20 > gpurun_out/t25/p_01.log 2>&1; pj gpurun_out/t25/p_01.log
SMCPP_SPEC_GAMMA=0 python bench.py --no-cpu --workload posterior --steps 20 > gpurun_out/t25/p_10.log 2>&1; pj gpurun_out/t25/p_10.log
SMCPP_GAMMA_SIDE=0 SMCPP_SPEC_GAMMA=0 python bench.py --no-cpu --workload posterior --steps 20 > gpurun_out/t25/p_00.log 2>&1; pj gpurun_out/t25/p_00.log
