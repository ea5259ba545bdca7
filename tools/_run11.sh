cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t17
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms')})"; }
python bench.py --no-cpu > gpurun_out/t17/b_base.log 2>&1; pj gpurun_out/t17/b_base.log
SMCPP_SLAB_ROWS=256 python bench.py --no-cpu > gpurun_out/t17/b_s256.log 2>&1; pj gpurun_out/t17/b_s256.log
SMCPP_SLAB_ROWS=192 python bench.py --no-cpu > gpurun_out/t17/b_s192.log 2>&1; pj gpurun_out/t17/b_s192.log
SMCPP_S1_FUSE=1 python bench.py --no-cpu > gpurun_out/t17/b_f1.log 2>&1; pj gpurun_out/t17/b_f1.log
SMCPP_S1_FUSE=1 SMCPP_SLAB_ROWS=256 python bench.py --no-cpu > gpurun_out/t17/b_f1s256.log 2>&1; pj gpurun_out/t17/b_f1s256.log
python bench.py --no-cpu > gpurun_out/t17/b_base2.log 2>&1; pj gpurun_out/t17/b_base2.log
