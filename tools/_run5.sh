cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t10
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','cold_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms')})"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ss.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -6
for v in 1 0 1 0; do SMCPP_S1_FUSE=$v python bench.py --no-cpu > gpurun_out/t10/b_f$v.log 2>&1; pj gpurun_out/t10/b_f$v.log; done
for w in c2 c3 c4 c5; do python bench.py --no-cpu --workload $w --steps 15 > gpurun_out/t10/b_$w.log 2>&1; pj gpurun_out/t10/b_$w.log; done
SMCPP_S1_FUSE=0 python bench.py --no-cpu --workload c3 --steps 15 > gpurun_out/t10/b_c3_f0.log 2>&1; pj gpurun_out/t10/b_c3_f0.log
SMCPP_S1_FUSE=0 python bench.py --no-cpu --workload c4 --steps 15 > gpurun_out/t10/b_c4_f0.log 2>&1; pj gpurun_out/t10/b_c4_f0.log
