cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t12
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','fwd_passes')})"; }
python bench.py --no-cpu --workload c3 --steps 15 > gpurun_out/t12/c3_w1.log 2>&1; pj gpurun_out/t12/c3_w1.log
SMCPP_SS_WPC=2 python bench.py --no-cpu --workload c3 --steps 15 > gpurun_out/t12/c3_w2.log 2>&1; pj gpurun_out/t12/c3_w2.log
SMCPP_SS_WPC=3 python bench.py --no-cpu --workload c3 --steps 15 > gpurun_out/t12/c3_w3.log 2>&1; pj gpurun_out/t12/c3_w3.log
python bench.py --no-cpu > gpurun_out/t12/b.log 2>&1; pj gpurun_out/t12/b.log
SHARD_RANKS=1 SHARD_MODES=ss python tools/shard_probe.py 2>&1 | tail -1 | cut -c1-200
SMCPP_SS_WPC=2 SHARD_RANKS=1 SHARD_MODES=ss python tools/shard_probe.py 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu --gpus 2 --workload c3 --steps 15 > gpurun_out/t12/c3_g2.log 2>&1; pj gpurun_out/t12/c3_g2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chunking or full_size or golden" 2>&1 | tail -3
