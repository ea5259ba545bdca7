cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t21
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms','fwd_passes')})"; }
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or posterior or G7 or chunking" 2>&1 | tail -4
python bench.py --no-cpu --workload posterior > gpurun_out/t21/post.log 2>&1; pj gpurun_out/t21/post.log || tail -5 gpurun_out/t21/post.log
SMCPP_HYB_TH=3 python bench.py --no-cpu --workload posterior > gpurun_out/t21/post_t3.log 2>&1; pj gpurun_out/t21/post_t3.log
SMCPP_HYB_TH=12 python bench.py --no-cpu --workload posterior > gpurun_out/t21/post_t12.log 2>&1; pj gpurun_out/t21/post_t12.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t21/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload posterior --steps 3 --warmup 1 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/t21/stats/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_chain_ss' in r['Kernel_Name']]
for r in rows[-8:]:
    print(round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), 'us', r['Kernel_Name'][:50])
PY
