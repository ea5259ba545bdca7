cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t9
pj() { tail -1 $1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['split_ms'].items() if k in ('host_prep_ms','cold_prep_ms','chains_wall_ms','stats_ms','finalize_ms','hmm_only_ms')})"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ss.py -x -q -m gpu 2>&1 | tail -6
python bench.py --no-cpu --workload posterior > gpurun_out/t9/b_post.log 2>&1; pj gpurun_out/t9/b_post.log
python bench.py --no-cpu > gpurun_out/t9/b_default.log 2>&1; pj gpurun_out/t9/b_default.log
SMCPP_SS=0 python bench.py --no-cpu > gpurun_out/t9/b_dense.log 2>&1; pj gpurun_out/t9/b_dense.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t9/stats_post -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload posterior --steps 5 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/t9/stats_post/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
n=[int(r['Calls']) for r in rows if 'k_loglik_final' in r['Name']][0]
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):4d} {float(r['TotalDurationNs'])/n/1e3:9.1f} us/step")
PY
