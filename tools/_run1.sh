mkdir -p gpurun_out/t3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t3/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t3/tests.log
for i in 1 2; do timeout 200 python bench.py --no-cpu > gpurun_out/t3/head$i.log 2>&1; done
SMCPP_SLAB_ROWS=64 timeout 200 python bench.py --no-cpu > gpurun_out/t3/head_s64.log 2>&1
SMCPP_SLAB_ROWS=96 timeout 200 python bench.py --no-cpu > gpurun_out/t3/head_s96.log 2>&1
SMCPP_SLAB_ROWS=192 timeout 200 python bench.py --no-cpu > gpurun_out/t3/head_s192.log 2>&1
timeout 200 python bench.py --no-cpu --workload c2 > gpurun_out/t3/c2.log 2>&1
timeout 300 python bench.py --no-cpu --workload c5 > gpurun_out/t3/c5.log 2>&1
timeout 300 python bench.py --no-cpu --workload posterior > gpurun_out/t3/posterior.log 2>&1
tail -3 gpurun_out/t3/tests.log
for f in head1 head2 head_s64 head_s96 head_s192 c2 c5 posterior; do tail -1 gpurun_out/t3/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done
