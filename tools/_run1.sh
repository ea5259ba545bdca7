mkdir -p gpurun_out/t2
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t2/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t2/tests.log
for i in 1 2; do timeout 200 python bench.py --no-cpu > gpurun_out/t2/head$i.log 2>&1; done
SMCPP_STATS_VARIANT=4 timeout 200 python bench.py --no-cpu > gpurun_out/t2/head_old.log 2>&1
timeout 200 python bench.py --no-cpu --workload c2 > gpurun_out/t2/c2.log 2>&1
timeout 300 python bench.py --no-cpu --workload c5 > gpurun_out/t2/c5.log 2>&1
timeout 300 python bench.py --no-cpu --workload c3 > gpurun_out/t2/c3.log 2>&1
tail -3 gpurun_out/t2/tests.log
for f in head1 head2 head_old c2 c5 c3; do tail -1 gpurun_out/t2/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done
