mkdir -p gpurun_out/t9
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/t9/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t9/tests.log
tail -3 gpurun_out/t9/tests.log
for i in 1 2 3; do timeout 200 python bench.py --no-cpu > gpurun_out/t9/h$i.log 2>&1; tail -1 gpurun_out/t9/h$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('head$i', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done
for b in 0 1 3; do SMCPP_OMP_BLOCKTIME=$b timeout 200 python bench.py --no-cpu --workload c4 > gpurun_out/t9/c4_$b.log 2>&1; tail -1 gpurun_out/t9/c4_$b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 blocktime $b', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done
