mkdir -p gpurun_out/t10
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "posterior or gamma or golden" > gpurun_out/t10/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t10/tests.log
tail -4 gpurun_out/t10/tests.log
for w in posterior64 posterior; do timeout 300 python bench.py --no-cpu --workload $w > gpurun_out/t10/$w.log 2>&1; tail -1 gpurun_out/t10/$w.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done
