mkdir -p gpurun_out/t7
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t7/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t7/tests.log
tail -3 gpurun_out/t7/tests.log
for w in headline c2 c4; do for v in new old; do
  if [ $v = new ]; then env="X=1"; else env="SMCPP_S1_FUSE=0"; fi
  env $env timeout 200 python bench.py --no-cpu --workload $w > gpurun_out/t7/$w$v.log 2>&1; tail -1 gpurun_out/t7/$w$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w $v', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms')})"; done; done
