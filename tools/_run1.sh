mkdir -p gpurun_out/t4
for k in 0 300 500 700 850; do SMCPP_SS_SKIP0=$k timeout 200 python bench.py --no-cpu > gpurun_out/t4/skip$k.log 2>&1; done
for k in 500 700; do SMCPP_SS_SKIP0=$k timeout 200 python bench.py --no-cpu --workload c3 > gpurun_out/t4/c3_skip$k.log 2>&1; done
for f in skip0 skip300 skip500 skip700 skip850 c3_skip500 c3_skip700; do tail -1 gpurun_out/t4/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['split_ms'].items() if k in ('host_prep_ms','chains_wall_ms','stats_ms','finalize_ms','fwd_passes')})"; done
