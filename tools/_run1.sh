mkdir -p gpurun_out/t5
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t5/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t5/tests.log
tail -3 gpurun_out/t5/tests.log
for w in posterior c5 qgrad qgrad; do timeout 300 python bench.py --no-cpu --workload $w > gpurun_out/t5/$w.log 2>&1; grep '^{"metric"' gpurun_out/t5/$w.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$w', round(d['value'],1), round(d['ms_per_step'],4), r.get('bound'), r.get('achieved'), r.get('peak'), r.get('frac'), d.get('speedup_vs_host_path'))"; done
timeout 200 python bench.py --no-cpu > gpurun_out/t5/head.log 2>&1; tail -1 gpurun_out/t5/head.log | cut -c1-250
