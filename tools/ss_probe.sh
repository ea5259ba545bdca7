#!/bin/bash
# quick probe of the scan chains on the GPU box: bench line + rocprofv3 kernel statistics of the default workload
TAG=${1:-ssprobe}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench.log 2>&1 </dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/stats.log 2>&1 </dev/null
python - <<PY
import json,glob,csv
l=[x for x in open("$O/bench.log") if x.startswith("{")][-1]; j=json.loads(l)
print("evals/s",j["value"],"ms",j["ms_per_step"]); print(j["split_ms"])
f=glob.glob("$O/stats/*/*kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:14]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
