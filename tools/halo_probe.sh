#!/bin/bash
# Per-launch durations of k_chain_ss for a few halo lengths (run through gpurun):  bash tools/halo_probe.sh <tag> [workload]
TAG=${1:-r04_e}; W=${2:-headline}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "2800 800 3900 1100" "3600 1200 5000 1600" "2000 600 2800 900" "off"; do
  set -- $cfg
  if [ "$1" = "off" ]; then export SMCPP_SS_HALO=0; else export SMCPP_SS_HALO=1 SMCPP_HALO_LF=$1 SMCPP_HALO_DF=$2 SMCPP_HALO_LB=$3 SMCPP_HALO_DB=$4; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1 -- python $R/bench.py --no-cpu --no-ref-width --workload $W --steps 10 > $O/trace_$1.log 2>&1
  python - "$O/trace_$1" "$cfg" <<'PY'
import glob, csv, json, sys
d = json.loads([l for l in open(sys.argv[1] + ".log") if l.startswith('{"metric"')][-1])
s = d["split_ms"]
print("halo", sys.argv[2], "evals/s", round(d["value"], 1), "chains", round(s["chains_wall_ms"], 3), "passes", s["fwd_passes"])
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_chain_ss" in r["Kernel_Name"]]
print("   last launches (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-12:]])
PY
done
