# The whole round's evidence on ONE box (run through gpurun):  bash tools/round_all.sh <tag>
# tools/final_round.sh (default bench with the CPU leg, kernel statistics, PMC passes, every workload's bench line, probes) + the device
# data shaping, the binned posterior decode (scan steps against eigensystems), the poison probe on the round's new paths, the whole
# GPU test-suite and the smoke entry.  python tools/collect_profiles.py <tag> afterwards (in the build container).
TAG=${1:-r06_c}
cd $GRAFT_REPO_ROOT
bash tools/final_round.sh $TAG
O=gpurun_out/$TAG
python bench.py --workload shaping > $O/bench_shaping.log 2>&1
PROBE_ROWS=100000 timeout 900 python tools/unbinned_probe.py > $O/unbinned_probe.log 2>&1; grep -a "differs" $O/unbinned_probe.log | cut -c1-200
timeout 900 python tools/gamma_scan_probe.py > $O/gamma_scan_probe.log 2>&1; tail -8 $O/gamma_scan_probe.log
timeout 1500 python tools/poison_probe.py cut:G1_M16_n4 biggamma:params_M256_n50 gamma:G4_M64_n20_2Mbp big:params_M64_n20 > $O/poison_probe.log 2>&1; tail -5 $O/poison_probe.log
bash tools/c4_watch.sh $TAG > /dev/null 2>&1; cut -c1-220 $O/c4_watch.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_suite.log; cat $O/gpu_suite.log
