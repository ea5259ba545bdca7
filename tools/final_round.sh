cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02_h > /dev/null 2>&1
O=gpurun_out/r02_h
for w in c2 c3 c4 c5 posterior; do python bench.py --workload $w > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log | cut -c1-200; done
python bench.py --gpus 2 > $O/bench_gpus2.log 2>&1; tail -1 $O/bench_gpus2.log | cut -c1-200
python bench.py --gpus 2 --workload c3 > $O/bench_c3_gpus2.log 2>&1; tail -1 $O/bench_c3_gpus2.log | cut -c1-200
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c5 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c5 --steps 5 > /dev/null 2>&1
tail -1 $GRAFT_REPO_ROOT/$O/bench_default.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c3 --steps 5 > /dev/null 2>&1
