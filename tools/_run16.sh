cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_f; mkdir -p $O
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-160
python bench.py --workload c5 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log | cut -c1-200
python bench.py --workload posterior > $O/bench_posterior.log 2>&1; tail -1 $O/bench_posterior.log | cut -c1-200
