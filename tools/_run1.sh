O=gpurun_out/r04_p; mkdir -p $O
python bench.py --workload posterior64 > $O/bench_posterior64.log 2>&1; grep '^{"metric"' $O/bench_posterior64.log | tail -1 | cut -c1-300
python bench.py --workload posterior > $O/bench_posterior.log 2>&1; grep '^{"metric"' $O/bench_posterior.log | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_posterior64 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload posterior64 --steps 5 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/$O/stats_posterior64/*/ | head -5
