#!/bin/bash
# Round profile of the default bench on the GPU box (run through gpurun):
#   bash tools/profile_round.sh <tag>        e.g.  bash tools/profile_round.sh r01_e
# Produces under gpurun_out/<tag>/: the default bench line, the rocprofv3 kernel-trace statistics of the same command
# and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only (never combined with other traces).
# Copy the summaries into profiles/ with tools/summarize_pmc.py afterwards.
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python $R/bench.py > $O/bench_default.log 2>&1 </dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu --no-ref-width > $O/stats.log 2>&1 </dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --no-cpu --no-ref-width --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1 </dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --no-cpu --no-ref-width --steps 2 --warmup 1 > $O/pmc_write.log 2>&1 </dev/null
tail -1 $O/bench_default.log | cut -c1-400
ls $O/*/*/ | head -20
