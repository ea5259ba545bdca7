mkdir -p gpurun_out/r04_n
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_n/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04_n/tests.log
tail -3 gpurun_out/r04_n/tests.log
bash tools/final_round.sh r04_n 2>&1 | grep -v warning | grep '^{"metric"\|^warm\|^world\|smoke' | cut -c1-260
