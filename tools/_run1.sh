mkdir -p gpurun_out/r04_m
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_m/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04_m/tests.log
tail -3 gpurun_out/r04_m/tests.log
bash tools/final_round.sh r04_m
