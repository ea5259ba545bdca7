mkdir -p gpurun_out/t12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ss.py -m gpu -x -q -k "hybrid or G18 or G7 or posterior" > gpurun_out/t12/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/t12/tests.log
tail -6 gpurun_out/t12/tests.log
