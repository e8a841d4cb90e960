mkdir -p gpurun_out/s12
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s12/round5_pytest_gpu.log 2>&1; echo "tests rc=$?" >> gpurun_out/s12/round5_pytest_gpu.log
tail -3 gpurun_out/s12/round5_pytest_gpu.log
timeout 1500 python tools/gpu_fuzz.py 7000 12 > gpurun_out/s12/round5_gpu_fuzz.log 2>&1; echo "fuzz rc=$?" >> gpurun_out/s12/round5_gpu_fuzz.log
tail -4 gpurun_out/s12/round5_gpu_fuzz.log
