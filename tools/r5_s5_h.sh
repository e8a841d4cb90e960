# session-5 lease H: the whole GPU tier on the final tree + a fuzz campaign of fresh seeds
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s5h/round5_pytest_gpu.log 2>&1; echo "tests rc=$?" >> gpurun_out/s5h/round5_pytest_gpu.log
tail -3 gpurun_out/s5h/round5_pytest_gpu.log
timeout 1200 python tools/gpu_fuzz.py 9300 6 > gpurun_out/s5h/round5_gpu_fuzz.log 2>&1; tail -3 gpurun_out/s5h/round5_gpu_fuzz.log
python bench.py > gpurun_out/s5h/bench_default.json 2> gpurun_out/s5h/bench_default.err; tail -c 600 gpurun_out/s5h/bench_default.json
