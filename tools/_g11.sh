mkdir -p gpurun_out/s11
timeout 1200 python -m pytest tests/test_gpu_locate.py -x -q -k "pair" > gpurun_out/s11/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s11/gputests.log
tail -3 gpurun_out/s11/gputests.log
python tools/bench_pairs.py 2000000 5 C3 15 2>&1 | grep workload | cut -c1-230
python tools/bench_pairs.py 500000 5 C5 15 2>&1 | grep workload | cut -c1-230
python tools/bench_pairs.py 2000000 5 C3 9 2>&1 | grep workload | cut -c1-230
