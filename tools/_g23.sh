timeout 600 python -m pytest tests/test_gpu_locate.py -x -q -k pair 2>&1 | tail -2
for i in 1 2; do python tools/bench_pairs.py 500000 5 C5 15 2>&1 | grep workload | cut -c1-220; done
python tools/bench_pairs.py 2000000 5 C3 15 2>&1 | grep workload | cut -c1-220
