set -x
mkdir -p gpurun_out/s1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s1/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s1/gputests.log
tail -3 gpurun_out/s1/gputests.log
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 2000000 3 C3 15 > gpurun_out/s1/pairs_c3_15.txt 2>&1
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 500000 3 C5 15 > gpurun_out/s1/pairs_c5_15.txt 2>&1
python tools/bench_pairs.py 2000000 5 C3 15 > gpurun_out/s1/pairs_plain.txt 2>&1
python tools/bench_pairs.py 500000 5 C5 15 >> gpurun_out/s1/pairs_plain.txt 2>&1
python tools/bench_pairs.py 2000000 5 C3 9 >> gpurun_out/s1/pairs_plain.txt 2>&1
cat gpurun_out/s1/pairs_*.txt
