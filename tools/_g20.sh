timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/kernel_times_cmd.sh python bench.py --config C4 --no-cpu-baseline --no-secondary --no-live-counters --steps 3 --warmup 1 2>&1 | grep -E "pack_kernel" | cut -c1-160
