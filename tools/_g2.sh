set -x
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_insert.py -x -q > gpurun_out/s2/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s2/gputests.log
tail -5 gpurun_out/s2/gputests.log
for i in 1 2; do
python bench.py --config C5 --no-cpu-baseline --no-secondary --steps 10 --warmup 2 > gpurun_out/s2/c5_fused_$i.json 2>gpurun_out/s2/c5_fused_$i.err
ATR_BENCH_C5_TWO_CALLS=1 python bench.py --config C5 --no-cpu-baseline --no-secondary --steps 10 --warmup 2 > gpurun_out/s2/c5_two_$i.json 2>gpurun_out/s2/c5_two_$i.err
done
bash tools/kernel_times_cmd.sh python bench.py --config C5 --no-cpu-baseline --no-secondary --no-live-counters --steps 10 --warmup 2 > gpurun_out/s2/c5_kernels.txt 2>&1
for f in gpurun_out/s2/c5_*.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
PY
done
cat gpurun_out/s2/c5_kernels.txt
