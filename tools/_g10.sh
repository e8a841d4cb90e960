mkdir -p gpurun_out/s10
timeout 900 python -m pytest tests/test_gpu_insert.py -x -q > gpurun_out/s10/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s10/gputests.log
tail -3 gpurun_out/s10/gputests.log
for i in 1 2 3; do
python bench.py --config C5 --no-cpu-baseline --no-secondary --no-live-counters --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 fused', d['value'], d['ms_per_step'])"
done
