# session-5 lease D: one window launch for a linked set -- parity (callers tier incl. the C4 shard against the oracle, linked fuzz), A/B, timeline
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5d
O=gpurun_out/s5d
timeout 1200 python -m pytest tests/test_gpu_callers.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-secondary --no-live-counters --steps 20 --warmup 3"
for rep in 1 2 3; do
for V in "ATR_ONE_WINDOW=0" "ATR_ONE_WINDOW=1"; do
    echo -n "$V C4: " >> $O/ab.txt
    env $V timeout 300 $B --config C4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'] / 1e9, 'G/s', d['ms_per_step'], 'ms')
" >> $O/ab.txt
done
done
cat $O/ab.txt
timeout 300 bash tools/kernel_timeline_cmd.sh linked_filter_kernel $B --config C4 --steps 5 > $O/timeline_c4.txt 2>&1
cat $O/timeline_c4.txt
timeout 900 python tools/gpu_fuzz.py 9200 1 > $O/gpu_fuzz.log 2>&1; tail -2 $O/gpu_fuzz.log
