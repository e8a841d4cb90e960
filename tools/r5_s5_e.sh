# session-5 lease E: window launches on a high-priority side stream, A/B (C2, C4) + timelines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5e
O=gpurun_out/s5e
B="python bench.py --no-cpu-baseline --no-secondary --no-live-counters --steps 20 --warmup 3"
for rep in 1 2 3; do
for V in "ATR_WINDOW_PRIORITY=0" "ATR_WINDOW_PRIORITY=1"; do
  for C in C2 C4; do
    echo -n "$V $C: " >> $O/ab.txt
    env $V timeout 300 $B --config $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'] / 1e9, 'G/s', d['ms_per_step'], 'ms')
" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
timeout 300 bash tools/kernel_timeline_cmd.sh atr_piece_spec $B --config C2 --steps 5 > $O/timeline_c2.txt 2>&1
cat $O/timeline_c2.txt
timeout 300 bash tools/kernel_timeline_cmd.sh linked_filter_kernel $B --config C4 --steps 5 > $O/timeline_c4.txt 2>&1
cat $O/timeline_c4.txt
timeout 600 python -m pytest tests/test_gpu_callers.py tests/test_gpu_locate.py -m gpu -x -q 2>&1 | tail -2
