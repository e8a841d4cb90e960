# session-5 lease B: scan-free offsets + band launch on the caller's stream, A/B on one lease (C2, C4), parity of the paths touched
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5b
O=gpurun_out/s5b
timeout 900 python -m pytest tests/test_gpu_locate.py tests/test_gpu_jit.py tests/test_gpu_callers.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-secondary --no-live-counters --steps 20 --warmup 3"
for rep in 1 2; do
for V in "ATR_FUSED_SCAN=0 ATR_BAND_MAIN=0" "ATR_FUSED_SCAN=1 ATR_BAND_MAIN=0" "ATR_FUSED_SCAN=0 ATR_BAND_MAIN=1" "ATR_FUSED_SCAN=1 ATR_BAND_MAIN=1"; do
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
