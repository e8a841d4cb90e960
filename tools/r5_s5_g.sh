# session-5 lease G: the per-pair aligner with more hardware queues (GPU_MAX_HW_QUEUES) -- does every class launch get a queue?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5g
O=gpurun_out/s5g
export ATR_PAIRS_PRIORITY=0
for rep in 1 2; do
for Q in 4 12; do
  for a in "2000000 5 C3 15" "500000 5 C5 15" "2000000 5 C3 9"; do
    echo -n "GPU_MAX_HW_QUEUES=$Q $a: " >> $O/ab.txt
    GPU_MAX_HW_QUEUES=$Q timeout 300 python tools/bench_pairs.py $a 2>/dev/null | grep workload | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], 'ms', d['pairs_per_s'] / 1e6, 'M pairs/s')
" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
echo "== GPU_MAX_HW_QUEUES=12, 2 M pairs 2 x 150, flags 15" >> $O/timeline.txt
GPU_MAX_HW_QUEUES=12 timeout 300 bash tools/kernel_timeline_cmd.sh pairs_myers_kernel python tools/bench_pairs.py 2000000 3 C3 15 >> $O/timeline.txt 2>&1
echo "== GPU_MAX_HW_QUEUES=12, 500 k pairs 2 x 250, flags 15" >> $O/timeline.txt
GPU_MAX_HW_QUEUES=12 timeout 300 bash tools/kernel_timeline_cmd.sh pairs_myers_kernel python tools/bench_pairs.py 500000 3 C5 15 >> $O/timeline.txt 2>&1
cat $O/timeline.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-live-counters --steps 20 --warmup 3"
for Q in 4 12; do for C in C2 C4; do echo -n "GPU_MAX_HW_QUEUES=$Q $C: "; GPU_MAX_HW_QUEUES=$Q timeout 300 $B --config $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'] / 1e9, 'G/s', d['ms_per_step'], 'ms')
"; done; done | tee $O/ab_bench.txt
