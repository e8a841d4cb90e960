# session-5 lease F: the per-pair aligner's band classes on two priority levels (two sets of hardware queues): A/B + timelines + parity
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5f
O=gpurun_out/s5f
for rep in 1 2; do
for V in "ATR_PAIRS_PRIORITY=0" "ATR_PAIRS_PRIORITY=1"; do
  for a in "2000000 5 C3 15" "500000 5 C5 15" "2000000 5 C3 9"; do
    echo -n "$V $a: " >> $O/ab.txt
    env $V timeout 300 python tools/bench_pairs.py $a 2>/dev/null | grep workload | cut -c1-220 >> $O/ab.txt
  done
done
done
cat $O/ab.txt
for V in 0 1; do
  echo "== ATR_PAIRS_PRIORITY=$V, 2 M pairs 2 x 150, flags 15" >> $O/timeline.txt
  ATR_PAIRS_PRIORITY=$V timeout 300 bash tools/kernel_timeline_cmd.sh pairs_myers_kernel python tools/bench_pairs.py 2000000 3 C3 15 >> $O/timeline.txt 2>&1
  echo "== ATR_PAIRS_PRIORITY=$V, 500 k pairs 2 x 250, flags 15" >> $O/timeline.txt
  ATR_PAIRS_PRIORITY=$V timeout 300 bash tools/kernel_timeline_cmd.sh pairs_myers_kernel python tools/bench_pairs.py 500000 3 C5 15 >> $O/timeline.txt 2>&1
done
cat $O/timeline.txt
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_insert.py tests/test_gpu_fastq.py -m gpu -x -q 2>&1 | tail -5
