# session-5 lease A: GPU tier on HEAD, kernel timelines of C2 / C4 (gaps between the launches of one call), FETCH_SIZE of the
# specialised pre-pass with and without pass B's gathers (calibration of the traffic figure)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5a
O=gpurun_out/s5a
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
timeout 300 bash tools/kernel_timeline_cmd.sh atr_piece_spec python bench.py --config C2 --no-cpu-baseline --no-secondary --no-live-counters --steps 5 --warmup 2 > $O/timeline_c2.txt 2>&1
cat $O/timeline_c2.txt
timeout 300 bash tools/kernel_timeline_cmd.sh linked_filter_kernel python bench.py --config C4 --no-cpu-baseline --no-secondary --no-live-counters --steps 5 --warmup 2 > $O/timeline_c4.txt 2>&1
cat $O/timeline_c4.txt
for V in "" "-DATR_X_NOGATHER=1"; do
  echo "== ATR_SPEC_FLAGS='$V'" >> $O/fetch_calibration.txt
  for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
    ATR_JIT=1 ATR_SPEC_FLAGS="$V" timeout 300 bash tools/pmc_cmd.sh $C python tools/jit/ab.py --child --rounds 1 --steps 2 >> $O/fetch_calibration.txt 2>&1
  done
done
cat $O/fetch_calibration.txt
