# Round-6 evidence on one lease: smoke, the default bench line (all four configs, live PMC traffic), then per config a
# kernel trace + PMC passes (tools/profile_r.sh) whose summary carries the traced run's own bench line.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6p
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6p/round6_smoke.log 2>&1
python bench.py > gpurun_out/r6p/round6_bench_default_all_configs.json 2> gpurun_out/r6p/bench_default.err
for C in ${CONFIGS:-C2 C3 C4 C5}; do
  bash tools/profile_r.sh r6_$C $C > gpurun_out/r6p/profile_$C.log 2>&1
  cp gpurun_out/prof_r6_$C/summary.txt gpurun_out/r6p/round6_$(echo $C | tr A-Z a-z)_rocprofv3_summary.txt
done
tail -2 gpurun_out/r6p/round6_smoke.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6p/round6_bench_default_all_configs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("configs_summary"))
PY
head -4 gpurun_out/r6p/round6_c*_rocprofv3_summary.txt
python tools/bench_small.py gpurun_out/r6p/round6_small_batches.json > gpurun_out/r6p/bench_small.log 2>&1
for a in "2000000 5 C3 15" "500000 5 C5 15" "2000000 5 C3 9"; do python tools/bench_pairs.py $a 2>/dev/null | grep workload; done > gpurun_out/r6p/round6_bench_pairs.jsonl
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 2000000 3 C3 15 > gpurun_out/r6p/round6_pairs_kernels_c3.txt 2>&1
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 500000 3 C5 15 > gpurun_out/r6p/round6_pairs_kernels_c5.txt 2>&1
cat gpurun_out/r6p/round6_bench_pairs.jsonl | cut -c1-200
# round 6: other adapters / flag sets (two-pass envelope), the grouped linked pipeline's timeline, the fused ASCII entry
python tools/bench_adapters.py 10000000 > gpurun_out/r6p/round6_adapters.json 2> gpurun_out/r6p/adapters.err
bash tools/kernel_timeline_cmd.sh linked_front_ascii python tools/bench_linked_group.py 12500000 1 1 > gpurun_out/r6p/round6_c4_group_timeline.txt 2>&1
bash tools/kernel_times_cmd.sh python tools/bench_fused_ascii.py 10000000 5 > gpurun_out/r6p/round6_fused_ascii_kernels.txt 2>&1
python tools/bench_fused_ascii.py 10000000 10 >> gpurun_out/r6p/round6_fused_ascii_kernels.txt 2>&1
tail -3 gpurun_out/r6p/round6_fused_ascii_kernels.txt
# file -> file (secondary.file_to_file at three sizes) and C5 by pairs per launch
ATR_STAGING_POOL_GB=24 ATR_IO_THREADS=16 python tools/bench_file_to_file.py 16000000 3 1,8 128 > gpurun_out/r6p/round6_file_to_file.json 2> gpurun_out/r6p/f2f.err
for n in 1000000 2000000 8000000; do python bench.py --config C5 --reads $n --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-live-counters --no-other-configs 2>/dev/null | tail -1; done > gpurun_out/r6p/round6_c5_by_launch_size.jsonl
cut -c1-160 gpurun_out/r6p/round6_c5_by_launch_size.jsonl
