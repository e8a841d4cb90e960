# Round-5 evidence on one lease: smoke, the default bench line (all four configs, live PMC traffic), then per config a
# kernel trace + PMC passes (tools/profile_r.sh) whose summary carries the traced run's own bench line.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5p
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5p/round5_smoke.log 2>&1
python bench.py > gpurun_out/r5p/round5_bench_default_all_configs.json 2> gpurun_out/r5p/bench_default.err
for C in ${CONFIGS:-C2 C3 C4 C5}; do
  bash tools/profile_r.sh r5_$C $C > gpurun_out/r5p/profile_$C.log 2>&1
  cp gpurun_out/prof_r5_$C/summary.txt gpurun_out/r5p/round5_$(echo $C | tr A-Z a-z)_rocprofv3_summary.txt
done
tail -2 gpurun_out/r5p/round5_smoke.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p/round5_bench_default_all_configs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("configs_summary"))
PY
head -4 gpurun_out/r5p/round5_c*_rocprofv3_summary.txt
python tools/bench_small.py gpurun_out/r5p/round5_small_batches.json > gpurun_out/r5p/bench_small.log 2>&1
for a in "2000000 5 C3 15" "500000 5 C5 15" "2000000 5 C3 9"; do python tools/bench_pairs.py $a 2>/dev/null | grep workload; done > gpurun_out/r5p/round5_bench_pairs.jsonl
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 2000000 3 C3 15 > gpurun_out/r5p/round5_pairs_kernels_c3.txt 2>&1
bash tools/kernel_times_cmd.sh python tools/bench_pairs.py 500000 3 C5 15 > gpurun_out/r5p/round5_pairs_kernels_c5.txt 2>&1
cat gpurun_out/r5p/round5_bench_pairs.jsonl | cut -c1-200
