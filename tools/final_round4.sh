#!/bin/bash
# Round-4 evidence in one call on the GPU box: the full -m gpu tier, the default bench line (all four configs, live
# counters, CPU baselines), the rocprofv3 summaries of C2 / C3 / C4 / C5 on the same lease, the smoke test, the ragged /
# read-length / two-stream tables.   bash tools/final_round4.sh   (outputs under gpurun_out/r4f/)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4f
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4f/round4_pytest_gpu.log 2>&1; tail -3 gpurun_out/r4f/round4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4f/round4_smoke.log 2>&1; tail -1 gpurun_out/r4f/round4_smoke.log
timeout 900 python bench.py > gpurun_out/r4f/round4_bench_default_all_configs.json 2> gpurun_out/r4f/bench_default.err; tail -c 600 gpurun_out/r4f/round4_bench_default_all_configs.json
for C in C2 C3 C4 C5; do
  bash tools/profile_r.sh r4_$C $C > gpurun_out/r4f/profile_$C.log 2>&1
  cp gpurun_out/prof_r4_$C/summary.txt gpurun_out/r4f/round4_$(echo $C | tr A-Z a-z)_rocprofv3_summary.txt
done
timeout 300 python tools/bench_lengths.py 5000000 > gpurun_out/r4f/round4_read_lengths.jsonl 2>/dev/null
timeout 300 python tools/bench_lengths.py 5000000 ragged > gpurun_out/r4f/round4_ragged_batches.jsonl 2>/dev/null
timeout 300 python tools/micro/two_streams.py > gpurun_out/r4f/round4_two_streams.jsonl 2>/dev/null
timeout 300 python bench.py --gpus 2 --oversubscribe --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-live-counters > gpurun_out/r4f/round4_bench_self_launch_2ranks_1gpu.json 2>/dev/null
ls gpurun_out/r4f
