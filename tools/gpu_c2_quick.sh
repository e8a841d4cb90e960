#!/bin/bash
# C2 on the GPU box: the piece-pipeline tests, the bench line, kernel times and instruction counters in one call.
#   bash tools/gpu_c2_quick.sh <tag>
TAG=${1:-x}
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_locate.py -x -q -k "piece or c2_sample" > gpurun_out/r4/pytest_$TAG.log 2>&1; tail -3 gpurun_out/r4/pytest_$TAG.log
timeout 600 python bench.py --config C2 --steps 20 --warmup 5 --no-live-counters --no-secondary --no-cpu-baseline > gpurun_out/r4/bench_$TAG.json 2>gpurun_out/r4/bench_$TAG.err
python -c "
import json
d=json.loads(open('gpurun_out/r4/bench_$TAG.json').read().strip().splitlines()[-1])
print('C2 %.2f G reads/s, %.4f ms/step, kernel_ms %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms']))"
CMD="python bench.py --config C2 --no-cpu-baseline --no-secondary --no-live-counters --steps 5 --warmup 2"
bash tools/kernel_times_cmd.sh $CMD 2>&1 | grep "atr::" | grep -v pack_kernel > gpurun_out/r4/kernels_$TAG.txt; cat gpurun_out/r4/kernels_$TAG.txt
bash tools/pmc_cmd.sh "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" $CMD 2>&1 | grep "piece_filter\|piece_wide" > gpurun_out/r4/pmc_$TAG.txt
bash tools/pmc_cmd.sh "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES" $CMD 2>&1 | grep "piece_filter\|piece_wide" >> gpurun_out/r4/pmc_$TAG.txt
cat gpurun_out/r4/pmc_$TAG.txt
