#!/bin/bash
# One gpurun call of round 2: GPU parity tests, the four bench configs, rocprofv3 passes.
# usage (on the GPU box, via gpurun): tools/gpu_round2.sh <tag> [what ...]   what: tests bench prof_C2 prof_C3 prof_C4 prof_C5
set -u
TAG=${1:-r2a}
shift
WHAT=${*:-tests bench}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for W in $WHAT; do
  case $W in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
      tail -5 gpurun_out/${TAG}_pytest_gpu.log ;;
    bench)
      for C in C2 C4 C3 C5; do
        timeout 900 python bench.py --config $C > gpurun_out/${TAG}_bench_${C}.json 2> gpurun_out/${TAG}_bench_${C}.err
        echo "bench $C rc=$?"; tail -c 600 gpurun_out/${TAG}_bench_${C}.json; tail -3 gpurun_out/${TAG}_bench_${C}.err
      done ;;
    prof_*)
      C=${W#prof_}
      timeout 1500 bash tools/profile_r.sh ${TAG}_${C} $C > gpurun_out/${TAG}_prof_${C}.log 2>&1
      tail -30 gpurun_out/${TAG}_prof_${C}.log ;;
  esac
done
