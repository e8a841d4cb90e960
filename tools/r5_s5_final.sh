# session-5 final lease: the GPU tier on the final code, then the round's evidence (tools/profile_round5.sh)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5p
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5p/round5_pytest_gpu.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5p/round5_pytest_gpu.log
tail -3 gpurun_out/r5p/round5_pytest_gpu.log
bash tools/profile_round5.sh
