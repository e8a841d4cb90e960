# session-5 lease L: instruction schedulers for the run-time compiled pre-pass (hiprtc options through ATR_SPEC_FLAGS)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5l
O=gpurun_out/s5l
for rep in 1 2 3; do
  timeout 900 python tools/jit/ab.py "default:ATR_JIT=1" "ilp:ATR_JIT=1,ATR_SPEC_FLAGS=-mllvm -misched=gcn-iterative-ilp" "minreg:ATR_JIT=1,ATR_SPEC_FLAGS=-mllvm -misched=gcn-iterative-minreg" "maxocc:ATR_JIT=1,ATR_SPEC_FLAGS=-mllvm -misched=gcn-iterative-max-occupancy-experimental" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
