# session-5 lease I: tiles dealt dynamically inside a block (ATR_PIECE_DYNAMIC), A/B through the run-time compiler
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5i
O=gpurun_out/s5i
for rep in 1 2 3; do
  timeout 600 python tools/jit/ab.py "static:ATR_JIT=1" "dynamic:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_DYNAMIC=1" >> $O/ab.txt 2>&1
done
timeout 600 python tools/jit/ab.py --ragged "static_ragged:ATR_JIT=1" "dynamic_ragged:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_DYNAMIC=1" >> $O/ab.txt 2>&1
timeout 600 python tools/jit/ab.py --e 0.12 "static_k4:ATR_JIT=1" "dynamic_k4:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_DYNAMIC=1" >> $O/ab.txt 2>&1
cat $O/ab.txt
