# session-5 lease M: prefetch placement / waves per SIMD of the specialised pre-pass with the LDS stash in place
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5m
O=gpurun_out/s5m
for rep in 1 2; do
  timeout 900 python tools/jit/ab.py "default:ATR_JIT=1" "late:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_PREFETCH_EARLY=0" "waves3:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_WAVES(NW)=3" "waves5:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_WAVES(NW)=5" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
