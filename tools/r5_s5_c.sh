# session-5 lease C: pass B's window out of an LDS stash -- parity (specialised == generic == oracle, equal-length and ragged,
# random aligners, the C2 batch at size), A/B on one lease, FETCH_SIZE of the pre-pass
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s5c
O=gpurun_out/s5c
timeout 1200 python -m pytest tests/test_gpu_jit.py tests/test_gpu_locate.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for rep in 1 2 3; do
  timeout 600 python tools/jit/ab.py "stash:ATR_JIT=1" "gather:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_STASH=0" "generic:ATR_JIT=0" >> $O/ab.txt 2>&1
done
timeout 600 python tools/jit/ab.py --ragged "stash_ragged:ATR_JIT=1" "gather_ragged:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_STASH=0" >> $O/ab.txt 2>&1
timeout 600 python tools/jit/ab.py --e 0.12 "stash_k4:ATR_JIT=1" "gather_k4:ATR_JIT=1,ATR_SPEC_FLAGS=-DATR_PIECE_STASH=0" >> $O/ab.txt 2>&1
cat $O/ab.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ATR_JIT=1 timeout 300 bash tools/pmc_cmd.sh $C python tools/jit/ab.py --child --rounds 1 --steps 2 >> $O/fetch_stash.txt 2>&1
done
cat $O/fetch_stash.txt
timeout 900 python tools/gpu_fuzz.py 9100 2 > $O/gpu_fuzz.log 2>&1; tail -3 $O/gpu_fuzz.log
