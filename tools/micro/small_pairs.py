#!/usr/bin/env python3
"""Per-call time of the pair-wise entry points on short batches: InsertAligner.match_insert_batch and
PairAligner.locate_batch (MergeOverlapping's aligner; wave / fast / full kernel families, parity asserted) on C3 and C5
pairs.  usage: tools/micro/small_pairs.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from atropos_amd import _lib, synth                    # noqa: E402
from atropos_amd.align import InsertAligner, PairAligner   # noqa: E402


def timed(fn, reps=100):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


be = _lib.get_backend()
for config in ("C3", "C5"):
    w = synth.workload(config, 0, 131072, device="cuda")
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    pa = PairAligner(0.2, 15, revcomp_ref=True)
    for n in (1, 1000, 4096, 16384, 32768, 65536, 131072):
        r1, r2 = w["reads1"][:n].contiguous(), w["reads2"][:n].contiguous()
        b1, b2 = ia.pack(r1), ia.pack(r2, check=True)
        rb, qb = pa._pack(r2, _lib.TABLE_DNA15, be, True), pa._pack(r1, _lib.TABLE_DNA15, be, True)
        row = {"config": config, "n": n, "insert_us": round(timed(lambda: ia.match_insert_batch(b1, b2)), 1)}
        ref = pa.locate_batch(rb, qb, path="full").records
        for path in ("wave", "fast", "full"):
            assert torch.equal(pa.locate_batch(rb, qb, path=path).records, ref), (config, n, path)
            row["pairs_%s_us" % path] = round(timed(lambda: pa.locate_batch(rb, qb, path=path), 20), 1)
        print(json.dumps(row), flush=True)
