#!/usr/bin/env python3
"""Throughput of atr_locate_pairs_batch (the per-pair aligner behind MergeOverlapping) on
workload C3's read pairs: reference = reverse complement of read 2 (formed on the device),
query = read 1, flags SEMIGLOBAL, e = 0.2.  usage: tools/bench_pairs.py [npairs] [steps] [C3|C5] [flags] [full]
("full": the full-matrix sweep alone, atr_locate_pairs_full_batch)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import _lib, synth                    # noqa: E402
from atropos_amd.align import PairAligner              # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
config = sys.argv[3] if len(sys.argv) > 3 else "C3"          # C3: 2 x 150 bp, C5: 2 x 250 bp
w = synth.workload(config, 0, npairs, device="cuda")
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 15
full = len(sys.argv) > 5 and sys.argv[5] == "full"
pa = PairAligner(0.2, flags, revcomp_ref=True)
be = _lib.get_backend()
rb = pa._pack(w["reads2"], _lib.TABLE_DNA15, be, True)
qb = pa._pack(w["reads1"], _lib.TABLE_DNA15, be, True)
from atropos_amd.align import LocateResult               # noqa: E402


def run():
    if full:
        return LocateResult(be.locate_pairs_full_batch(rb.packed, rb.lens, rb.max_len, True, qb.packed, qb.lens, qb.max_len,
                                                       rb.nreads, 0.2, flags, 1, 1))
    return pa.locate_batch(rb, qb)


res = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    res = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
found = float(res.found().float().mean().item())
n = w["reads1"].shape[1]
print(json.dumps({"workload": config + " pairs, Aligner(rc(read2), 0.2, flags %d).locate(read1), %d x 2x%d bp%s" % (
                      flags, npairs, n, ", full sweep only" if full else ""),
                  "npairs": npairs, "ms_per_step": ms, "pairs_per_s": npairs / ms * 1e3, "found_fraction": found,
                  "cell_updates_per_s": npairs * n * n / ms * 1e3}))
