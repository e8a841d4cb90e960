#!/usr/bin/env python3
"""C2's reads as ASCII through the fused entry (atr_locate_ascii_planes_batch) and through pack + locate, a few calls each
(for tools/kernel_times_cmd.sh).  usage: bench_fused_ascii.py [reads] [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                       # noqa: E402
from atropos_amd.align import Aligner               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = synth.workload("C2", 0, n, device="cuda")
al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
reads = w["reads"]
be = al._backend
res, left = al.locate_ascii(reads)
planes = left.packed
for _ in range(2):
    be.locate_ascii_planes_batch(al._handle, reads, None, 150, planes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    rec, _ = be.locate_ascii_planes_batch(al._handle, reads, None, 150, planes)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(calls):
    rec2 = al.locate_batch(al.pack(reads, layout="plane64")).records
torch.cuda.synchronize()
t2 = time.perf_counter()
batch = al.pack(reads, layout="plane64")
for _ in range(calls):
    rec3 = al.locate_batch(batch).records
torch.cuda.synchronize()
t3 = time.perf_counter()
print("reads %d: fused %.3f ms (%.2f G/s), pack + locate %.3f ms (%.2f G/s), locate alone %.3f ms; equal %s" % (
    n, (t1 - t0) / calls * 1e3, n * calls / (t1 - t0) / 1e9, (t2 - t1) / calls * 1e3, n * calls / (t2 - t1) / 1e9,
    (t3 - t2) / calls * 1e3, bool(torch.equal(rec, rec2))))
