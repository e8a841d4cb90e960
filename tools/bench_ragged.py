#!/usr/bin/env python3
"""The C2 batch as quality-trimmed data looks: the same 10 M reads cut to lengths 100 .. 150 (RAGGED path of
atr_locate_batch: no row-count bins, per-lane last columns).  usage: tools/bench_ragged.py [nreads] [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                          # noqa: E402
from atropos_amd.align import Aligner                  # noqa: E402
from atropos_amd.batch import ReadBatch                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = synth.workload("C2", 0, n, device="cuda")
al = Aligner(w["adapter"], w["max_error_rate"], flags=14, min_overlap=w["min_overlap"])
al.indel_cost = w["indel_cost"]
g = torch.Generator(device="cuda").manual_seed(5)
lens = torch.randint(100, 151, (n,), generator=g, device="cuda", dtype=torch.int32)
rb = ReadBatch.from_ascii(w["reads"], lens, 150, al.table_kind, al._table)
eq = al.pack(w["reads"], layout="auto")
out = {}
for name, batch in (("equal_150", eq), ("ragged_100_150", rb)):
    for _ in range(2):
        res = al.locate_batch(batch)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        res = al.locate_batch(batch)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    out[name] = {"ms": ms, "reads_per_s": n / ms * 1e3, "found": float(res.found().float().mean().item())}
print(json.dumps(out))
