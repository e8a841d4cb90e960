#!/usr/bin/env python3
"""atr_insert_correct_batch alone on C5 pairs: ms per call for the three mismatch actions, with and without
qualities.  usage: tools/bench_correct.py [npairs]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import _lib, synth                    # noqa: E402
from atropos_amd.align import InsertAligner            # noqa: E402
from atropos_amd.modifiers import COMP_TABLE           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
be = _lib.get_backend()
w = synth.workload("C5", 0, n, device="cuda")
ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
b1, b2 = ia.pack(w["reads1"]), ia.pack(w["reads2"], check=True)
rec = ia.match_insert_batch(b1, b2).records
out = {"npairs": n, "with_insert_match": float((rec[:, 0, 1] >= 0).float().mean().item()),
       "with_errors": float(((rec[:, 0, 1] >= 0) & (rec[:, 0, 5] > 0)).float().mean().item())}
for name, action, quals in (("N", 0, False), ("N_quals", 0, True), ("conservative", 1, True), ("liberal", 2, True)):
    times = []
    for rep in range(4):
        s1, s2 = w["reads1"].clone(), w["reads2"].clone()
        q1 = w["quals1"].clone() if quals else None
        q2 = w["quals2"].clone() if quals else None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ch, nl = be.insert_correct_batch(rec, s1, q1, None, s2, q2, None, action, 1, COMP_TABLE, planes1=b1, planes2=b2)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    out[name] = {"ms": min(times[1:]), "changed_pairs": int((ch.sum(dim=1) > 0).sum().item()),
                 "changed_bases": int(ch.sum().item())}
print(json.dumps(out))
