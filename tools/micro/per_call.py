#!/usr/bin/env python3
"""Latency of the per-read / per-pair API calls the module swap of INTEGRATION.md section 1 makes, one object per call.
usage: tools/micro/per_call.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from atropos_amd import synth                           # noqa: E402
from atropos_amd.align import Aligner, InsertAligner, MultiAligner, PairAligner, compare_prefixes   # noqa: E402
from atropos_amd.util import reverse_complement          # noqa: E402
from atropos_amd.adapters import Adapter, BACK           # noqa: E402


def timed(fn, reps=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


w = synth.workload("C3", 0, 64)
r1 = [bytes(x).decode() for x in w["reads1"].cpu().numpy()]
r2 = [bytes(x).decode() for x in w["reads2"].cpu().numpy()]
c2 = [bytes(x).decode() for x in synth.workload("C2", 0, 64)["reads"].cpu().numpy()]
al = Aligner(synth.TRUSEQ_34, 0.1, 14, False, False, 3, 1)
ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
pa = PairAligner(0.2, 15, revcomp_ref=True)
ad = Adapter(synth.TRUSEQ_34, BACK, 0.1, 3)
rc2 = reverse_complement(r2[2])
out = {"Aligner.locate": timed(lambda: al.locate(c2[3])),
       "compare_prefixes": timed(lambda: compare_prefixes(synth.TRUSEQ_34, c2[5])),
       "InsertAligner.match_insert": timed(lambda: ia.match_insert(r1[2], r2[2])),
       "PairAligner.locate": timed(lambda: pa.locate(r2[4], r1[4])),
       "Adapter.match_to": timed(lambda: ad.match_to(c2[7]))}
out["Aligner(150-base ref).locate, built per call (MergeOverlapping)"] = timed(lambda: Aligner(rc2, 0.2, 15).locate(r1[2]), 100)
out["Aligner(34-base ref) built per call + locate"] = timed(lambda: Aligner(synth.TRUSEQ_34, 0.1, 14).locate(c2[3]), 100)
ma = MultiAligner(0.2, 9, 1)
out["MultiAligner.locate (2 x 150 bp, flags 9)"] = timed(lambda: ma.locate(rc2, r1[2]), 50)
print(json.dumps(out))
