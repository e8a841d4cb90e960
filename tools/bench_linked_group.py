#!/usr/bin/env python3
"""C4's reads through the grouped linked pipeline (atr_linked_group_pack / atr_linked_group_match): a few calls of each for
the kernel traces (tools/kernel_times_cmd.sh, tools/kernel_timeline_cmd.sh).  usage: bench_linked_group.py [reads] [packs] [matches]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                                                    # noqa: E402
from atropos_amd.adapters import LinkedAdapter, LinkedSet, upper_ascii           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
packs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
matches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
chunks = [synth.workload("C4", lo, min(2_500_000, n - lo), device="cuda") for lo in range(0, n, 2_500_000)]
w = chunks[0]
reads = upper_ascii(torch.cat([c["reads"] for c in chunks]))
del chunks
lset = LinkedSet([LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                                min_overlap=w["min_overlap"], indel_cost=w["indel_cost"]) for f, b in zip(w["fronts"], w["backs"])])
groups = lset.pack_groups(reads)
for _ in range(3):
    lset.match_groups(groups)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(packs):
    groups = lset.pack_groups(reads)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(matches):
    res = lset.match_groups(groups)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("reads %d: pack %.3f ms, match %.3f ms (%.2f G reads/s), groups %s" % (
    n, (t1 - t0) / max(1, packs) * 1e3, (t2 - t1) / max(1, matches) * 1e3, n * matches / (t2 - t1) / 1e9, groups.group_reads()))
