#!/usr/bin/env python3
"""200 calls of atr_locate_batch on a resident 1000-read C2 batch (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from atropos_amd import synth
from atropos_amd.align import Aligner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w = synth.workload("C2", 0, n, device="cuda")
al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
b = al.pack(w["reads"])
for _ in range(200):
    al.locate_batch(b)
torch.cuda.synchronize()
