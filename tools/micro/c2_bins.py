#!/usr/bin/env python3
"""Which scatter bins do the unresolved reads of the C2 workload fall into?  Reads `binbase` out of the
library's workspace after one atr_locate_batch (layout: locate_fast.hpp, fast_carve) -- a diagnostic for
DESIGN.md section 8, not a test.   usage (GPU box): python tools/micro/c2_bins.py [nreads]"""
import sys
sys.path.insert(0, '.')
import torch
from atropos_amd import _lib, synth
be = _lib.HipBackend(0)
_lib.set_backend(be)
from atropos_amd.align import Aligner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
w = synth.workload("C2", synth.SEEDS["C2"], n, device="cuda")
al = Aligner(synth.TRUSEQ_34, 0.1, flags=14, min_overlap=3)
batch = al.pack(w["reads"])
res = al.locate_batch(batch)
torch.cuda.synchronize()
FAST_BLOCKS, NBINS = 8192, 256
base = be._work.data_ptr()
off = n * 4
off = ((n + 1) & ~1) * 4 + n * 8                    # win, order
off += ((n + 63) // 64) * 8                          # mask
off = ((base + off + 15) & ~15) - base               # counts (16-byte aligned)
off += (FAST_BLOCKS + FAST_BLOCKS // 64) * NBINS * 4
binbase = be._work[off:off + (NBINS + 1) * 4].view(torch.int32).cpu().tolist()
cnt = [binbase[i + 1] - binbase[i] for i in range(NBINS)]
found = int(res.found().sum())
print("reads", n, "found", found, "unresolved", binbase[NBINS])
print("band bins [0,96):", sum(cnt[:96]), " full-row window bins [96,192):", sum(cnt[96:192]), " row-count bins [192,256):", sum(cnt[192:]))
print("window bins by start/8:", {i - 96: c for i, c in enumerate(cnt) if 96 <= i < 192 and c})
print("row-count bins:", {i - 192: c for i, c in enumerate(cnt) if i >= 192 and c})
