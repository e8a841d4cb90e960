#!/usr/bin/env python3
"""10 M x 150 bp reads against adapters of other lengths / flag sets than C2's (synth.single_end with that adapter): the
two-pass pipeline on bit planes where its envelope takes the aligner, the one-pass pipeline on tile64 beside it.
usage: bench_adapters.py [reads]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                       # noqa: E402
from atropos_amd.align import Aligner               # noqa: E402


def run(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def measure(n=10_000_000, only=None):
    out = {}
    cases = [("truseq34_back_e0.10", synth.TRUSEQ_34, 0.1, 14), ("pe64_back_e0.10", synth.PE_ADAPTER1, 0.1, 14),
             ("pe58_back_e0.10", synth.PE_ADAPTER2, 0.1, 14), ("pe64_back_e0.08", synth.PE_ADAPTER1, 0.08, 14),
             ("first48_back_e0.10", synth.PE_ADAPTER1[:48], 0.1, 14),
             ("truseq34_front_e0.10", synth.TRUSEQ_34, 0.1, 11), ("truseq34_anywhere_e0.10", synth.TRUSEQ_34, 0.1, 15),
             ("first24_front_e0.10", synth.TRUSEQ_34[:24], 0.1, 11), ("first24_anywhere_e0.10", synth.TRUSEQ_34[:24], 0.1, 15)]
    for name, adapter, e, flags in cases:
        if only and only not in name:
            continue
        reads = synth.single_end(0, n, 150, adapter, 0xA72050007, "cuda")
        al = Aligner(adapter, e, flags, False, False, 3, 1)
        res = {}
        batch = al.pack(reads, layout="auto")
        res["layout"] = batch.layout
        rec = al.locate_batch(batch).records
        ms = run(lambda: al.locate_batch(batch))
        res["reads_per_s"] = n / (ms * 1e-3)
        res["ms"] = ms
        tiles = al.pack(reads, layout="tile64") if batch.layout != "tile64" else batch
        ms1 = run(lambda: al.locate_batch(tiles, path="filtered"), reps=5)
        res["one_pass_tile64_reads_per_s"] = n / (ms1 * 1e-3)
        res["records_equal_one_pass"] = bool(torch.equal(rec, al.locate_batch(tiles, path="filtered").records))
        res["matched_fraction"] = float((rec[:, 1] >= 0).float().mean().item())
        out[name] = res
        del reads, batch, tiles, rec
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
