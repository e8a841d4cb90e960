#!/usr/bin/env python3
"""Timing of BASELINE config C4 on one GPU shard: 150 bp SE reads, four linked adapters
(anchored 5' + regular 3'), e = 0.12.  The whole device-resident pipeline is timed: per
adapter the 5' match over all reads, re-pack of read[front.rstop:], the 3' match, and the
selection of the (single) adapter whose 5' part matched.  Results go into BASELINE.md."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from atropos_amd import synth
    from atropos_amd.adapters import LinkedAdapter, linked_best_records, upper_ascii
    chunks = []
    for lo in range(0, args.reads, 2_000_000):
        chunks.append(synth.workload("C4", lo, min(2_000_000, args.reads - lo), device="cuda")["reads"])
    reads = upper_ascii(torch.cat(chunks))
    del chunks
    linked = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=0.12, min_overlap=3,
                            indel_cost=1) for f, b in zip(synth.LINKED_FRONTS, synth.LINKED_BACKS)]

    def step():
        which, f, b = linked_best_records(linked, reads)
        return f, b, (which == -2).to(torch.int32) * 2

    for _ in range(2):
        res = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    f, b, nm = res
    algo = 75 + 16 + 16
    print(json.dumps({"config": "C4 (one GPU shard)", "reads": args.reads, "ms_per_step": dt * 1e3,
                      "reads_per_s": args.reads / dt, "algorithmic_bytes_per_read": algo,
                      "achieved_GBs": algo * args.reads / dt / 1e9, "hbm_frac": algo * args.reads / dt / 1e9 / 8000.0,
                      "front_matched": float((f[:, 1] >= 0).float().mean().item()),
                      "back_matched": float((b[:, 1] >= 0).float().mean().item()),
                      "reads_with_two_fronts": int((nm > 1).sum().item())}))


if __name__ == "__main__":
    main()
