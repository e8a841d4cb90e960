#!/usr/bin/env python3
"""Stage timings of the paired-end device pipeline (insert aligner) on one MI355X: N pairs of
2 x 150 bp (workload C3) as two FASTQ texts generated in HBM -> two trimmed FASTQ texts in HBM.
usage: tools/bench_fastq_pe.py [npairs] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import _lib, synth                    # noqa: E402
from atropos_amd.fastq import FastqBatch               # noqa: E402
from atropos_amd.trim import pipeline_from_args        # noqa: E402


def device_fastq(reads, tag):
    nreads, read_len = reads.shape
    dev = reads.device
    idx = torch.arange(nreads, device=dev, dtype=torch.int64)
    width = 2 + 9 + 2 + 1 + read_len + 1 + 2 + read_len + 1
    rec = torch.empty((nreads, width), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("p")
    for d in range(9):
        rec[:, 2 + d] = ((idx // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
    rec[:, 11] = ord("/")
    rec[:, 12] = ord(tag)
    rec[:, 13] = 10
    rec[:, 14:14 + read_len] = reads
    o = 14 + read_len
    rec[:, o] = 10
    rec[:, o + 1] = ord("+")
    rec[:, o + 2] = 10
    g = torch.Generator(device=dev)
    g.manual_seed(7 + ord(tag))
    slope = torch.rand((nreads, 1), device=dev, generator=g) * 0.2
    q = 38 - (torch.arange(read_len, device=dev)[None, :] * slope).to(torch.int64) \
        + torch.randint(-3, 4, (nreads, read_len), device=dev, generator=g)
    rec[:, o + 3:o + 3 + read_len] = (q.clamp_(2, 40) + 33).to(torch.uint8)
    rec[:, width - 1] = 10
    nbytes = nreads * width
    data = torch.zeros(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=dev)
    data[:nbytes] = rec.view(-1)
    return data, nbytes


def timed(fn, steps):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    config = sys.argv[3] if len(sys.argv) > 3 else "C3"        # optional 4th argument "merge"
    be = _lib.get_backend()
    w = synth.workload(config, 0, npairs, device="cuda")
    d1, n1 = device_fastq(w["reads1"], "1")
    d2, n2 = device_fastq(w["reads2"], "2")
    del w
    args = "--aligner insert -a %s -A %s -q 20 -m 30" % (synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    if config == "C5":
        # BASELINE C5: error correction + read wildcards.  No quality trimming in front of it: reads of
        # unequal length make the reference's correct_errors raise IndexError / ValueError on some
        # pairs (and this pipeline raises the same exception for the same pair).
        args = "--aligner insert -a %s -A %s --correct-mismatches liberal --match-read-wildcards -m 30" % (
            synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    merge = len(sys.argv) > 4 and sys.argv[4] == "merge"
    if merge:                                   # MergeOverlapping as the last stage (the merged text counts as output)
        args += " -R --merge-min-overlap 0.5"
    pipe = pipeline_from_args(args)
    stages = {}
    stages["index_ms"], (b1, b2) = timed(lambda: (FastqBatch.from_device(d1, n1, True, be)[0],
                                                  FastqBatch.from_device(d2, n2, True, be)[0]), steps)
    stages["run_ms"], res = timed(lambda: pipe.run(b1, b2), steps)
    stages["emit_ms"], outs = timed(lambda: tuple(
        be.fastq_emit(r.batch.data, r.batch.records, r.begin, r.end, None, None, r.dest, _lib.DEST_KEEP)
        for r in (res.read1, res.read2)), steps)
    total_ms = sum(stages.values())
    out_bytes = int(outs[0].numel() + outs[1].numel()) + (int(res.merged.numel()) if res.merged is not None else 0)
    print(json.dumps({
        "workload": "%s: 2 FASTQ texts in HBM -> 2 trimmed FASTQ texts in HBM, %d pairs x 2x%d bp, atropos trim %s"
                    % (config, npairs, (n1 // npairs - 18) // 2, args.replace(synth.PE_ADAPTER1, "A1").replace(synth.PE_ADAPTER2, "A2")),
        "corrected_pairs": pipe.corrected_pairs // (steps + 1) if pipe.corrected_pairs else 0,
        "npairs": npairs, "input_bytes": n1 + n2, "output_bytes": out_bytes, "counts": res.counts(),
        "stages": {k: round(v, 3) for k, v in stages.items()}, "total_ms": round(total_ms, 3),
        "pairs_per_s": npairs / total_ms * 1e3, "reads_per_s": 2 * npairs / total_ms * 1e3,
        "text_GBps": (n1 + n2 + out_bytes) / total_ms / 1e6}))


if __name__ == "__main__":
    main()
