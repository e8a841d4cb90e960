#!/usr/bin/env python3
"""Stage timings of the device-resident FASTQ trimming pipeline (atropos_amd.trim) on one
MI355X: N reads x 150 bp (workload C2's reads) as FASTQ text generated in HBM, then
index -> quality trim -> pack -> locate -> trim -> N-ends -> filters -> formatted output.
Prints one JSON line.  usage: tools/bench_fastq.py [nreads] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import _lib, synth                    # noqa: E402
from atropos_amd.fastq import FastqBatch               # noqa: E402
from atropos_amd.trim import pipeline_from_args        # noqa: E402


def device_fastq(nreads, read_len=150):
    """FASTQ text built on the GPU: '@r' + 9 digits, the C2 read, '+', qualities decaying to
    the 3' end.  Fixed record width, but nothing downstream knows that."""
    dev = torch.device("cuda")
    reads = synth.workload("C2", 0, nreads, device=dev)["reads"]                    # uint8 [n, 150]
    idx = torch.arange(nreads, device=dev, dtype=torch.int64)
    width = 2 + 9 + 1 + read_len + 1 + 2 + read_len + 1
    rec = torch.empty((nreads, width), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    for d in range(9):
        rec[:, 2 + d] = ((idx // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
    rec[:, 11] = 10
    rec[:, 12:12 + read_len] = reads
    o = 12 + read_len
    rec[:, o] = 10
    rec[:, o + 1] = ord("+")
    rec[:, o + 2] = 10
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    slope = torch.rand((nreads, 1), device=dev, generator=g) * 0.25
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
    nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    be = _lib.get_backend()
    data, nbytes = device_fastq(nreads)
    args = "-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -q 20 --trim-n -m 20"
    pipe = pipeline_from_args(args)
    stages = {}
    stages["index_ms"], (batch, _) = timed(lambda: FastqBatch.from_device(data, nbytes, True, be), steps)
    stages["run_ms"], res = timed(lambda: pipe.run(batch), steps)
    stages["emit_ms"], out = timed(lambda: be.fastq_emit(batch.data, batch.records, res.begin, res.end, None, None,
                                                         res.dest, _lib.DEST_KEEP), steps)
    # finer split of run(): the individual library calls
    n = len(batch)
    begin = torch.zeros((n,), dtype=torch.int32, device=data.device)
    end = batch.seq_lens.clone()
    stages["quality_trim_ms"], _ = timed(lambda: be.quality_trim_batch(batch.data, batch.records, begin.clone(),
                                                                       end.clone(), 0, 20, 33, False), steps)
    table = be.translate_table(_lib.TABLE_DNA15)
    stages["pack_ms"], (packed, lens) = timed(lambda: be.pack_records(batch.data, batch.records, begin, end, 150, table),
                                              steps)
    stages["nend_ms"], _ = timed(lambda: be.nend_trim_batch(batch.data, batch.records, begin.clone(), end.clone()), steps)
    stages["filter_ms"], _ = timed(lambda: be.read_filter_batch(batch.data, batch.records, begin, end, None, None, None,
                                                                20, -1, -1.0, False, False), steps)
    total_ms = stages["index_ms"] + stages["run_ms"] + stages["emit_ms"]
    print(json.dumps({
        "workload": "FASTQ text in HBM -> trimmed FASTQ text in HBM, %d x 150 bp, atropos trim %s" % (nreads, args),
        "nreads": nreads, "input_bytes": nbytes, "output_bytes": int(out.numel()),
        "counts": res.counts(), "stages": {k: round(v, 3) for k, v in stages.items()},
        "total_ms": round(total_ms, 3), "reads_per_s": nreads / total_ms * 1e3,
        "text_GBps": (nbytes + int(out.numel())) / total_ms / 1e6}))


if __name__ == "__main__":
    main()
