#!/usr/bin/env python3
"""End-to-end (host memory -> GPU -> host memory) rate of the single-end FASTQ pipeline: the
PCIe-inclusive number DESIGN.md quotes next to the HBM-resident one.
usage: tools/bench_fastq_e2e.py [nreads] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                          # noqa: E402
from atropos_amd.trim import pipeline_from_args        # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_fastq import device_fastq                   # noqa: E402

from atropos_amd.fastq import FastqSink                # noqa: E402
FastqSink.set_writers(int(os.environ.get("ATR_SINK_WRITERS", "1")))     # threads per output buffer (fastq.py)
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
data, nbytes = device_fastq(nreads)
host = bytes(data[:nbytes].cpu().numpy().tobytes())    # the "file" in host memory
del data
pipe = pipeline_from_args("-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -q 20 --trim-n -m 20")
out = pipe.trim_bytes(host)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = pipe.trim_bytes(host)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
res = {"workload": "host FASTQ bytes -> trimmed host bytes (trim_bytes), %d x 150 bp" % nreads, "input_bytes": nbytes,
       "sink_writers": FastqSink.WRITERS,
       "output_bytes": len(out), "ms": ms, "reads_per_s": nreads / ms * 1e3, "host_GBps": (nbytes + len(out)) / ms / 1e6}
# file -> file through page-locked staging buffers with read-ahead / write-behind threads
src, dst = "/tmp/atr_e2e_in.fastq", "/tmp/atr_e2e_out.fastq"
with open(src, "wb") as fh:
    fh.write(host)


def file_to_file(dst, chunk, keep, parts=1):
    """steps runs of trim_file; the output file is removed before each run unless keep (then it is
    overwritten in place: cached pages are reused instead of allocated).  parts > 1: the output as part files."""
    names = [dst] if parts == 1 else ["%s.part%d" % (dst, i) for i in range(parts)]
    pipe.trim_file(src, dst, chunk_bytes=chunk, keep_output=keep, output_parts=parts)
    total, stages = 0.0, {}
    for _ in range(steps):
        for name in names:
            if not keep and os.path.exists(name):
                os.remove(name)
        t0 = time.perf_counter()
        counts = pipe.trim_file(src, dst, chunk_bytes=chunk, keep_output=keep, output_parts=parts)
        total += time.perf_counter() - t0
        for k, v in pipe.stage_seconds.items():
            stages[k] = stages.get(k, 0.0) + v
    ms = total / steps * 1e3
    if parts == 1:
        assert open(dst, "rb").read() == out
    else:
        assert sum(os.path.getsize(name) for name in names) == len(out)
    for name in names:
        os.remove(name)
    return {"ms": ms, "output_parts": parts, "reads_per_s": nreads / ms * 1e3, "host_GBps": (nbytes + len(out)) / ms / 1e6, "chunk_bytes": chunk,
            "counts": counts, "wait_ms_per_run": {k: round(v / steps * 1e3, 2) for k, v in stages.items()}}


res["trim_file"] = file_to_file(dst, 256 << 20, False)
res["trim_file_128M_chunks"] = file_to_file(dst, 128 << 20, False)
res["trim_file_overwrite"] = file_to_file(dst, 128 << 20, True)
res["trim_file_4_parts"] = file_to_file(dst, 128 << 20, False, parts=4)
res["trim_file_4_parts_64M_chunks"] = file_to_file(dst, 64 << 20, False, parts=4)
res["trim_file_8_parts_64M_chunks"] = file_to_file(dst, 64 << 20, False, parts=8)
if os.path.isdir("/dev/shm"):
    res["trim_file_to_shm"] = file_to_file("/dev/shm/atr_e2e_out.fastq", 128 << 20, False)
os.remove(src)
# paired-end, insert aligner: two files in lock step
from bench_fastq_pe import device_fastq as device_fastq_pe
npairs = nreads // 2
w = synth.workload("C3", 0, npairs, device="cuda")
paths = ["/tmp/atr_e2e_%s.fastq" % t for t in ("i1", "i2", "o1", "o2")]
sizes = 0
for k, key in enumerate(("reads1", "reads2")):
    d, nb = device_fastq_pe(w[key], "12"[k])
    with open(paths[k], "wb") as fh:
        fh.write(bytes(d[:nb].cpu().numpy().tobytes()))
    sizes += nb
del w
pe = pipeline_from_args("--aligner insert -a %s -A %s -q 20 -m 30" % (synth.PE_ADAPTER1, synth.PE_ADAPTER2))
for label, keep, parts in (("trim_files_paired", False, 1), ("trim_files_paired_overwrite", True, 1), ("trim_files_paired_4_parts", False, 4)):
    outs = paths[2:] if parts == 1 else ["%s.part%d" % (p, i) for p in paths[2:] for i in range(parts)]
    pe.trim_files(*paths, chunk_bytes=128 << 20, keep_output=keep, output_parts=parts)
    total, stages = 0.0, {}
    for _ in range(steps):
        if not keep:
            for p in outs:
                os.remove(p)
        t0 = time.perf_counter()
        pe.trim_files(*paths, chunk_bytes=128 << 20, keep_output=keep, output_parts=parts)
        total += time.perf_counter() - t0
        for k, v in pe.stage_seconds.items():
            stages[k] = stages.get(k, 0.0) + v
    ms3 = total / steps * 1e3
    out_bytes = sum(os.path.getsize(p) for p in outs)
    if parts > 1:
        for p in outs:
            os.remove(p)
    res[label] = {"npairs": npairs, "output_parts": parts, "ms": ms3, "pairs_per_s": npairs / ms3 * 1e3, "reads_per_s": 2 * npairs / ms3 * 1e3,
                  "host_GBps": (sizes + out_bytes) / ms3 / 1e6,
                  "wait_ms_per_run": {k: round(v / steps * 1e3, 2) for k, v in stages.items()}}
for p in paths:
    if os.path.exists(p):
        os.remove(p)
print(json.dumps(res))
