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
       "output_bytes": len(out), "ms": ms, "reads_per_s": nreads / ms * 1e3, "host_GBps": (nbytes + len(out)) / ms / 1e6}
# file -> file through page-locked staging buffers with read-ahead / write-behind threads
src, dst = "/tmp/atr_e2e_in.fastq", "/tmp/atr_e2e_out.fastq"
with open(src, "wb") as fh:
    fh.write(host)
pipe.trim_file(src, dst, chunk_bytes=256 << 20)
t0 = time.perf_counter()
for _ in range(steps):
    counts = pipe.trim_file(src, dst, chunk_bytes=256 << 20)
ms2 = (time.perf_counter() - t0) / steps * 1e3
assert open(dst, "rb").read() == out
res["trim_file"] = {"ms": ms2, "reads_per_s": nreads / ms2 * 1e3, "host_GBps": (nbytes + len(out)) / ms2 / 1e6,
                    "chunk_bytes": 256 << 20, "counts": counts}
os.remove(src)
os.remove(dst)
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
pe.trim_files(*paths, chunk_bytes=128 << 20)
t0 = time.perf_counter()
for _ in range(steps):
    pe.trim_files(*paths, chunk_bytes=128 << 20)
ms3 = (time.perf_counter() - t0) / steps * 1e3
out_bytes = os.path.getsize(paths[2]) + os.path.getsize(paths[3])
res["trim_files_paired"] = {"npairs": npairs, "ms": ms3, "pairs_per_s": npairs / ms3 * 1e3,
                            "host_GBps": (sizes + out_bytes) / ms3 / 1e6}
for p in paths:
    os.remove(p)
print(json.dumps(res))
