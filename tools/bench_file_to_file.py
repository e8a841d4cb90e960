#!/usr/bin/env python3
"""File -> file rate of the single-end FASTQ pipeline (page-cached input file, output into /tmp): reads/s, the waits per
stage, the PCIe floor of the same bytes.  usage: tools/bench_file_to_file.py [nreads] [steps] [parts,...] [chunk MB,...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from atropos_amd.trim import pipeline_from_args        # noqa: E402
from bench_fastq import device_fastq                   # noqa: E402

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
parts_list = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 8]
chunks = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [128]
data, nbytes = device_fastq(nreads)
src, dst = "/tmp/atr_f2f_in.fastq", "/tmp/atr_f2f_out.fastq"
with open(src, "wb") as fh:
    fh.write(bytes(data[:nbytes].cpu().numpy().tobytes()))
del data
pipe = pipeline_from_args("-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -q 20 --trim-n -m 20")
res = {"nreads": nreads, "input_bytes": nbytes, "io_threads": __import__("atropos_amd.fastq", fromlist=["x"]).IO_THREADS, "runs": []}
for parts in parts_list:
    for chunk_mb in chunks:
        names = [dst] if parts == 1 else ["%s.part%d" % (dst, i) for i in range(parts)]
        pipe.trim_file(src, dst, chunk_bytes=chunk_mb << 20, output_parts=parts)
        total, stages = 0.0, {}
        for _ in range(steps):
            for name in names:
                if os.path.exists(name):
                    os.remove(name)
            t0 = time.perf_counter()
            pipe.trim_file(src, dst, chunk_bytes=chunk_mb << 20, output_parts=parts)
            total += time.perf_counter() - t0
            for k, v in pipe.stage_seconds.items():
                stages[k] = stages.get(k, 0.0) + v
        out_bytes = sum(os.path.getsize(name) for name in names)
        for name in names:
            os.remove(name)
        ms = total / steps * 1e3
        res["runs"].append({"parts": parts, "chunk_mb": chunk_mb, "ms": round(ms, 2), "reads_per_s": nreads / ms * 1e3,
                            "host_GBps": (nbytes + out_bytes) / ms / 1e6, "output_bytes": out_bytes,
                            "wait_ms": {k: round(v / steps * 1e3, 2) for k, v in stages.items()}})
os.remove(src)
print(json.dumps(res))
