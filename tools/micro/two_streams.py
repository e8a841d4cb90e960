#!/usr/bin/env python3
"""C2's call issued on one stream back to back, and alternating between two streams with a workspace each (batch
i + 1's pre-pass next to batch i's DP phase): does the device overlap them?   python tools/micro/two_streams.py [reads]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from atropos_amd import _lib, synth
from atropos_amd.align import Aligner


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    mat = synth.single_end(0, reads, 150, synth.TRUSEQ_34, synth.SEEDS["C2"], "cuda")
    als, batches = [], []
    for _ in range(2):
        _lib.set_backend(_lib.HipBackend(), _test_double=True)
        al = Aligner(synth.TRUSEQ_34, 0.1, 14, False, False, 3, 1)
        als.append(al)
        batches.append(al.pack(mat))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ref = als[0].locate_batch(batches[0]).records.clone()

    def run(nstreams, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for i in range(steps):
            k = i % nstreams
            with torch.cuda.stream(streams[k]):
                outs.append(als[k].locate_batch(batches[k]).records)
            if len(outs) > 4:
                outs.pop(0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert torch.equal(outs[-1], ref) and torch.equal(outs[-2], ref)
        return dt

    for n in (1, 2):
        run(n, 10)
        dt = min(run(n, 40) for _ in range(3))
        print(json.dumps({"streams": n, "reads": reads, "ms_per_call": dt * 1e3, "reads_per_s": reads / dt}), flush=True)

    # one call = the batch as two halves on two streams, forked from and joined to the caller's stream
    half = (reads // 2 + 63) // 64 * 64
    hb = [als[0].pack(mat[:half]), als[1].pack(mat[half:])]
    main_stream = torch.cuda.current_stream()

    def run_split(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            outs = []
            for k in range(2):
                streams[k].wait_stream(main_stream)
                with torch.cuda.stream(streams[k]):
                    outs.append(als[k].locate_batch(hb[k]).records)
            for k in range(2):
                main_stream.wait_stream(streams[k])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert torch.equal(torch.cat(outs), ref)
        return dt

    run_split(10)
    dt = min(run_split(40) for _ in range(3))
    print(json.dumps({"streams": "2 halves per call, joined", "reads": reads, "ms_per_call": dt * 1e3, "reads_per_s": reads / dt}), flush=True)


if __name__ == "__main__":
    main()
