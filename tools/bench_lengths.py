#!/usr/bin/env python3
"""Aligner.locate_batch on C2-like batches of other read lengths (same generator, adapter and parameters): the
two-pass pre-pass on plane64 against the one-pass filtered pipeline on tile64, records compared.

    python tools/bench_lengths.py [reads]            # GPU box; one JSON line per length
    python tools/bench_lengths.py [reads] ragged     # 150-base reads cut to lengths spread over 100 .. 150 (a
                                                     # quality-trimmed file): ragged batches on both pipelines"""
import json
import sys

import torch

sys.path.insert(0, ".")
from atropos_amd import synth
from atropos_amd.align import Aligner


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def ragged(reads):
    from atropos_amd.batch import ReadBatch
    al = Aligner(synth.TRUSEQ_34, 0.1, 14, False, False, 3, 1)
    mat = synth.single_end(0, reads, 150, synth.TRUSEQ_34, synth.SEEDS["C2"], "cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    for lo in (150, 140, 100, 20):
        lens = torch.randint(lo, 151, (reads,), generator=gen, device="cuda", dtype=torch.int32)
        table = al._backend.translate_table(al.table_kind)
        tiles = ReadBatch.from_ascii(mat, lens, 150, al.table_kind, table, al._backend)
        planes = ReadBatch.from_ascii(mat, lens, 150, al.table_kind, table, al._backend, planes=True)
        one = timed(lambda: al.locate_batch(tiles, path="filtered"))
        two = timed(lambda: al.locate_batch(planes))
        print(json.dumps({"read_len": "%d..150" % lo, "reads": reads, "one_pass_ms": one, "one_pass_reads_per_s": reads / (one * 1e-3),
                          "two_pass_ms": two, "two_pass_reads_per_s": reads / (two * 1e-3),
                          "equal": bool(torch.equal(al.locate_batch(planes).records, al.locate_batch(tiles, path="filtered").records)),
                          "matched": int((al.locate_batch(planes).records[:, 1] >= 0).sum())}), flush=True)


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    if len(sys.argv) > 2 and sys.argv[2] == "ragged":
        return ragged(reads)
    for n in (76, 100, 125, 150, 175, 250, 300):
        mat = synth.single_end(0, reads, n, synth.TRUSEQ_34, synth.SEEDS["C2"], "cuda")
        al = Aligner(synth.TRUSEQ_34, 0.1, 14, False, False, 3, 1)
        tiles = al.pack(mat, layout="tile64")
        one = timed(lambda: al.locate_batch(tiles, path="filtered"))
        line = {"read_len": n, "reads": reads, "one_pass_ms": one, "one_pass_reads_per_s": reads / (one * 1e-3)}
        try:
            planes = al.pack(mat, layout="plane64")
        except Exception as exc:                      # outside the two-pass envelope (word count not instantiated)
            line["two_pass"] = "refused: %s" % exc
        else:
            two = timed(lambda: al.locate_batch(planes))
            line.update(two_pass_ms=two, two_pass_reads_per_s=reads / (two * 1e-3),
                        equal=bool(torch.equal(al.locate_batch(planes).records, al.locate_batch(tiles, path="filtered").records)))
        print(json.dumps(line), flush=True)
        del mat, tiles


if __name__ == "__main__":
    main()
