#!/usr/bin/env python3
"""Round-5 golden vectors, generated from the REFERENCE itself (build container only: needs /root/reference; the
reference package is copied to a scratch directory and its Cython modules are built there by
tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round5_golden.py

Fixture (inputs are generated here; expected values are what the reference returned):
    insert_longer.json.gz   InsertAligner.match_insert (align/__init__.py:250-377) on pairs of 2 x 321 .. 600 bases --
                            beyond the 320 bases of the insert kernel: inserts shorter and longer than the reads,
                            noise, unrelated reads, low-complexity pairs; six configurations.
The oracle's restatement is pinned on every case as well, or the script aborts.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import build_reference, dump, match_fields  # noqa: E402


def main():
    build_reference("/tmp/atropos_ref_build")
    from atropos.align import InsertAligner
    from atropos.util import reverse_complement
    from atropos_amd import synth
    from oracle import oracle as O
    O.build(force=True)
    rng = random.Random(20261001)
    A1, A2 = synth.PE_ADAPTER1, synth.PE_ADAPTER2

    def rseq(n, alpha="ACGT"):
        return "".join(rng.choice(alpha) for _ in range(n))

    def noise(s, p):
        return "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in s)

    cases, pinned = [], 0
    cfgs = [dict(), dict(max_insert_mismatch_frac=0.1, max_adapter_mismatch_frac=0.1), dict(read_wildcards=True),
            dict(adapter_wildcards=False), dict(min_insert_overlap=5, min_adapter_overlap=3),
            dict(insert_max_rmp=1e-3, adapter_max_rmp=1e-2)]
    for cfg in cfgs:
        ref = InsertAligner(A1, A2, **cfg)
        orc = O.InsertOracle(A1, A2, **cfg)
        for it in range(40):
            n = rng.choice([321, 322, 352, 400, 400, 480, 512, 600])
            f = rng.randint(0, int(1.4 * n))
            F = rseq(f)
            r1 = (F + A1 + rseq(n))[:rng.choice([n, n, n - 3, n - 40])]
            r2 = (reverse_complement(F) + A2 + rseq(n))[:rng.choice([n, n, n - 1, n - 70])]
            p = rng.choice([0, 0.01, 0.03, 0.1])
            r1, r2 = noise(r1, p), noise(r2, p)
            if rng.random() < 0.08:
                r1 = rseq(len(r1))
            if rng.random() < 0.04:
                r1, r2 = "A" * len(r1), "T" * len(r2)
            if rng.random() < 0.05:
                r1, r2 = ("AC" * n)[:len(r1)], ("GT" * n)[:len(r2)]
            res = ref.match_insert(r1, r2)
            out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
            try:
                mine = orc.match_insert(r1, r2)
                mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                                  None if mine[2] is None else list(mine[2])]
                assert mine == out, (cfg, r1, r2, mine, out)
                pinned += 1
            except (ValueError, OverflowError):
                pass                                        # (the C restatement's own length limit)
            cases.append(dict(a1=A1, a2=A2, r1=r1, r2=r2, kw=cfg, out=out))
    dump("insert_longer.json.gz", cases)
    print("insert_longer: %d cases, %d with a match, oracle pinned on %d" % (len(cases), sum(c["out"] is not None for c in cases), pinned))


if __name__ == "__main__":
    main()
