#!/usr/bin/env python3
"""Round-5 golden vectors, generated from the REFERENCE itself (build container only: needs /root/reference; the
reference package is copied to a scratch directory and its Cython modules are built there by
tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round5_golden.py

Fixture (inputs are generated here; expected values are what the reference returned):
    insert_longer.json.gz   InsertAligner.match_insert (align/__init__.py:250-377) on pairs of 2 x 321 .. 600 bases --
                            beyond the 320 bases of the insert kernel: inserts shorter and longer than the reads,
                            noise, unrelated reads, low-complexity pairs; six configurations.
    long_multi_compare.json.gz  MultiAligner.locate (_align.pyx:548-787) and compare_prefixes / compare_suffixes
                            (_align.pyx:501-544, align/__init__.py:28-44) on strings of 737 .. 3 000 characters -- past
                            the 736 bases of the batch pipelines, compare references past the 1 024 bytes one device
                            call takes: planted common stretches, noise, N, all four wildcard settings, five flag sets;
                            LinkedAdapter.match_to (adapters/__init__.py:648-706) on reads of 737 .. 3 000 bases.
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


def long_multi_compare():
    build_reference("/tmp/atropos_ref_build")
    from atropos.align import MultiAligner, compare_prefixes, compare_suffixes
    rng = random.Random(20261002)

    def rseq(n, alpha="ACGT"):
        return "".join(rng.choice(alpha) for _ in range(n))

    def noise(s, p):
        return "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in s)

    compare, multi = [], []
    for it in range(36):
        n = rng.choice([737, 800, 1000, 1025, 1500, 3000])
        m = rng.choice([n, n - 5, 700, 1200, 2049])
        core = rseq(max(m, n))
        ref, q = noise(core[:m], 0.02), noise(core[:n], 0.02)
        if it % 9 == 8:
            ref, q = ref[::-1], q[::-1]                       # (common SUFFIX instead of a common prefix)
        outs = []
        for wr in (False, True):
            for wq in (False, True):
                outs.append([list(compare_prefixes(ref, q, wr, wq)), list(compare_suffixes(ref, q, wr, wq))])
        compare.append(dict(ref=ref, query=q, out=outs))
    for it in range(30):
        n = rng.choice([737, 800, 1000, 1500, 2500])
        m = rng.choice([n, 300, 900, 1600])
        L = rng.randint(20, min(m, n))
        frag = rseq(L)
        ref = frag + rseq(m - L) if rng.random() < 0.5 else rseq(m - L) + frag
        q = rseq(n - L) + noise(frag, 0.03) if rng.random() < 0.5 else noise(frag, 0.03) + rseq(n - L)
        runs = []
        for flags in (15, 9, 5, 6, 14):
            mo, e = rng.choice([1, 10]), rng.choice([0.0, 0.05, 0.1, 0.2])
            res = MultiAligner(e, flags, mo).locate(ref, q)
            runs.append(dict(flags=flags, min_overlap=mo, e=e, out=None if res is None else [list(t) for t in res]))
        multi.append(dict(ref=ref, query=q, runs=runs))
    # LinkedAdapter.match_to (adapters/__init__.py:648-706) on reads of 737 .. 3 000 bases
    from atropos.adapters import LinkedAdapter
    from atropos.io.seqio import Sequence
    linked = []
    for it in range(60):
        fr, bk = rseq(rng.randint(8, 20)), rseq(rng.randint(15, 34))
        n = rng.choice([737, 900, 1200, 2000, 3000])
        f = rng.randint(50, n)
        read = ((noise(fr, 0.03) if rng.random() < 0.8 else rseq(len(fr))) + rseq(f)
                + (noise(bk, 0.04) if rng.random() < 0.8 else "") + rseq(n))[:n]
        kw = dict(max_error_rate=rng.choice([0.1, 0.12, 0.2]), min_overlap=3)
        m = LinkedAdapter(fr, bk, **kw).match_to(Sequence("r", read))
        linked.append(dict(front=fr, back=bk, kw=kw, read=read,
                           out=None if m is None else [match_fields(m.front_match), match_fields(m.back_match)]))
    dump("long_multi_compare.json.gz", dict(compare=compare, multi=multi, linked=linked))
    print("long_multi_compare: %d compare pairs x 8, %d multi pairs x 5 (%d with hits), %d linked reads (%d with a match)" % (
        len(compare), len(multi), sum(any(r["out"] for r in c["runs"]) for c in multi), len(linked),
        sum(c["out"] is not None for c in linked)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "long_multi_compare":
        long_multi_compare()
    else:
        main()
        long_multi_compare()
