#!/usr/bin/env python3
"""Round-4 golden vectors, generated from the REFERENCE itself (build container only: needs /root/reference; the
reference package is copied to a scratch directory and its Cython modules are built there by
tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round4_golden.py

Fixtures (inputs are generated here; expected values are what the reference returned):
    long_reads.json.gz    Aligner.locate (_align.pyx:266-491) on reads of 737 .. 2 600 bases -- beyond the 736 bases
                          of the batch pipelines -- with the adapter whole, edited or cut at the columns where the
                          long-read sweep moves its origin base (multiples of 256), at both read ends and twice;
                          all flag sets, indel costs, wildcard modes.  A read is stored as (seed, length, planted
                          pieces) and rebuilt by tests/_cases.long_read_case(), so that the file stays small.
    long_pairs.json.gz    Aligner.locate with references of 321 .. 1 500 bases (beyond the per-pair aligner's packed
                          cell word): overlaps at either end, inside, absent; all flag sets.  Stored as seeds
                          (tests/_cases.long_pair_case()).
While doing so it pins the oracle's C restatement on every case, or aborts.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import build_reference, dump  # noqa: E402


def main():
    build_reference("/tmp/atropos_ref_build")
    from atropos.align import Aligner
    from oracle import oracle as O
    from tests._cases import long_read_case, mutate, rseq
    O.build(force=True)
    rng = random.Random(20260930)
    FLAGS = [14] * 10 + [11, 15, 15, 10, 10, 6, 6] + [14, 11, 15, 10, 6, 9, 8, 2, 1, 4, 0, 12, 3, 5, 7, 13]
    cases = []
    for _ in range(700):
        m = rng.choice([rng.randint(1, 12), rng.randint(8, 40), rng.randint(30, 128)])
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.8 else "ACGTNRYKMSWBDHV")
        c = dict(ref=ref, e=rng.choice([0, 0.1, 0.1, 0.12, 0.2, 0.2, 0.3]), flags=rng.choice(FLAGS),
                 ic=rng.choice([1, 1, 2, 100000]), mo=rng.choice([1, 3, 5, 10]), wr=rng.random() < 0.2, wq=rng.random() < 0.2)
        n = rng.choice([737, 767, 768, 769, 1023, 1024, 1025, 1279, 1281, rng.randint(737, 1300), rng.randint(737, 2600)])
        pieces = []
        w = rng.random()
        part = mutate(rng, ref, rng.choice([0, 0, 0.04, 0.04, 0.1]))
        if w < 0.3:                                          # around a step of the origin base
            pieces.append([rng.choice([256, 512, 768, 1024, 1280, 1536]) + rng.randint(-m - 3, 3), part])
        elif w < 0.45:                                       # cut by the read end
            cut = part[:rng.randint(1, max(1, len(part)))]
            pieces.append([n - len(cut), cut])
        elif w < 0.55:                                       # at the read start, its head possibly missing
            pieces.append([0, part[rng.randint(0, max(0, len(part) - 1)):]])
        elif w < 0.75:                                       # twice
            a, b = mutate(rng, ref, 0.1), mutate(rng, ref, rng.choice([0, 0.1]))
            if rng.random() < 0.5:
                a, b = b, a
            p1 = rng.randint(0, max(0, n - 2 * m - 20))
            pieces += [[p1, a], [p1 + len(a) + rng.randint(0, 700), b]]
        elif w < 0.9:
            pieces.append([rng.randint(0, n), part])
        c.update(seed=rng.randint(0, 1 << 30), n=n, pieces=pieces, alpha="ACGTN" if rng.random() < 0.15 else "ACGT")
        q = long_read_case(c)
        assert len(q) == n
        out = Aligner(c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]).locate(q)
        orc = O.locate(c["ref"], q, c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        assert out == orc, (c, out, orc)
        c["out"] = out
        cases.append(c)
    far = sum(1 for c in cases if c["out"] is not None and c["out"][3] > 767)
    print("%d cases, %d matched, %d with a query stop beyond column 767" % (len(cases), sum(c["out"] is not None for c in cases), far))
    dump("long_reads.json.gz", cases)

    # long_pairs.json.gz: Aligner(reference of 321 .. 1 500 bases).locate(query): beyond the per-pair aligner's 320
    from tests._cases import long_pair_case
    pairs = []
    for _ in range(400):
        m = rng.choice([rng.randint(321, 700), rng.randint(321, 1500), rng.randint(256, 330)])
        n = rng.choice([rng.randint(321, 900), rng.randint(0, 320), rng.randint(600, 1500)])
        c = dict(seed=rng.randint(0, 1 << 30), m=m, n=n, alpha="ACGT" if rng.random() < 0.8 else "ACGTN",
                 kind=rng.choice([0, 0, 1, 2, 3]), a=rng.randint(0, min(300, m - 1)), b=rng.randint(0, 50),
                 rate=rng.choice([0, 0.03, 0.05, 0.1]), e=rng.choice([0, 0.05, 0.1, 0.2, 0.3]),
                 flags=rng.choice([15, 9, 14, 11, 14, 15, 8, 2, 0, 5, 10, 6]), ic=rng.choice([1, 1, 2, 100000]),
                 mo=rng.choice([1, 3, 10]), wr=rng.random() < 0.15, wq=rng.random() < 0.15)
        ref, q = long_pair_case(c)
        out = Aligner(ref, c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]).locate(q)
        orc = O.locate(ref, q, c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        assert out == orc, (c, out, orc)
        c["out"] = out
        pairs.append(c)
    print("%d long pairs, %d matched" % (len(pairs), sum(c["out"] is not None for c in pairs)))
    dump("long_pairs.json.gz", pairs)


if __name__ == "__main__":
    main()
