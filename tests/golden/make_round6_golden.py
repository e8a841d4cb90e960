#!/usr/bin/env python3
"""Round-6 golden vectors, generated from the REFERENCE itself (build container only: needs /root/reference; the
reference package is copied to a scratch directory and its Cython modules are built there by
tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round6_golden.py

Fixture (inputs are generated here; expected values are what the reference returned):
    qualtrim_fuzz.json.gz   quality_trim_index / nextseq_trim_index (commands/trim/_qualtrim.pyx:7-84) and
                            NEndTrimmer (commands/trim/modifiers.py:766-784): quality strings that decay, recover,
                            sit at the cutoff, dip below base; cutoffs 0 .. 40 on either end, both quality bases;
                            G runs at the 3' end (upper and lower case); N runs at both ends, all-N and empty reads.
The oracle's restatement (oracle/align_oracle.c: orc_quality_trim_index, orc_nextseq_trim_index, orc_n_end_trim) is
pinned on every case and on 200 000 more that are not committed, or the script aborts.
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
    from atropos.commands.trim._qualtrim import quality_trim_index, nextseq_trim_index
    from atropos.commands.trim.modifiers import NEndTrimmer
    from atropos.io.seqio import Sequence
    from oracle import oracle as O
    O.build(force=True)
    rng = random.Random(20261006)

    def quals(n, base):
        kind = rng.randrange(6)
        if kind == 0:       # decaying towards the 3' end
            slope = rng.uniform(0, 0.5)
            q = [max(0, min(41, int(38 - slope * i + rng.randint(-4, 4)))) for i in range(n)]
        elif kind == 1:     # bad at both ends
            q = [max(0, min(41, int(40 - 60 * abs(i / max(1, n - 1) - 0.5) ** 2 * rng.uniform(0, 4) + rng.randint(-3, 3))))
                 for i in range(n)]
        elif kind == 2:     # at / around one value: ties of the running sums
            c = rng.randint(0, 40)
            q = [max(0, c + rng.choice([-1, 0, 0, 0, 1])) for _ in range(n)]
        elif kind == 3:     # dips and recoveries
            q = [rng.choice([2, 2, 38, 38, 38, 20]) for _ in range(n)]
        elif kind == 4:     # uniform noise
            q = [rng.randint(0, 41) for _ in range(n)]
        else:               # characters below the base (negative qualities) as well
            q = [rng.randint(-3, 12) for _ in range(n)]
        return "".join(chr(max(1 if base == 33 else 33, base + v)) for v in q)

    def bases(n):
        s = [rng.choice("ACGT") for _ in range(n)]
        if rng.random() < 0.5:
            g = rng.randint(0, n)
            for i in range(n - g, n):
                s[i] = rng.choice("GGGGGGGg") if rng.random() < 0.9 else rng.choice("ACT")
        if rng.random() < 0.3:
            for i in range(rng.randint(0, min(n, 6))):
                s[i] = "N"
            for i in range(rng.randint(0, min(n, 6))):
                s[n - 1 - i] = rng.choice("NNNn")
        if rng.random() < 0.03:
            s = ["N"] * n
        return "".join(s)

    ntrim = NEndTrimmer()

    def one():
        n = rng.choice([0, 1, 2, 5, 20, 50, 100, 150, 150, 250, rng.randint(0, 320)])
        base = rng.choice([33, 33, 33, 64])
        q, s = quals(n, base), bases(n)
        cf, cb = rng.choice([0, 0, 10, 15, 20, 30, rng.randint(0, 41)]), rng.choice([0, 10, 15, 20, 20, 30, rng.randint(0, 41)])
        cg = rng.choice([1, 10, 20, 20, 30, rng.randint(0, 41)])
        want_q = list(quality_trim_index(q, cf, cb, base))
        want_g = int(nextseq_trim_index(Sequence("r", s, q), cg, base)) if n else 0
        if n:
            t = ntrim(Sequence("r", s, q))
            want_n = [t.sequence, t.qualities]
        else:
            want_n = [s, q]
        # the oracle, pinned
        assert list(O.quality_trim_index(q, cf, cb, base)) == want_q, (q, cf, cb, base, want_q)
        if n:
            assert O.nextseq_trim_index(s, q, cg, base) == want_g, (s, q, cg, base, want_g)
        a, b = O.n_end_trim(s)
        got_n = [s[a:b], q[a:b]] if a < b else ["", ""]
        assert got_n == want_n, (s, want_n, got_n)
        return {"seq": s, "qual": q, "base": base, "cf": cf, "cb": cb, "cg": cg, "qtrim": want_q, "nextseq": want_g,
                "nend": want_n}

    cases = [one() for _ in range(3000)]
    for _ in range(200000):
        one()
    dump("qualtrim_fuzz.json.gz", {"cases": cases})
    print("qualtrim: %d committed + 200000 uncommitted cases, oracle pinned on all" % len(cases))


if __name__ == "__main__":
    main()
