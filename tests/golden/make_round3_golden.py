#!/usr/bin/env python3
"""Round-3 golden vectors, generated from the REFERENCE itself (build container only: needs
/root/reference; the reference package is copied to a scratch directory and its Cython modules are
built there by tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round3_golden.py [fixture]

Fixtures (inputs are generated here; expected values are what the reference returned):
    correct_errors_fuzz.json.gz   ErrorCorrectorMixin.correct_errors (commands/trim/modifiers.py:219-350) on
                                  read pairs with qualities: the three mismatch actions, min_qual_difference,
                                  truncate_seqs, unequal lengths, synthesized overlaps (:408-415), index
                                  wrap-around / IndexError, bases without a complement (KeyError), reads without
                                  qualities (ValueError), equal-quality ties decided by the mean (:303-322)
While doing so it pins the oracle's C restatement (oracle.correct_errors): every case must agree, or the
script aborts.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import build_reference, dump  # noqa: E402

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    build_reference("/tmp/atropos_ref_build")
    from atropos.align import InsertAligner
    from atropos.commands.trim.modifiers import ErrorCorrectorMixin
    from atropos.io.seqio import Sequence
    from oracle import oracle as O
    O.build(force=True)
    rng = random.Random(20260929)

    def rseq(n, alpha="ACGT"):
        return "".join(rng.choice(alpha) for _ in range(n))

    def rc(s):
        return "".join(COMP[c] for c in reversed(s))

    def noisy(s, p_sub, p_n):
        out = []
        for c in s:
            r = rng.random()
            out.append(rng.choice("ACGT") if r < p_sub else ("N" if r < p_sub + p_n else c))
        return "".join(out)

    def quals(n, mode):
        if mode == "flat":                       # equal qualities everywhere: liberal defers every mismatch
            return chr(33 + rng.randint(2, 40)) * n
        if mode == "two":                        # two levels one apart: mean differences around +-1
            a = rng.randint(10, 30)
            return "".join(chr(33 + a + (rng.random() < 0.5)) for _ in range(n))
        return "".join(chr(33 + rng.randint(2, 40)) for _ in range(n))

    def run_reference(c):
        mix = ErrorCorrectorMixin(c["action"], c["mqd"])
        r1 = Sequence(name="p", sequence=c["seq1"], qualities=c["qual1"])
        r2 = Sequence(name="p", sequence=c["seq2"], qualities=c["qual2"])
        try:
            mix.correct_errors(r1, r2, tuple(c["im"]), truncate_seqs=c["truncate"])
        except (KeyError, IndexError, ValueError) as exc:
            return dict(error=type(exc).__name__)
        return dict(seq1=r1.sequence, qual1=r1.qualities, seq2=r2.sequence, qual2=r2.qualities,
                    corrected=[int(r1.corrected), int(r2.corrected)], pairs=mix.corrected_pairs,
                    bp=list(mix.corrected_bp))

    def run_oracle(c):
        try:
            s1, q1, s2, q2, ch = O.correct_errors(c["seq1"], c["qual1"], c["seq2"], c["qual2"], c["im"], c["action"],
                                                  c["mqd"], c["truncate"])
        except (KeyError, IndexError, ValueError) as exc:
            return dict(error=type(exc).__name__)
        return dict(seq1=s1, qual1=q1, seq2=s2, qual2=q2, corrected=list(ch), pairs=int(ch[0] > 0 or ch[1] > 0),
                    bp=list(ch))

    if only in (None, "correct_errors"):
        ia = InsertAligner("AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", read_wildcards=True)
        cases = []
        while len(cases) < 4000:
            kind = rng.random()
            n1 = rng.choice([20, 30, 50, 50, 75, 100, 150])
            n2 = n1 if rng.random() < 0.7 else max(8, n1 + rng.randint(-12, 12))
            action = rng.choice(["N", "conservative", "liberal", "liberal"])
            mqd = rng.choice([1, 1, 1, 2, 5])
            qmode = rng.choice(["random", "random", "flat", "two"])
            if kind < 0.55:
                # an overlapping pair: fragment shorter than the reads, adapters behind it, the tuple from the reference
                f = rng.randint(8, max(9, min(n1, n2)))
                frag = rseq(f)
                p_sub, p_n = rng.choice([(0.02, 0.0), (0.05, 0.01), (0.1, 0.03)])
                seq1 = noisy((frag + ia.adapter1 + rseq(n1))[:n1], p_sub, p_n)
                seq2 = noisy((rc(frag) + ia.adapter2 + rseq(n2))[:n2], p_sub, p_n)
                m = ia.match_insert(seq1, seq2)
                if m is None:
                    continue
                im, truncate = list(m[0][:4]), True
            elif kind < 0.75:
                # complementary adapter matches: the synthesized tuple of modifiers.py:408-415 (any lengths)
                rstart = rng.randint(0, min(n1, n2) + 3)
                frag = rseq(rstart)
                seq1 = noisy((frag + rseq(n1))[:n1], 0.05, 0.02)
                seq2 = noisy((rc(frag) + rseq(n2))[:n2], 0.05, 0.02)
                im, truncate = [n2 - rstart, n2, 0, rstart], True
            elif kind < 0.9:
                # MergeOverlapping's call: a semi-global alignment tuple, truncate_seqs False (:900-902)
                seq1, seq2 = rseq(n1, "ACGTN" if rng.random() < 0.3 else "ACGT"), rseq(n2)
                ov = rng.randint(1, min(n1, n2))
                if rng.random() < 0.5:
                    im = [0, ov, n1 - ov, n1]
                    seq2 = noisy(rc(seq1[n1 - ov:]).replace("N", "A") + rseq(n2), 0.08, 0.02)[:n2]
                else:
                    im = [n2 - ov, n2, 0, ov]
                    seq2 = noisy(rseq(n2) + rc(seq1[:ov]).replace("N", "A"), 0.08, 0.02)[-n2:]
                truncate = False
            else:
                # arbitrary small tuples: wrap-around, IndexError, empty ranges
                seq1, seq2 = rseq(n1), rseq(n2)
                im = [rng.randint(-3, n2 + 3), rng.randint(-3, n2 + 6), rng.randint(-4, n1 + 2), rng.randint(-2, n1 + 6)]
                truncate = rng.random() < 0.5
            qual1, qual2 = quals(len(seq1), qmode), quals(len(seq2), qmode)
            r = rng.random()
            if r < 0.03:
                qual1 = qual2 = None                                 # no qualities: ValueError unless action 'N'
            elif r < 0.05:
                qual1 = None
            elif r < 0.08:                                           # a base without a complement
                pos = rng.randrange(len(seq2))
                seq2 = seq2[:pos] + rng.choice("X.") + seq2[pos + 1:]
            elif r < 0.10:
                pos = rng.randrange(len(seq1))
                seq1 = seq1[:pos] + rng.choice("Xx") + seq1[pos + 1:]
            elif r < 0.14:                                           # soft-masked bases
                seq1, seq2 = seq1.lower() if rng.random() < 0.5 else seq1, seq2.lower()
            c = dict(seq1=seq1, qual1=qual1, seq2=seq2, qual2=qual2, im=im, action=action, mqd=mqd, truncate=truncate)
            c["out"] = run_reference(c)
            got = run_oracle(c)
            if got != c["out"]:
                raise SystemExit("oracle.correct_errors differs from the reference on %r:\n reference %r\n oracle    %r"
                                 % ({k: v for k, v in c.items() if k != "out"}, c["out"], got))
            cases.append(c)
        dump("correct_errors_fuzz.json.gz", cases)
        errs = {}
        for c in cases:
            key = c["out"].get("error", "changed" if sum(c["out"].get("corrected", [0])) else "unchanged")
            errs[key] = errs.get(key, 0) + 1
        print("correct_errors: %d cases, oracle == reference on all; outcomes %r" % (len(cases), errs))


if __name__ == "__main__":
    main()
