#!/usr/bin/env python3
"""Generates tests/golden/merge_fuzz.json.gz: read pairs and what the REFERENCE's
MergeOverlapping modifier (commands/trim/modifiers.py:864-931) does with them.  Run in this
container only (imports the reference from a scratch build); the committed file is data.

usage: python tests/golden/make_merge_golden.py [--scratch /tmp/oracle_ref]
"""
import argparse
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def rseq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def noisy(rng, s, p):
    out = []
    for ch in s:
        r = rng.random()
        if r < p * 0.8:
            out.append(rng.choice("ACGT"))
        elif r < p * 0.9:
            continue
        elif r < p:
            out.append(ch + rng.choice("ACGT"))
        else:
            out.append(ch)
    return "".join(out)


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def make_pair(rng, L):
    """Paired reads of length <= L from a fragment of random size (overlap from none to total)."""
    kind = rng.random()
    if kind < 0.15:
        frag = rseq(rng, rng.randint(2 * L + 10, 3 * L))            # no overlap
    elif kind < 0.3:
        frag = rseq(rng, rng.randint(max(1, L // 3), L))           # fragment shorter than the reads
    else:
        frag = rseq(rng, rng.randint(L, 2 * L))                    # partial overlap
    l1 = rng.randint(max(1, L - 30), L) if rng.random() < 0.3 else L
    l2 = rng.randint(max(1, L - 30), L) if rng.random() < 0.3 else L
    p = rng.choice([0, 0, 0.01, 0.03, 0.08])
    r1 = noisy(rng, frag[:l1], p)
    r2 = noisy(rng, rc(frag)[:l2], p)
    if rng.random() < 0.1 and r1:
        k = rng.randrange(len(r1))
        r1 = r1[:k] + "N" + r1[k + 1:]
    q1 = "".join(chr(33 + rng.randint(2, 40)) for _ in r1)
    q2 = "".join(chr(33 + rng.randint(2, 40)) for _ in r2)
    return r1, q1, r2, q2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/oracle_ref")
    args = ap.parse_args()
    sys.path.insert(0, args.scratch)
    from atropos.commands.trim.modifiers import MergeOverlapping
    from atropos.io.seqio import Sequence

    rng = random.Random(20260928)
    configs = [dict(min_overlap=0.9, error_rate=0.1, mismatch_action=None),
               dict(min_overlap=0.5, error_rate=0.2, mismatch_action=None),
               dict(min_overlap=20, error_rate=0.2, mismatch_action="liberal"),
               dict(min_overlap=10, error_rate=0.1, mismatch_action="conservative"),
               dict(min_overlap=0.3, error_rate=0.2, mismatch_action="N")]
    cases = []
    # (the MiSeq-length pairs come last so that the cases before them stay what they were)
    plan = [(cfg, sizes, False) for sizes in (((40, 60), (100, 70), (150, 50), (250, 20)), ((300, 14), (308, 8))) for cfg in configs]
    plan += [(cfg, ((60, 10), (150, 6)), True) for cfg in configs]          # soft-masked reads: the aligner compares characters
    for cfg, sizes, soft in plan:
        for L, count in sizes:
            for _ in range(count):
                r1, q1, r2, q2 = make_pair(rng, L)
                if soft:
                    kind = rng.random()
                    if kind < 0.4 and r1:
                        a = rng.randrange(len(r1)); b = rng.randint(a, len(r1))
                        r1 = r1[:a] + r1[a:b].lower() + r1[b:]
                    if 0.2 < kind < 0.7 and r2:
                        a = rng.randrange(len(r2)); b = rng.randint(a, len(r2))
                        r2 = r2[:a] + r2[a:b].lower() + r2[b:]
                    if kind > 0.85:
                        r1, r2 = r1.lower(), r2.lower()
                with_quals = cfg["mismatch_action"] in ("liberal", "conservative") or rng.random() < 0.7
                insert = rng.random() < 0.15
                a = Sequence("p/1", r1, q1 if with_quals else None)
                b = Sequence("p/2", r2, q2 if with_quals else None)
                a.insert_overlap = b.insert_overlap = insert
                mod = MergeOverlapping(**cfg)
                case = dict(cfg=cfg, r1=r1, q1=q1 if with_quals else None, r2=r2, q2=q2 if with_quals else None,
                            insert=insert)
                try:
                    o1, o2 = mod(a, b)
                    case["out"] = dict(seq1=o1.sequence, qual1=o1.qualities, merged=bool(o1.merged),
                                       read2_none=o2 is None, seq2=None if o2 is None else o2.sequence,
                                       qual2=None if o2 is None else o2.qualities,
                                       corrected=[int(o1.corrected), 0 if o2 is None else int(o2.corrected)],
                                       corrected_pairs=mod.corrected_pairs, corrected_bp=list(mod.corrected_bp))
                except Exception as err:                              # noqa: BLE001 (recorded, not handled)
                    case["error"] = type(err).__name__
                cases.append(case)
    merged = sum(1 for c in cases if c.get("out", {}).get("merged"))
    errors = sum(1 for c in cases if "error" in c)
    out = os.path.join(HERE, "merge_fuzz.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps(cases, sort_keys=True).encode())
    print("wrote", out, os.path.getsize(out), "bytes;", len(cases), "pairs,", merged, "merged,", errors, "errors")


if __name__ == "__main__":
    main()
