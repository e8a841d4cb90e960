"""Shared case generators / checkers for the emulation (CPU) and GPU parity tests."""
import random

from .conftest import load_golden, tup


def rseq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def mutate(rng, s, p, alpha="ACGT"):
    out = []
    for c in s:
        r = rng.random()
        if r < p:
            out.append(rng.choice(alpha))
        elif r < p * 1.3:
            pass
        elif r < p * 1.6:
            out.append(c)
            out.append(rng.choice(alpha))
        else:
            out.append(c)
    return "".join(out)


def planted_reads(rng, ref, count, max_len=200, fixed_len=None):
    reads = []
    for _ in range(count):
        if rng.random() < 0.6:
            a = mutate(rng, ref, rng.choice([0, 0.03, 0.08, 0.15]))
            cut, pos = rng.randint(0, len(a)), rng.randint(0, max(0, max_len - 50))
            w = rng.random()
            q = (rseq(rng, pos) + a + rseq(rng, rng.randint(0, 30)) if w < 0.4 else
                 a[cut:] + rseq(rng, pos) if w < 0.7 else rseq(rng, pos) + a[:cut])
        else:
            q = rseq(rng, rng.randint(0, max_len), "ACGTN")
        if fixed_len is not None:
            q = (q + rseq(rng, fixed_len))[:fixed_len]
        reads.append(q[:max_len])
    return reads


def check_golden_locate(Aligner, unsupported_exc):
    """Every committed locate case through the per-read API; returns (#checked, #unsupported)."""
    cases = load_golden("locate_fuzz.json.gz") + load_golden("kats.json")["locate"]
    checked = unsupported = 0
    for c in cases:
        try:
            al = Aligner(c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        except unsupported_exc:
            unsupported += 1
            continue
        assert al.locate(c["query"]) == tup(c["out"]), c
        checked += 1
    return checked, unsupported


def check_batches_against_oracle(Aligner, oracle, unsupported_exc, seed, rounds, max_m=128):
    """Random aligner settings (all 16 flag sets, every indel regime, wildcard modes),
    ragged and fixed-length batches around the 64-read tile boundary; each result
    record must equal the oracle's tuple for that read."""
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        m = rng.randint(1, max_m) if rng.random() < 0.3 else rng.randint(1, 40)
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.7 else "ACGTNRY")
        flags = rng.randint(0, 15) if rng.random() < 0.5 else rng.choice([14, 11, 8, 2, 15, 9])
        e = rng.choice([0, 0.05, 0.1, 0.12, 0.2, 0.3, 0.5])
        ic = rng.choice([1, 1, 2, 3, 100000])
        mo = rng.choice([1, 3, 5])
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        fixed = rng.randint(0, 200) if rng.random() < 0.3 else None
        reads = planted_reads(rng, ref, rng.choice([1, 63, 64, 65, 130, 200]), 200, fixed)
        got = al.locate_batch(reads).tuples()
        assert len(got) == len(reads)
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, q, e, flags, wr, wq, mo, ic, g)
            total += 1
    return total
