"""Shared case generators / checkers for the emulation (CPU) and GPU parity tests."""
import json
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
        if rng.random() < 0.08:
            e = rng.choice([1.0, 1.5])             # k = int(e * m) >= m: every row is "within k" (round-1 advisor finding)
        ic = rng.choice([1, 1, 2, 3, 100000])
        mo = rng.choice([1, 3, 5])
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        fixed = rng.randint(0, 200) if rng.random() < 0.3 else None
        reads = planted_reads(rng, ref, rng.choice([1, 63, 64, 65, 130, 200]), 200, fixed)
        got = al.locate_batch(reads, path="filtered").tuples()      # filtered pipeline where it applies
        plain = al.locate_batch(reads, path="full").tuples()         # full sweep
        assert len(got) == len(reads) and plain == got
        assert al.locate_batch(reads).tuples() == got                # what a caller gets: short batch -> a wave per read
        assert al.locate_batch(reads, path="wave").tuples() == got
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, q, e, flags, wr, wq, mo, ic, g)
            total += 1
    return total


def check_filtered_pipeline(Aligner, oracle, unsupported_exc, seed, rounds, m_range=(1, 64),
                            flag_choices=(14, 14, 11, 15, 10)):
    """The bit-parallel pre-pass + windowed DP against the oracle on the adapter types it
    applies to (START_WITHIN_SEQ2 and STOP_WITHIN_SEQ2 set, m <= 64): every indel regime,
    wildcard mode and overlap threshold, ragged and fixed-length batches, reads holding the
    adapter several times (first-perfect-hit shortcut)."""
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        m = rng.randint(*m_range)
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.7 else "ACGTNRY")
        flags = rng.choice(flag_choices)
        e = rng.choice([0, 0.05, 0.1, 0.12, 0.2, 0.3, 0.5])
        ic = rng.choice([1, 1, 1, 2, 3, 100000])
        mo = rng.choice([1, 3, 5, 40])
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        fixed = rng.randint(0, 200) if rng.random() < 0.4 else None
        reads = planted_reads(rng, ref, rng.choice([1, 64, 65, 130, 300]), 220, fixed)
        if rng.random() < 0.2:
            reads = [r if rng.random() < 0.5 else ref * 3 for r in reads]
        if m > 32 and fixed is None:
            # the 32-row pre-pass of 33..40-base adapters: a perfect 32-base prefix with a broken
            # tail before the real occurrence, and reads that end inside the tail rows
            bad = ref[:32] + "".join(rng.choice([c for c in "ACGT" if c != x] or "A") for x in ref[32:])
            for cut in range(30, m + 1):
                reads.append(rseq(rng, rng.randint(0, 40)) + ref[:cut])
            reads += [rseq(rng, 20) + bad + rseq(rng, 9) + ref + rseq(rng, 5), rseq(rng, 7) + bad,
                      bad + ref[:33], rseq(rng, 3) + mutate(rng, ref[:32], 0.05) + ref[32:] + rseq(rng, 11)]
        got = al.locate_batch(reads, path="filtered").tuples()
        assert got == al.locate_batch(reads, path="full").tuples()
        assert got == al.locate_batch(reads).tuples() and got == al.locate_batch(reads, path="wave").tuples()
        if fixed and all(len(r) == fixed for r in reads):
            # the same reads as an equal-length batch (no lens array): the pre-pass then bins the
            # partial overlaps by row count and the window DP sweeps a triangle only
            import numpy as np
            mat = np.frombuffer("".join(reads).encode(), np.uint8).reshape(len(reads), fixed).copy()
            assert al.locate_batch(al.pack(mat), path="filtered").tuples() == got
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, q, e, flags, wr, wq, mo, ic, g)
            total += 1
    return total


def piece_reads(rng, ref, n, count, e):
    """Equal-length reads (n bases) that press on every rule of the two-pass pre-pass (piece_core.hpp): the
    adapter at every distance from the read end, whole and cut, with 0 .. k + 1 edits placed so that they spare
    one piece, kill several or sit between two pieces; single pieces without the rest (chance hits); two
    occurrences far apart (windows of more than PIECE_NARROW columns); low-complexity reads and N runs."""
    m, reads = len(ref), []
    k = int(e * m)

    def edited(s, edits):
        s = list(s)
        for _ in range(edits):
            if not s:
                break
            i, kind = rng.randrange(len(s)), rng.random()
            if kind < 0.5:
                s[i] = rng.choice("ACGT")
            elif kind < 0.75:
                del s[i]
            else:
                s.insert(i, rng.choice("ACGT"))
        return "".join(s)

    def fit(q):
        return (q + rseq(rng, n))[:n] if len(q) < n else q[:n]

    for _ in range(count):
        w = rng.random()
        if w < 0.25:                                      # the adapter (edited) ending anywhere, incl. cut by the read end
            a = edited(ref, rng.randint(0, k + 1))
            pos = rng.randint(0, n)
            q = rseq(rng, pos) + a + rseq(rng, n)
        elif w < 0.45:                                    # partial adapter exactly at the read end, every overlap length
            cut = rng.randint(1, m)
            a = edited(ref[:cut], rng.choice([0, 0, 1, 1, 2, 3]))
            q = rseq(rng, max(0, n - len(a))) + a
            q = q[len(q) - n:] if len(q) > n else q
        elif w < 0.55:                                    # one piece only
            L = rng.choice([5, 6, 7, 8])
            i = rng.randrange(0, max(1, m - L))
            pos = rng.randint(0, n)
            q = rseq(rng, pos) + ref[i:i + L] + rseq(rng, n)
        elif w < 0.7:                                     # two occurrences: the first broken or not, the second somewhere else
            a1 = edited(ref, rng.randint(0, k + 2))
            a2 = edited(ref[:rng.randint(3, m)], rng.randint(0, 2))
            q = rseq(rng, rng.randint(0, 30)) + a1 + rseq(rng, rng.randint(0, 60)) + a2
        elif w < 0.8:
            q = rng.choice(["A", "C", "G", "T", "N", "AC", ref[:3], ref[:5] + "N"]) * n
        elif w < 0.9:                                     # soft edits of the whole read
            q = mutate(rng, rseq(rng, rng.randint(0, n)) + ref, 0.04, "ACGTN")
        else:
            q = rseq(rng, n, "ACGTN" if rng.random() < 0.3 else "ACGT")
        reads.append(fit(q))
    return reads


def check_piece_pipeline(Aligner, oracle, unsupported_exc, seed, rounds, count=200,
                         lengths=(70, 100, 128, 150, 160, 180, 200, 224, 250, 260, 288, 300),      # (3 .. 10 plane words)
                         mrange=(20, 40), flag_choices=(14, 14, 14, 10)):
    """The two-pass pre-pass on plane64 batches (atr_locate_planes_batch) against the other kernel families and the
    oracle, on aligners inside its envelope (and a few outside: pack(layout="plane64") must refuse those)."""
    import numpy as np
    rng = random.Random(seed)
    total = refused = 0
    for _ in range(rounds):
        m = rng.randint(*mrange)
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.9 else "ACGTN")
        flags = rng.choice(list(flag_choices))
        e = rng.choice([0, 0.03, 0.05, 0.08, 0.1, 0.1, 0.1, 0.12])
        ic = rng.choice([1, 1, 1, 2, 100000])
        mo = rng.choice([1, 3, 3, 5, 12, 25])
        wr, wq = rng.random() < 0.2, rng.random() < 0.25
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        n = rng.choice(lengths)
        if rng.random() < 0.4:
            # a ragged batch (the pre-pass moves every read to the end of its words): the same kinds of reads at
            # several lengths, from empty to n, shuffled
            reads = piece_reads(rng, ref, n, count // 4, e)
            for _ in range(5):
                ni = rng.choice([rng.randint(0, 8), rng.randint(0, m + 4), rng.randint(0, n), rng.randint(max(0, n - 40), n), n - 1, n - 32])
                reads += piece_reads(rng, ref, max(0, ni), count // 6, e)
            rng.shuffle(reads)
            mat = reads
        else:
            reads = piece_reads(rng, ref, n, count, e)
            mat = np.frombuffer("".join(reads).encode(), np.uint8).reshape(len(reads), n).copy()
        try:
            planes = al.pack(mat, layout="plane64")
        except unsupported_exc:
            refused += 1
            continue
        assert planes.layout == "plane64" and (planes.lens is None) == (len(set(map(len, reads))) == 1)
        got = al.locate_batch(planes).tuples()
        tiles = al.pack(mat, layout="tile64")
        assert got == al.locate_batch(tiles, path="full").tuples()
        assert got == al.locate_batch(tiles, path="filtered").tuples()
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, q, e, flags, wr, wq, mo, ic, g)
            total += 1
    return total, refused


def check_certificates(Aligner, oracle, unsupported_exc, seed, rounds, count=500, mrange=(20, 40)):
    """The DP-free decisions of the pre-pass (filter_core.hpp: perfect-overlap and single-substitution certificates) under
    pressure: adapters that are random, low-complexity, periodic (with a few defects), the TruSeq prefix, or a repeated
    half; the adapter in the read with one or two substitutions (any row, the first and last three more often), flanks
    that are random, end in a prefix of the adapter, start with its suffix, continue its period or repeat one base; now
    and then an indel on top.  Every record of the two-pass pipeline against the oracle."""
    rng = random.Random(seed)
    total = 0

    def adapter(kind, m):
        if kind == 0:
            return rseq(rng, m, "ACGT")
        if kind == 1:
            return rseq(rng, m, rng.choice(["AC", "AG", "CT", "ACG"]))
        if kind == 2:
            unit = rseq(rng, rng.randint(1, 5), "ACGT")
            s = list((unit * 80)[:m])
            for _ in range(rng.randint(0, 3)):
                s[rng.randrange(m)] = rng.choice("ACGT")
            return "".join(s)
        if kind == 3:
            return "AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"[:m] if m <= 34 else rseq(rng, m, "ACGT")
        half = rseq(rng, m // 2 + 1, "ACGT")
        return (half + half)[:m]

    for _ in range(rounds):
        m = rng.randint(*mrange)
        ref = adapter(rng.randrange(5), m)
        e = rng.choice([0.03, 0.05, 0.08, 0.1, 0.1, 0.12])
        mo = rng.choice([1, 3, 5])
        flags = rng.choice([14, 14, 10])
        al = Aligner(ref, e, flags, False, False, mo, 1)
        n = rng.choice([100, 128, 150, 160])
        reads = []
        for _ in range(count):
            a = list(ref)
            for _ in range(1 if rng.random() < 0.8 else 2):
                p = rng.randrange(m) if rng.random() < 0.7 else rng.choice([0, 1, 2, m - 1, m - 2, m - 3])
                a[p] = rng.choice([c for c in "ACGT" if c != ref[p]] if rng.random() < 0.9 else "ACGTN")
            a = "".join(a)
            w, v, x = rng.random(), rng.random(), rng.randint(1, 6)
            fb = (rseq(rng, 12) if w < 0.4 else rseq(rng, 12 - x) + ref[:x] if w < 0.6 else
                  rseq(rng, 8) + ref[rng.randint(0, 3):][:4] if w < 0.7 else (ref * 2)[rng.randint(0, m):][:12] if w < 0.85 else ref[0] * 12)
            fa = (rseq(rng, 12) if v < 0.4 else ref[-x:] + rseq(rng, 12) if v < 0.6 else
                  (ref * 2)[rng.randint(0, m):][:12] if v < 0.8 else ref[-1] * 12)
            pos = rng.randint(0, n - m)
            q = ((rseq(rng, pos) + fb)[-pos:] if pos else "") + a + fa + rseq(rng, n)
            if rng.random() < 0.15:                       # an indel near by as well
                i = rng.randrange(n)
                q = q[:i] + q[i + 1:] if rng.random() < 0.5 else q[:i] + rng.choice("ACGT") + q[i:]
            if rng.random() < 0.3:
                # a PERFECT partial adapter at the read end (the overlap certificates, one- and two-word sweeps): its first i
                # bases verbatim, behind a flank that is random, continues the adapter's period backwards, repeats the
                # adapter's start or one base -- now and then with a substitution inside the overlap
                i = rng.randint(1, m - 1)
                u = rng.random()
                flank = (rseq(rng, 20) if u < 0.4 else (ref * 3)[m - 20 + rng.randint(0, 3):][:20] if u < 0.6 else
                         (ref[:rng.randint(1, 8)] * 20)[:20] if u < 0.8 else ref[rng.randrange(m)] * 20)
                tail = list(ref[:i])
                if rng.random() < 0.2:
                    tail[rng.randrange(i)] = rng.choice("ACGT")
                q = (rseq(rng, n) + flank + "".join(tail))[-n:]
            reads.append(q[:n])
        try:
            planes = al.pack(reads, layout="plane64")
        except unsupported_exc:
            continue
        got = al.locate_batch(planes).tuples()
        for q, g in zip(reads, got):
            exp = oracle.locate(ref, q, e, flags, False, False, mo, 1)
            assert g == exp, (ref, q, e, flags, mo, g, exp)
            total += 1
    return total


def check_uniform_partial_overlaps(Aligner, oracle, unsupported_exc, seed, rounds, count=400,
                                   flag_choices=(14, 14, 15, 10), short=False):
    """Equal-length batches full of partial adapter occurrences at the read end (the row-binned
    triangle sweep of the window DP) next to whole and absent adapters."""
    import numpy as np
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        m = rng.randint(8, 64)
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.8 else "ACGTN")
        flags = rng.choice(flag_choices)
        e = rng.choice([0.05, 0.1, 0.1, 0.2, 0.3])
        ic = rng.choice([1, 1, 2, 100000])
        mo = rng.choice([1, 3, 5])
        wr, wq = rng.random() < 0.2, rng.random() < 0.2
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        n = rng.randint(m // 2, m + 8 if short else 160)
        reads = []
        for _ in range(count):
            w = rng.random()
            if w < 0.7:
                part = mutate(rng, ref, rng.choice([0, 0.03, 0.08, 0.15]))[:rng.randint(1, m)]
                q = rseq(rng, n) + part
                q = q[len(q) - n:] if rng.random() < 0.8 else (q + rseq(rng, rng.randint(1, 6)))[-n:]
            elif w < 0.85:
                q = (rseq(rng, rng.randint(0, n)) + mutate(rng, ref, 0.05) + rseq(rng, n))[:n]
            else:
                q = rseq(rng, n, "ACGTN")
            reads.append(q)
        mat = np.frombuffer("".join(reads).encode(), np.uint8).reshape(count, n).copy()
        got = al.locate_batch(al.pack(mat), path="filtered").tuples()
        assert got == al.locate_batch(al.pack(mat), path="full").tuples()
        assert got == al.locate_batch(al.pack(mat), path="wave").tuples()
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, q, e, flags, wr, wq, mo, ic, g)
            total += 1
    return total


def norm_insert(res):
    """(tuple, Match|None, Match|None) | None -> the JSON shape of the golden files."""
    if res is None:
        return None

    def f(m):
        return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    return [list(res[0]), f(res[1]), f(res[2])]


def check_golden_insert(InsertAligner):
    kats = load_golden("kats.json")["match_insert"]
    # insert_long: 2 x 257 .. 320 bp; insert_longer: 2 x 321 .. 600 bp (beyond the insert kernel: InsertAligner._match_insert_long)
    cases = kats + load_golden("insert_fuzz.json.gz") + load_golden("insert_long.json.gz") + load_golden("insert_longer.json.gz")
    cache = {}
    for c in cases:
        key = (c["a1"], c["a2"], repr(sorted(c["kw"].items())))
        if key not in cache:
            cache[key] = InsertAligner(c["a1"], c["a2"], **c["kw"])
        assert norm_insert(cache[key].match_insert(c["r1"], c["r2"])) == c["out"], c
    # the same cases again as batches (one per configuration, ragged lengths)
    by_cfg = {}
    for c in cases:
        by_cfg.setdefault((c["a1"], c["a2"], repr(sorted(c["kw"].items()))), []).append(c)
    for key, cs in by_cfg.items():
        res = cache[key].match_insert_batch([c["r1"] for c in cs], [c["r2"] for c in cs]).results()
        assert [norm_insert(r) for r in res] == [c["out"] for c in cs]
    return len(cases)


def check_golden_multi_compare(align):
    kats = load_golden("kats.json")
    n = 0
    for c in kats["multi_locate"] + load_golden("multi_fuzz.json.gz"):
        got = align.MultiAligner(c["e"], c["flags"], c["mo"]).locate(c["ref"], c["query"], c["mx"])
        assert got == (None if c["out"] is None else [tup(x) for x in c["out"]]), c
        n += 1
    for c in load_golden("prefix_fuzz.json.gz"):
        assert align.compare_prefixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["prefix"]), c
        assert align.compare_suffixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["suffix"]), c
        n += 1
    for c in kats["compare_prefixes"]:
        assert align.compare_prefixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["out"]), c
    for c in kats["compare_suffixes"]:
        assert align.compare_suffixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["out"]), c
    return n


def check_insert_batches_against_oracle(InsertAligner, oracle, seed, rounds):
    """Random insert-aligner settings, batches of synthetic pairs (with and without a
    true overlap, ragged lengths), every result against the oracle."""
    from atropos_amd import synth
    from atropos_amd.util import reverse_complement
    rng = random.Random(seed)
    cfgs = [dict(), dict(max_insert_mismatch_frac=0.1, max_adapter_mismatch_frac=0.1), dict(read_wildcards=True),
            dict(adapter_wildcards=False), dict(adapter_wildcards=False, read_wildcards=True),
            dict(min_insert_overlap=5, min_adapter_overlap=3), dict(insert_max_rmp=1e-3, adapter_max_rmp=1e-2),
            dict(adapter_check_cutoff=3), dict(base_probs=dict(match_prob=0.33, mismatch_prob=0.67))]
    total = 0
    for it in range(rounds):
        cfg = cfgs[it % len(cfgs)]
        a1, a2 = synth.PE_ADAPTER1, synth.PE_ADAPTER2
        if it % 2 and cfg.get("adapter_wildcards", True):
            a1 = a1[:20].replace("G", "N", 1)
        if it % 5 == 3:                        # adapters of more than 64 bases (indexed TruSeq adapters are 66): two halves
            a1, a2 = a1 + rseq(rng, rng.choice([2, 13, 64 - 0])), a2 + rseq(rng, rng.choice([1, 30, 60]))
            a1, a2 = a1[:128], a2[:128]
        ia = InsertAligner(a1, a2, **cfg)
        orc = oracle.InsertOracle(a1, a2, **cfg)
        n = rng.choice([20, 50, 100, 150, 250, 257, 290, 300, 320])
        r1s, r2s = [], []
        for _ in range(rng.choice([1, 64, 65, 130])):
            f = rng.randint(0, int(1.6 * n))
            F = rseq(rng, f)
            r1 = (F + a1.replace("N", "A") + rseq(rng, n))[:rng.choice([n, n, n, n - 3])]
            r2 = (reverse_complement(F) + a2 + rseq(rng, n))[:rng.choice([n, n, n - 1])]
            p = rng.choice([0, 0.01, 0.03, 0.1])
            r1 = "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in r1)
            r2 = "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in r2)
            if rng.random() < 0.1:
                r1 = rseq(rng, n)
            if rng.random() < 0.05:
                r1 = "A" * len(r1)
                r2 = "T" * len(r2)          # low complexity: many hits
            soft = rng.random()              # soft-masked reads: characters compare as they are (_align.pyx:690)
            if soft < 0.04 and r1:
                at = rng.randrange(len(r1))
                r1 = r1[:at] + r1[at:].lower()
            elif soft < 0.07:
                r1, r2 = r1.lower(), r2.lower()
            elif soft < 0.1 and r2:
                at = rng.randrange(len(r2))
                r2 = r2[:at] + r2[at:at + 9].lower() + r2[at + 9:]
            r1s.append(r1)
            r2s.append(r2)
        got = ia.match_insert_batch(r1s, r2s).results()
        for x, y, g in zip(r1s, r2s, got):
            exp = orc.match_insert(x, y)
            exp = None if exp is None else [list(exp[0]), None if exp[1] is None else list(exp[1]),
                                            None if exp[2] is None else list(exp[2])]
            assert norm_insert(g) == exp, (cfg, x, y, norm_insert(g), exp)
            total += 1
    return total


def read_state(r):
    m = r.match if (r.match is not None and hasattr(r.match, "astart")) else None
    return dict(seq=r.sequence, qual=r.qualities, corrected=r.corrected, overlap=bool(r.insert_overlap),
                match=None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors],
                adapter=None if m is None else m.adapter.name, n_info=len(r.match_info) if r.match_info else 0)


def check_cutter_golden(limit=None):
    """AdapterCutter (per read and batched) against the reference's outputs."""
    from atropos_amd.adapters import Adapter
    from atropos_amd.modifiers import AdapterCutter
    from atropos_amd.reads import Sequence
    cases = load_golden("cutter_fuzz.json.gz")
    if limit:
        cases = cases[:limit]
    n = 0
    for c in cases:
        def mk():
            ads = [Adapter(sp["seq"], sp["where"], name=sp["name"], **c["kw"]) for sp in c["specs"]]
            return AdapterCutter(ads, times=c["times"], action=c["action"])
        cutter = mk()
        got = [read_state(cutter(Sequence("r", q, qual))) for q, qual in c["reads"]]
        assert got == c["out"], (c["specs"], c["kw"], c["times"], c["action"])
        assert cutter.with_adapters == c["with_adapters"]
        cutter = mk()
        got = [read_state(r) for r in cutter.call_batch([Sequence("r", q, qual) for q, qual in c["reads"]])]
        assert got == c["out"], ("batch", c["specs"], c["kw"], c["times"], c["action"])
        assert cutter.with_adapters == c["with_adapters"]
        n += len(c["reads"])
    return n


def check_insert_cutter_golden(limit=None):
    from atropos_amd import synth
    from atropos_amd.adapters import Adapter, BACK
    from atropos_amd.modifiers import InsertAdapterCutter
    from atropos_amd.reads import Sequence
    from atropos_amd.util import RandomMatchProbability
    cases = load_golden("insert_cutter_fuzz.json.gz")
    if limit:
        cases = cases[:limit]
    rmp = RandomMatchProbability()
    n = 0
    for c in cases:
        def mk():
            kw = dict(max_error_rate=0.2, min_overlap=1, indel_cost=3, match_probability=rmp, max_rmp=1e-6)
            return InsertAdapterCutter(Adapter(synth.PE_ADAPTER1, BACK, name="a1", **kw),
                                       Adapter(synth.PE_ADAPTER2, BACK, name="a2", **kw),
                                       action=c["trim_action"], **c["kw"])
        cutter = mk()
        got = []
        for r1, q1, r2, q2 in c["pairs"]:
            a, b = cutter(Sequence("p", r1, q1), Sequence("p", r2, q2))
            got.append([read_state(a), read_state(b)])
        assert got == c["out"], (c["kw"], c["trim_action"])
        assert list(cutter.with_adapters) == c["with_adapters"]
        assert cutter.corrected_pairs == c["corrected_pairs"] and list(cutter.corrected_bp) == c["corrected_bp"]
        cutter = mk()
        res = cutter.call_batch([Sequence("p", p[0], p[1]) for p in c["pairs"]],
                                [Sequence("p", p[2], p[3]) for p in c["pairs"]])
        assert [[read_state(a), read_state(b)] for a, b in res] == c["out"], ("batch", c["kw"], c["trim_action"])
        assert list(cutter.with_adapters) == c["with_adapters"]
        assert cutter.corrected_pairs == c["corrected_pairs"] and list(cutter.corrected_bp) == c["corrected_bp"]
        n += len(c["pairs"])
    return n


def check_caller_kats():
    from atropos_amd import synth
    from atropos_amd.adapters import Adapter, AdapterParser, BACK
    from atropos_amd.align import InsertAligner
    from atropos_amd.modifiers import AdapterCutter, ErrorCorrectorMixin, InsertAdapterCutter
    from atropos_amd.reads import Sequence
    k = load_golden("caller_kats.json")
    c = k["mismatched_adapter_overlaps"]
    parser = AdapterParser()
    cutter = InsertAdapterCutter(parser.parse_from_spec(c["a1"]), parser.parse_from_spec(c["a2"]),
                                 max_insert_mismatch_frac=0.3, max_adapter_mismatch_frac=0.3)
    n1, n2 = cutter(Sequence('foo', c["r1"], '#' * 125), Sequence('foo', c["r2"], '#' * 125))
    assert len(n1) == 91 and len(n2) == 91
    for got, exp in zip((n1, n2), c["out"]):
        g = read_state(got)
        g["adapter"] = exp["adapter"]               # auto-generated adapter names are process-global counters
        assert g == exp
    i = k["unequal_lengths_in"]
    e1, e2 = Sequence('read1', i["s1"], i["q1"]), Sequence('read1', i["s2"], i["q2"])
    im, _, _ = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2).match_insert(e1.sequence, e2.sequence)
    assert list(im) == k["unequal_lengths_out"]["insert"]
    ec = ErrorCorrectorMixin('N')
    ec.correct_errors(e1, e2, im, truncate_seqs=True)
    assert e1.corrected == 3 and e2.corrected == 3
    for pos in (80, 86, 104):
        assert e1.sequence[pos] == 'N' and e2.sequence[104 - pos] == 'N'
    assert read_state(e1) == k["unequal_lengths_out"]["r1"] and read_state(e2) == k["unequal_lengths_out"]["r2"]
    ad = Adapter('CCCC', BACK, 0.1)
    t = AdapterCutter([ad], times=3)(Sequence('name', 'AAAACCCCAAAA'))
    g = read_state(t)
    g["adapter"] = k["statistics"]["out"]["adapter"]
    assert g == k["statistics"]["out"]
    assert {str(a): b for a, b in ad.lengths_back.items()} == k["statistics"]["lengths_back"]


def check_match_to_golden():
    from atropos_amd.adapters import Adapter
    from atropos_amd.reads import Sequence
    from atropos_amd.util import RandomMatchProbability
    rmp = RandomMatchProbability()
    cases = load_golden("match_to_fuzz.json.gz") + load_golden("kats.json")["match_to"]

    def f(m):
        return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    for c in cases:
        use = c.get("use_rmp", False)
        ad = Adapter(c["seq"], c["where"], match_probability=rmp if use else None, max_rmp=1e-6 if use else None,
                     **c["kw"])
        assert f(ad.match_to(Sequence("r", c["read"]))) == c["out"], c
        assert f(ad.match_to_batch([Sequence("r", c["read"])])[0]) == c["out"], ("batch", c)
    return len(cases)


def check_linked_c4():
    from atropos_amd import synth
    from atropos_amd.adapters import LinkedAdapter
    from atropos_amd.reads import Sequence
    heads = load_golden("synth_heads.json.gz")["C4"]
    w = synth.workload("C4", 0, heads["count"])
    reads = [Sequence("r", bytes(x.tolist()).decode("ascii")) for x in w["reads"]]
    linked = [LinkedAdapter(fr, bk, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                            min_overlap=w["min_overlap"], indel_cost=w["indel_cost"])
              for fr, bk in zip(w["fronts"], w["backs"])]

    def f(m):
        return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    per = [la.match_to_batch(reads) for la in linked]
    for i in range(len(reads)):
        got = [None if per[a][i] is None else [f(per[a][i].front_match), f(per[a][i].back_match)] for a in range(4)]
        assert got == heads["out"][i]
    for i in range(0, len(reads), 7):                # the per-read path on a sample
        lm = linked[i % 4].match_to(reads[i])
        got = None if lm is None else [f(lm.front_match), f(lm.back_match)]
        assert got == heads["out"][i][i % 4]
    return len(reads)


# ---------------------------------------------------------------------------------------------
# fused linked-adapter pipeline (atr_linked_match_batch) against the oracle
def oracle_match_to(oracle, seq, flags, read, e, min_overlap, indel_cost, adapter_wildcards, read_wildcards):
    """Adapter.match_to (reference adapters/__init__.py:338-400) restated on top of the oracle's
    locate, for adapters WITH indels and no RMP filter: upper-case, literal shortcut unless the
    adapter has wildcards, alignment, acceptance test.  Returns the 6-tuple or None."""
    read = read.upper()
    m = len(seq)
    if not adapter_wildcards:
        pos = (0 if read.startswith(seq) else -1) if flags == 8 else read.find(seq)
        if pos >= 0:
            return (0, m, pos, pos + m, m, 0)
    al = oracle.locate(seq, read, e, flags, adapter_wildcards, read_wildcards, min(min_overlap, m), indel_cost)
    if al is None:
        return None
    size = al[1] - al[0]
    return al if (size >= min(min_overlap, m) and al[5] / size <= e) else None


def oracle_linked(oracle, fronts, backs, read, kw):
    """AdapterCutter._best_match over LinkedAdapter.match_to for one read: (which, count, front, back)."""
    which, count, fm, bm = -1, 0, None, None
    for a, (fs, bs) in enumerate(zip(fronts, backs)):
        aw_f, aw_b = kw["adapter_wildcards"] and not set(fs) <= set("ACGT"), kw["adapter_wildcards"] and not set(bs) <= set("ACGT")
        f = oracle_match_to(oracle, fs, 8, read, kw["e"], kw["min_overlap"], kw["indel_cost"], aw_f, kw["read_wildcards"])
        if f is None:
            continue
        count += 1
        if which < 0:
            which, fm = a, f
            bm = oracle_match_to(oracle, bs, 14, read[f[3]:], kw["e"], kw["min_overlap"], kw["indel_cost"], aw_b,
                                 kw["read_wildcards"])
    return which, count, fm, bm


def check_linked_sets_against_oracle(oracle, seed, rounds, reads_per_round=(1, 64, 65, 200)):
    """Random sets of 1..4 linked adapters (5' parts of 6..26 bases, 3' parts of 8..64: the one-word,
    NARROW and two-word forms of the pre-pass), error rates, indel costs, wildcard modes, ragged and
    equal-length batches; reads carry the 5' adapter (exact, mutated, shifted, absent, or TWO
    adapters' worth) and the matching, a wrong or no 3' adapter.  Every read's (which, count, front,
    back) through the fused pipeline must equal the reference's rule restated on the oracle."""
    import numpy as np
    import torch
    from atropos_amd import _lib
    from atropos_amd.adapters import LinkedAdapter, LinkedSet, AsciiSource, upper_ascii
    be = _lib.get_backend()
    rng = random.Random(seed)
    total = fused_sets = 0
    grouped_sets = [0]
    check_linked_sets_against_oracle.grouped_sets = grouped_sets
    for it in range(rounds):
        na = rng.randint(1, 4)
        e = rng.choice([0.05, 0.1, 0.12, 0.2, 0.3])
        with_n = rng.random() < 0.2                  # every adapter of the set holds an N, or none does -- mostly
        alpha_f = "ACGTN" if with_n else "ACGT"
        def force_n(x):
            if not with_n or "N" in x or rng.random() < 0.1:
                return x
            p = rng.randrange(len(x))
            return x[:p] + "N" + x[p + 1:]
        kw = dict(e=e, min_overlap=rng.choice([1, 3, 5]), indel_cost=rng.choice([1, 1, 1, 2, 3]),
                  adapter_wildcards=True, read_wildcards=rng.random() < 0.2)
        same_len = rng.random() < 0.6
        top = 26 if rng.random() < 0.15 else min(26, int(31 / (1 + e)), int(7.9 / e))      # mostly inside the fused envelope
        fl = rng.randint(6, top)
        fronts = [force_n(rseq(rng, fl if same_len else rng.randint(6, top), alpha_f)) for _ in range(na)]
        style = rng.choice(["short", "narrow", "wide", "mixed"])
        def blen():
            return {"short": rng.randint(8, 32), "narrow": rng.randint(33, 40), "wide": rng.randint(41, 64),
                    "mixed": rng.randint(8, 64)}[style]
        backs = [force_n(rseq(rng, blen(), alpha_f)) for _ in range(na)]
        if kw["read_wildcards"] and any(set(x) <= set("ACGT") for x in fronts + backs):
            # ACGT-only adapter + read wildcards: literal shortcut on a wildcard compare -> not fused; keep a few
            if rng.random() < 0.7:
                kw["read_wildcards"] = False
        las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=e, min_overlap=kw["min_overlap"],
                             indel_cost=kw["indel_cost"], read_wildcards=kw["read_wildcards"]) for f, b in zip(fronts, backs)]
        lset = LinkedSet(las)
        fused_sets += int(lset.fused)
        n = rng.randint(30, 170)
        fixed = rng.random() < 0.5
        reads = []
        for _ in range(rng.choice(reads_per_round)):
            a = rng.randrange(na)
            w = rng.random()
            head = (fronts[a] if w < 0.35 else mutate(rng, fronts[a], rng.choice([0.03, 0.08, 0.15])) if w < 0.7 else
                    rseq(rng, rng.randint(0, 3)) + fronts[a] if w < 0.75 else
                    fronts[a] + fronts[(a + 1) % na] if w < 0.8 else "")
            head = head.replace("N", "A")
            frag = rseq(rng, rng.randint(0, n), "ACGT" if rng.random() < 0.9 else "ACGTN")
            b = backs[a if rng.random() < 0.85 else rng.randrange(na)].replace("N", "C")
            v = rng.random()
            tail = (b if v < 0.3 else mutate(rng, b, rng.choice([0.03, 0.08, 0.15])) if v < 0.6 else
                    b[:rng.randint(1, len(b))] if v < 0.8 else "")
            q = head + frag + tail + (rseq(rng, rng.randint(0, 20)) if rng.random() < 0.5 else "")
            if fixed:
                q = (q + rseq(rng, n))[:n]
            q = q[:rng.choice([n, n, len(head) + rng.randint(0, 3)])] if not fixed and rng.random() < 0.1 else q[:n + 40]
            if rng.random() < 0.1:
                q = q.lower()
            reads.append(q if q else "A")
        width = max(len(r) for r in reads)
        mat = np.zeros((len(reads), width), dtype=np.uint8)
        for i, r in enumerate(reads):
            mat[i, :len(r)] = np.frombuffer(r.encode(), dtype=np.uint8)
        a_t = upper_ascii(torch.from_numpy(mat).to(be.device))
        ragged = not (fixed and all(len(r) == len(reads[0]) for r in reads))
        l_t = torch.tensor([len(r) for r in reads], dtype=torch.int32, device=be.device) if ragged else None
        which, count, front, back = lset.match_source(AsciiSource(a_t, l_t))
        if lset.fused and 65 <= width <= 256 and hasattr(lset, "group_applies") and lset.group_applies(width):
            # round 6: the grouped form (5' parts at pack time, a plane64 sub-batch per adapter) on the same reads
            gw, gc, gf_, gb_ = lset.match_groups(lset.pack_groups(a_t, l_t, width))
            assert torch.equal(gw, which) and torch.equal(gc, count), (fronts, backs, kw)
            assert torch.equal(gf_[:, :6], front[:, :6]) and torch.equal(gb_[:, :6], back[:, :6]), (fronts, backs, kw)
            grouped_sets[0] += 1
        which, count = which.cpu().numpy(), count.cpu().numpy()
        front, back = front.cpu().numpy(), back.cpu().numpy()
        for i, q in enumerate(reads):
            ew, ec, ef, eb = oracle_linked(oracle, fronts, backs, q, kw)
            gf = None if front[i, 1] < 0 else tuple(int(v) for v in front[i, :6])
            gb = None if back[i, 1] < 0 else tuple(int(v) for v in back[i, :6])
            assert (int(which[i]), int(count[i]), gf, gb) == (ew, ec, ef, eb), (
                fronts, backs, kw, q, lset.fused, (int(which[i]), int(count[i]), gf, gb), (ew, ec, ef, eb))
            total += 1
    return total, fused_sets


def check_linked_golden():
    """LinkedSet (the fused pipeline where the set is inside its envelope, the step-wise device path
    otherwise) and LinkedAdapter.match_to_batch against what the reference returned (linked_fuzz.json.gz)."""
    import numpy as np
    import torch
    from atropos_amd import _lib
    from atropos_amd.adapters import LinkedAdapter, LinkedSet, AsciiSource, upper_ascii
    from atropos_amd.reads import Sequence
    be = _lib.get_backend()
    total = fused = 0

    def f(m):
        return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    for c in load_golden("linked_fuzz.json.gz"):
        las = [LinkedAdapter(fr, bk, front_anchored=True, back_anchored=False, **c["kw"]) for fr, bk in zip(c["fronts"], c["backs"])]
        lset = LinkedSet(las)
        fused += int(lset.fused)
        reads = c["reads"]
        width = max(len(r) for r in reads)
        mat = np.zeros((len(reads), width), dtype=np.uint8)
        for i, r in enumerate(reads):
            mat[i, :len(r)] = np.frombuffer(r.encode(), dtype=np.uint8)
        a_t = upper_ascii(torch.from_numpy(mat).to(be.device))
        l_t = torch.tensor([len(r) for r in reads], dtype=torch.int32, device=be.device)
        which, count, front, back = lset.match_source(AsciiSource(a_t, l_t))
        per = [la.match_to_batch([Sequence("r", r) for r in reads]) for la in las]
        for i, (ew, ec, ef, eb) in enumerate(c["out"]):
            got = [int(which[i]), int(count[i]), None if front[i, 1] < 0 else [int(v) for v in front[i, :6]],
                   None if back[i, 1] < 0 else [int(v) for v in back[i, :6]]]
            assert got == [ew, ec, ef, eb], (c["fronts"], c["backs"], c["kw"], reads[i], lset.fused, got, [ew, ec, ef, eb])
            hits = [a for a in range(len(las)) if per[a][i] is not None]               # the object path, adapter by adapter
            assert (hits[0] if hits else -1) == ew and len(hits) == ec
            if hits:
                assert [f(per[ew][i].front_match), f(per[ew][i].back_match)] == [ef, eb]
            total += 1
    return total, fused


def check_info_records():
    """Match.get_info_record through AdapterCutter (per read and batched) against the reference's
    MatchInfo rows (info_records.json.gz; cf. the reference's tests/cut/*.info.txt)."""
    from atropos_amd.adapters import Adapter
    from atropos_amd.modifiers import AdapterCutter
    from atropos_amd.reads import Sequence
    rows = 0
    for c in load_golden("info_records.json.gz"):
        def mk():
            return AdapterCutter([Adapter(sp["seq"], sp["where"], name=sp["name"], **c["kw"]) for sp in c["specs"]],
                                 times=c["times"], action=c["action"])
        for batched in (False, True):
            cutter = mk()
            reads = [Sequence(name, q, qual) for name, q, qual in c["reads"]]
            outs = cutter.call_batch(reads) if batched else [cutter(r) for r in reads]
            got = [None if not r.match_info else [list(info) for info in r.match_info] for r in outs]
            assert got == c["out"], (c["specs"], c["kw"], c["times"], c["action"], batched)
        rows += sum(len(o) for o in c["out"] if o)
    return rows


def _c5_states(pairs):
    def f(m):
        return None if m is None or not hasattr(m, "astart") else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    return [[[x.sequence, x.qualities, int(x.corrected), bool(x.insert_overlap), f(x.match)] for x in pair] for pair in pairs]


def _digest(state):
    import hashlib
    return hashlib.sha256(repr(state).encode()).hexdigest()[:16]


def check_c5_head(count=None, text_pipeline=True):
    """The head of BASELINE config C5 (2 x 250 bp with qualities, read wildcards, liberal error
    correction) through InsertAdapterCutter.call_batch and -- as FASTQ text -- through the device
    pipeline (atr_insert_match_batch + atr_insert_plan_batch with in-place correction), against the
    reference's outputs (c5_head.json.gz: every pair by digest, the first pairs in full)."""
    from atropos_amd import synth
    from atropos_amd.adapters import Adapter, BACK
    from atropos_amd.modifiers import InsertAdapterCutter
    from atropos_amd.reads import Sequence
    from atropos_amd.util import RandomMatchProbability
    g = load_golden("c5_head.json.gz")
    n = g["count"] if count is None else min(count, g["count"])
    w = synth.workload("C5", 0, n)

    def rows(t):
        return [bytes(x.tolist()).decode("ascii") for x in t]
    r1, q1, r2, q2 = rows(w["reads1"]), rows(w["quals1"]), rows(w["reads2"]), rows(w["quals2"])
    akw = dict(g["adapter_kw"], match_probability=RandomMatchProbability())
    cutter = InsertAdapterCutter(Adapter(synth.PE_ADAPTER1, BACK, name="a1", **akw), Adapter(synth.PE_ADAPTER2, BACK, name="a2", **akw),
                                 action='trim', mismatch_action='liberal', read_wildcards=True)
    out = cutter.call_batch([Sequence("p", a, b) for a, b in zip(r1, q1)], [Sequence("p", a, b) for a, b in zip(r2, q2)])
    states = _c5_states(out)
    for k in range(min(n, len(g["full"]))):
        assert states[k] == g["full"][k], (k, states[k], g["full"][k])
    assert [_digest(s) for s in states] == g["digests"][:n]
    if n == g["count"]:
        assert list(cutter.with_adapters) == g["with_adapters"] and cutter.corrected_pairs == g["corrected_pairs"]
        assert list(cutter.corrected_bp) == g["corrected_bp"]
    if text_pipeline:
        from atropos_amd.trim import pipeline_from_args
        fq = lambda seqs, quals: "".join("@p%d\n%s\n+\n%s\n" % (k, s, q) for k, (s, q) in enumerate(zip(seqs, quals))).encode()
        pipe = pipeline_from_args("-a %s -A %s --aligner insert --correct-mismatches liberal --match-read-wildcards "
                                  "-e 0.2 -O 1 --indel-cost 3 --adapter-max-rmp 1e-6" % (synth.PE_ADAPTER1, synth.PE_ADAPTER2))
        o1, o2 = pipe.trim_bytes(fq(r1, q1), fq(r2, q2))
        rec1, rec2 = o1.decode().split("\n"), o2.decode().split("\n")
        assert len(rec1) == 4 * n + 1 and len(rec2) == 4 * n + 1
        for k in range(n):
            got = [rec1[4 * k + 1], rec1[4 * k + 3], rec2[4 * k + 1], rec2[4 * k + 3]]
            assert _digest(got) == g["text_digests"][k], (k, got, g["full"][k] if k < len(g["full"]) else None)
    return n


def check_plane_guided_correction(n=768, seed=5):
    """atr_insert_correct_batch with the plane buffers (the disagreeing positions found 32 at a time)
    against the same call without them (the byte walk of correct_errors_one, itself pinned to the
    reference by the cutter fixtures): all three correction modes, ragged lengths, reads without
    qualities for 'N'."""
    import numpy as np
    import torch
    from atropos_amd import _lib, synth
    from atropos_amd.align import InsertAligner
    from atropos_amd.modifiers import COMP_TABLE
    be = _lib.get_backend()
    rng = random.Random(seed)
    w = synth.workload("C5", 12345, n, device=str(be.device))
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
    done = 0
    for action, ragged, quals, width in ((2, False, True, 250), (1, False, True, 250), (0, False, True, 250),
                                         (0, False, False, 250), (2, True, True, 250), (2, True, True, 310), (1, False, True, 320)):
        # width > 250: MiSeq-length rows (ten 32-base chunks), the C5 reads with a stretch of themselves appended
        s1, s2, qa, qb = (torch.cat([t, t[:, 20:20 + width - 250]], dim=1).contiguous()
                          for t in (w["reads1"], w["reads2"], w["quals1"], w["quals2"]))
        l1 = l2 = None
        if ragged:
            l1 = torch.tensor([rng.randint(60, width) for _ in range(n)], dtype=torch.int32, device=be.device)
            l2 = torch.tensor([rng.randint(60, width) for _ in range(n)], dtype=torch.int32, device=be.device)
            cols = torch.arange(width, device=be.device)[None, :]
            s1 = torch.where(cols < l1[:, None], s1, torch.zeros_like(s1))
            s2 = torch.where(cols < l2[:, None], s2, torch.zeros_like(s2))
        from atropos_amd.batch import ReadBatch
        table = be.translate_table(_lib.TABLE_DNA15)
        b1 = ReadBatch(be.pack_reads(s1, l1, width, table, planes=True), l1, n, width, _lib.TABLE_DNA15, table, layout="plane64")
        b2 = ReadBatch(be.pack_reads(s2, l2, width, table, planes=True), l2, n, width, _lib.TABLE_DNA15, table, layout="plane64")
        rec = ia.match_insert_batch(b1, b2).records
        outs = []
        for planes in (False, True):
            a1, a2 = s1.clone(), s2.clone()
            q1 = qa.clone() if quals else None
            q2 = qb.clone() if quals else None
            ch, nl = be.insert_correct_batch(rec, a1, q1, l1, a2, q2, l2, action, 1, COMP_TABLE,
                                             planes1=b1 if planes else None, planes2=b2 if planes else None)
            outs.append((a1, a2, q1, q2, ch, nl))
        for x, y in zip(*outs):
            assert (x is None and y is None) or torch.equal(x, y), (action, ragged, quals)
        assert ragged or int((outs[0][4] > 0).any(dim=1).sum()) > n // 8
        done += n
    return done


def check_long_multi_compare():
    """MultiAligner.locate and compare_prefixes / compare_suffixes on strings of 737 .. 3 000 characters against the
    reference's own outputs (long_multi_compare.json.gz, tests/golden/make_round5_golden.py): past the 736 bases of
    the batch pipelines; compare references of more than 1 024 characters are composed from pieces."""
    from atropos_amd.align import MultiAligner, compare_prefixes, compare_suffixes, compare_batch
    g = load_golden("long_multi_compare.json.gz")
    done = 0
    for c in g["compare"]:
        k = 0
        for wr in (False, True):
            for wq in (False, True):
                exp_p, exp_s = c["out"][k]
                k += 1
                assert list(compare_prefixes(c["ref"], c["query"], wr, wq)) == exp_p, (len(c["ref"]), len(c["query"]), wr, wq)
                assert list(compare_suffixes(c["ref"], c["query"], wr, wq)) == exp_s, (len(c["ref"]), len(c["query"]), wr, wq)
                done += 2
        # the batch form: the same reference against the query, a prefix of it and an empty string
        qs = [c["query"], c["query"][:len(c["query"]) // 2], ""]
        for sfx in (False, True):
            rec = compare_batch(c["ref"], qs, False, True, sfx).cpu().numpy()
            assert [int(v) for v in rec[0, :6]] == c["out"][1][1 if sfx else 0]
            one = (compare_suffixes if sfx else compare_prefixes)(c["ref"], qs[1], False, True)
            assert tuple(int(v) for v in rec[1, :6]) == tuple(one)
            assert int(rec[2, 4]) == 0 and int(rec[2, 5]) == 0
    for c in g["multi"]:
        for r in c["runs"]:
            got = MultiAligner(r["e"], r["flags"], r["min_overlap"]).locate(c["ref"], c["query"])
            assert (None if got is None else [list(t) for t in got]) == r["out"], (len(c["ref"]), len(c["query"]), r["flags"])
            done += 1
    # LinkedAdapter.match_to on reads of 737 .. 3 000 bases (one by one and as a batch)
    from atropos_amd.adapters import LinkedAdapter
    from atropos_amd.reads import Sequence

    def fields(m):
        return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
    for c in g["linked"]:
        la = LinkedAdapter(c["front"], c["back"], **c["kw"])
        m = la.match_to(Sequence("r", c["read"]))
        assert (None if m is None else [fields(m.front_match), fields(m.back_match)]) == c["out"], (len(c["read"]), c["front"], c["back"])
        done += 1
    c0 = g["linked"][0]
    la = LinkedAdapter(c0["front"], c0["back"], **c0["kw"])
    batch = la.match_to_batch([Sequence("r%d" % k, c["read"]) for k, c in enumerate(g["linked"][:12])])
    one = [la.match_to(Sequence("r", c["read"])) for c in g["linked"][:12]]
    assert [None if m is None else (fields(m.front_match), fields(m.back_match)) for m in batch] == \
           [None if m is None else (fields(m.front_match), fields(m.back_match)) for m in one]
    return done


def check_fused_match_correct(n=4096, seed=9):
    """atr_insert_match_correct_batch (match + correction in one kernel, the planes streamed once) against
    atr_insert_match_batch followed by atr_insert_correct_batch -- which the cutter fixtures and the checker pin to
    the reference: records, corrected bases and qualities, counts and lengths, all three correction modes, reads
    without qualities for 'N', ragged lengths, layout widths of 4 .. 8 chunks (the fused kernels) and 3 / 9 / 10
    (the two kernels inside the call)."""
    import torch
    from atropos_amd import _lib, synth
    from atropos_amd.align import InsertAligner
    from atropos_amd.batch import ReadBatch
    from atropos_amd.modifiers import COMP_TABLE
    be = _lib.get_backend()
    rng = random.Random(seed)
    w = synth.workload("C5", 77777 + 50021 * seed, n, device=str(be.device))
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
    table = be.translate_table(_lib.TABLE_DNA15)
    done = 0
    for action, ragged, quals, width in ((2, False, True, 250), (1, False, True, 250), (0, False, True, 250), (0, False, False, 250),
                                         (2, True, True, 250), (2, True, True, 150), (2, False, True, 128), (1, True, True, 200),
                                         (2, True, True, 224), (2, False, True, 96), (2, True, True, 300), (2, False, True, 320)):
        if width <= 250:
            # (the C5 fragments are 100 .. 400 bases long: reads cut to `width` keep the overlaps of the shorter ones)
            s1, s2, qa, qb = (t[:, :width].contiguous() for t in (w["reads1"], w["reads2"], w["quals1"], w["quals2"]))
        else:
            s1, s2, qa, qb = (torch.cat([t, t[:, 20:20 + width - 250]], dim=1).contiguous()
                              for t in (w["reads1"], w["reads2"], w["quals1"], w["quals2"]))
        l1 = l2 = None
        if ragged:
            l1 = torch.tensor([rng.randint(60, width) for _ in range(n)], dtype=torch.int32, device=be.device)
            l2 = torch.tensor([rng.randint(60, width) for _ in range(n)], dtype=torch.int32, device=be.device)
            cols = torch.arange(width, device=be.device)[None, :]
            s1 = torch.where(cols < l1[:, None], s1, torch.zeros_like(s1))
            s2 = torch.where(cols < l2[:, None], s2, torch.zeros_like(s2))
        b1 = ReadBatch(be.pack_reads(s1, l1, width, table, planes=True), l1, n, width, _lib.TABLE_DNA15, table, layout="plane64")
        b2 = ReadBatch(be.pack_reads(s2, l2, width, table, planes=True), l2, n, width, _lib.TABLE_DNA15, table, layout="plane64")
        rec = ia.match_insert_batch(b1, b2).records
        a1, a2 = s1.clone(), s2.clone()
        q1, q2 = (qa.clone(), qb.clone()) if quals else (None, None)
        ch, nl = be.insert_correct_batch(rec, a1, q1, l1, a2, q2, l2, action, 1, COMP_TABLE, planes1=b1, planes2=b2)
        f1, f2 = s1.clone(), s2.clone()
        g1, g2 = (qa.clone(), qb.clone()) if quals else (None, None)
        frec, fch, fnl = be.insert_match_correct_batch(ia._handle, b1, b2, f1, g1, f2, g2, action, 1, COMP_TABLE)
        for x, y, what in ((rec, frec, "records"), (a1, f1, "read 1"), (a2, f2, "read 2"), (q1, g1, "qualities 1"),
                           (q2, g2, "qualities 2"), (ch, fch, "changed"), (nl, fnl, "newlen")):
            assert (x is None and y is None) or torch.equal(x, y), (what, action, ragged, quals, width)
        assert int((ch > 0).any(dim=1).sum()) > (n // 16 if width >= 200 and not ragged and n >= 1000 else -1), (action, ragged, width)
        done += n
    # the method of the aligner (mismatch_action by name)
    s1, s2, qa, qb = (w[k].clone() for k in ("reads1", "reads2", "quals1", "quals2"))
    b1, b2 = ia.pack(s1), ia.pack(s2, check=True)
    res, ch, nl = ia.match_insert_correct_batch(b1, b2, s1, qa, s2, qb, "liberal", 1)
    t1, t2, u1, u2 = (w[k].clone() for k in ("reads1", "reads2", "quals1", "quals2"))
    rec = ia.match_insert_batch(b1, b2).records
    ch2, nl2 = be.insert_correct_batch(rec, t1, u1, None, t2, u2, None, 2, 1, COMP_TABLE, planes1=b1, planes2=b2)
    for x, y in ((res.records, rec), (s1, t1), (s2, t2), (qa, u1), (qb, u2), (ch, ch2), (nl, nl2)):
        assert torch.equal(x, y)
    return done + n


def check_device_resident_adapters():
    """The device-resident twins (match_records, LinkedAdapter.match_records,
    best_adapter_records) against the object-level batch path, which is itself pinned to the
    reference's outputs."""
    import numpy as np
    import torch
    from atropos_amd import _lib, synth
    from atropos_amd.adapters import (Adapter, LinkedAdapter, best_adapter_records, upper_ascii,
                                      BACK, FRONT, PREFIX, SUFFIX, ANYWHERE)
    from atropos_amd.reads import Sequence
    from atropos_amd.util import RandomMatchProbability
    be = _lib.get_backend()
    rng = random.Random(123)

    def recs(matches):
        out = np.zeros((len(matches), 8), dtype=np.int16)
        out[:, 1] = -1
        for i, m in enumerate(matches):
            if m is not None:
                out[i, :6] = [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]
        return out

    total = 0
    rmp = RandomMatchProbability()
    for it in range(40):
        m = rng.randint(4, 40)
        seq = rseq(rng, m, "ACGT" if rng.random() < 0.7 else "ACGTN")
        where = rng.choice([BACK, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE])
        use_rmp = rng.random() < 0.3
        kw = dict(max_error_rate=rng.choice([0.1, 0.12, 0.2]), min_overlap=rng.choice([1, 3, 5]),
                  read_wildcards=rng.random() < 0.25, adapter_wildcards=rng.random() < 0.8,
                  indels=rng.random() < 0.8, indel_cost=rng.choice([1, 1, 3]),
                  match_probability=rmp if use_rmp else None, max_rmp=1e-6 if use_rmp else None)
        ad = Adapter(seq, where, **kw)
        reads = planted_reads(rng, seq.replace("N", "A"), 130, 150)
        reads = [r if rng.random() > 0.1 else r.lower() for r in reads]
        reads = [r if r else "A" for r in reads]
        width = max(len(r) for r in reads)
        mat = np.zeros((len(reads), width), dtype=np.uint8)
        for i, r in enumerate(reads):
            mat[i, :len(r)] = np.frombuffer(r.encode(), dtype=np.uint8)
        a_t = upper_ascii(torch.from_numpy(mat).to(be.device))
        l_t = torch.tensor([len(r) for r in reads], dtype=torch.int32, device=be.device)
        got = ad.match_records(a_t, l_t).cpu().numpy()
        exp = recs(ad.match_to_batch([Sequence("r", r) for r in reads]))
        assert np.array_equal(got[:, :6], exp[:, :6]), (seq, where, kw)
        total += len(reads)
    # linked adapters on the C4 head: device path == object path (== reference, see check_linked_c4)
    w = synth.workload("C4", 0, 512)
    reads = [Sequence("r", bytes(x.tolist()).decode("ascii")) for x in w["reads"]]
    a_t = upper_ascii(w["reads"].to(be.device))
    for fr, bk in zip(w["fronts"], w["backs"]):
        la = LinkedAdapter(fr, bk, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                           min_overlap=w["min_overlap"], indel_cost=w["indel_cost"])
        front, back = la.match_records(a_t)
        lm = la.match_to_batch(reads)
        assert np.array_equal(front.cpu().numpy()[:, :6], recs([None if x is None else x.front_match for x in lm])[:, :6])
        assert np.array_equal(back.cpu().numpy()[:, :6], recs([None if x is None else x.back_match for x in lm])[:, :6])
        total += len(reads)
    # grouped pipeline over all four linked adapters
    from atropos_amd.adapters import linked_best_records
    las = [LinkedAdapter(fr, bk, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                         min_overlap=w["min_overlap"], indel_cost=w["indel_cost"])
           for fr, bk in zip(w["fronts"], w["backs"])]
    which, front, back = linked_best_records(las, a_t)
    per_l = [la.match_to_batch(reads) for la in las]
    for i in range(len(reads)):
        hits = [k for k in range(4) if per_l[k][i] is not None]
        assert int(which[i]) == (hits[0] if len(hits) == 1 else (-1 if not hits else -2))
        if len(hits) == 1:
            lm = per_l[hits[0]][i]
            assert front[i, :6].tolist() == recs([lm.front_match])[0, :6].tolist()
            assert back[i, :6].tolist() == recs([lm.back_match])[0, :6].tolist()
    # best-of-N
    ads = [Adapter(b, BACK, max_error_rate=0.12, min_overlap=3) for b in w["backs"]]
    best, which = best_adapter_records(ads, a_t)
    per = [ad.match_to_batch(reads) for ad in ads]
    for i in range(len(reads)):
        choice, bi = None, -1
        for k in range(len(ads)):
            mt = per[k][i]
            if mt is not None and (choice is None or mt.matches > choice.matches):
                choice, bi = mt, k
        assert int(which[i]) == bi
        if choice is not None:
            assert best[i, :6].tolist() == [choice.astart, choice.astop, choice.rstart, choice.rstop, choice.matches, choice.errors]
    return total


# ---------------------------------------------------------------------------------------------
# device-resident FASTQ pipeline (atropos_amd.trim) against outputs of the reference's
# `atropos trim` command line (tests/golden/make_trim_golden.py)
def check_trim_golden(select=None):
    import base64
    import hashlib
    from atropos_amd.fastq import FormatError
    from atropos_amd.trim import pipeline_from_args
    doc = load_golden("trim_cases.json.gz")
    inputs = {k: base64.b64decode(v) for k, v in doc["inputs"].items()}
    done = 0
    for case in doc["cases"]:
        if select is not None and not select(case):
            continue
        data = inputs[case["input"]]
        label = "%s: %s" % (case["input"], case["args"])
        if case.get("aux"):
            # --info-file / --rest-file / --wildcard-file: the three texts next to the main output
            from atropos_amd.fastq import FastqBatch
            args = case["args"]
            for kind in case["aux"]:
                args = args.replace("{%s}" % kind, kind + ".txt")
            pipe = pipeline_from_args(args)
            batch, _ = FastqBatch.from_bytes(data, final=True)
            res = pipe.run(batch)
            lines = [k for k in case["aux"] if k in ("info", "rest", "wildcard")]
            aux = res.aux_text(tuple(lines)) if lines else {}
            from atropos_amd.trim import DEST_NAMES
            for code, kind in DEST_NAMES.items():          # --too-short-output etc.: the reads of that destination
                if kind in case["aux"]:
                    aux[kind] = res.text(code)
            for kind, want in case["aux"].items():
                want = base64.b64decode(want)
                assert aux[kind] == want, (label, kind, _first_diff(aux[kind], want))
            out = res.text()
            assert len(out) == case["size"] and hashlib.sha256(out).hexdigest() == case["sha256"], label
            done += 1
            continue
        pipe = pipeline_from_args(case["args"])
        if case["error"]:
            try:
                pipe.trim_bytes(data)
            except FormatError as err:
                assert type(err).__name__ == case["error"][0], label
                assert str(err) == case["error"][1], (label, str(err), case["error"][1])
                if len(case["error"]) > 2:
                    assert str(err.__cause__) == case["error"][2], (label, str(err.__cause__), case["error"][2])
            else:
                raise AssertionError("no FormatError raised for " + label)
        else:
            out = pipe.trim_bytes(data)
            want = base64.b64decode(case["output"])
            if case["size"] <= 20000:
                assert out == want, (label, out[:300], want[:300])
            else:
                head = out[:len(want)]
                assert head == want, (label, _first_diff(head, want))
            assert len(out) == case["size"], (label, len(out), case["size"])
            assert hashlib.sha256(out).hexdigest() == case["sha256"], label
        done += 1
    return done


def _first_diff(a, b):
    for i in range(min(len(a), len(b))):
        if a[i] != b[i]:
            lo = max(0, i - 200)
            return i, a[lo:i + 100], b[lo:i + 100]
    return min(len(a), len(b)), None, None


def check_fastq_chunking(tmp_path):
    """trim_file in small chunks == one batch (records straddling chunk borders are carried over)."""
    import base64
    from atropos_amd.trim import pipeline_from_args
    doc = load_golden("trim_cases.json.gz")
    data = base64.b64decode(doc["inputs"]["synth.fastq"])
    pipe = pipeline_from_args("-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCA -q 20 -m 20")
    whole = pipe.trim_bytes(data)
    src, dst = tmp_path / "in.fastq", tmp_path / "out.fastq"
    src.write_bytes(data)
    counts = pipe.trim_file(str(src), str(dst), chunk_bytes=50000)
    assert dst.read_bytes() == whole
    assert sum(counts.values()) == data.count(b"\n") // 4
    # compressed input (.gz / .bz2 / .xz, the extensions the reference's xopen knows), chunks that end inside a record
    import bz2
    import gzip
    import lzma
    for ext, mod in ((".gz", gzip), (".bz2", bz2), (".xz", lzma)):
        zsrc = tmp_path / ("in.fastq" + ext)
        with mod.open(str(zsrc), "wb") as fh:
            fh.write(data)
        zdst = tmp_path / ("out_from" + ext + ".fastq")
        assert pipe.trim_file(str(zsrc), str(zdst), chunk_bytes=37000) == counts
        assert zdst.read_bytes() == whole
        zout = tmp_path / ("out.fastq" + ext)                 # ... and compressed output
        assert pipe.trim_file(str(src), str(zout), chunk_bytes=41000) == counts
        with mod.open(str(zout), "rb") as fh:
            assert fh.read() == whole
    # the output as three part files with a writer each: chunk k in part k mod 3, in order inside a part -- the records
    # of all parts are those of the single file
    (tmp_path / "out.fastq.part3").write_bytes(b"stale part of an earlier run with more parts\n")
    (tmp_path / "out.fastq.part7").write_bytes(b"stale\n")
    (tmp_path / "out.fastq.partial").write_bytes(b"not a part\n")
    counts3 = pipe.trim_file(str(src), str(dst), chunk_bytes=50000, output_parts=3)
    assert not (tmp_path / "out.fastq.part3").exists() and not (tmp_path / "out.fastq.part7").exists()
    assert (tmp_path / "out.fastq.partial").exists()
    parts = [(tmp_path / ("out.fastq.part%d" % i)).read_bytes() for i in range(3)]
    assert counts3 == counts and all(parts) and sum(len(p) for p in parts) == len(whole)

    def records(text):
        lines = text.split(b"\n")
        return [b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4)]
    assert sorted(r for p in parts for r in records(p)) == sorted(records(whole))
    k, rebuilt, at = 0, [], [0, 0, 0]               # chunk k is the next run of part k % 3: walk the single file
    whole_records = records(whole)
    part_records = [records(p) for p in parts]
    pos = 0
    while pos < len(whole_records):
        i = k % 3
        n = 0
        while at[i] + n < len(part_records[i]) and pos + n < len(whole_records) and part_records[i][at[i] + n] == whole_records[pos + n]:
            n += 1
        assert n > 0, (k, pos)
        at[i] += n; pos += n; k += 1
    assert at == [len(p) for p in part_records]
    # the sink on its own: odd chunk sizes around the 4 KiB block, buffered and (where the file system and the
    # staging buffers allow it) O_DIRECT
    import torch
    from atropos_amd import _lib
    from atropos_amd.fastq import FastqSink
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(3)
    pieces = [torch.randint(0, 255, (sz,), dtype=torch.uint8, generator=gen) for sz in (1, 4095, 4096, 4097, 123457, 0, 8191, 70000)]
    for direct in (False, True):
        out = tmp_path / ("sink_%d.bin" % direct)
        sink = FastqSink(str(out), 200000, be, direct=direct)
        for piece in pieces:
            sink.write(piece.to(be.device))
        sink.close()
        assert out.read_bytes() == b"".join(bytes(p.numpy().tobytes()) for p in pieces), direct
    return counts


# ---------------------------------------------------------------------------------------------
# PairAligner (Aligner.locate with a per-pair reference) against the oracle
def check_pairs_against_oracle(PairAligner, oracle, unsupported_exc, seed, rounds, max_len=320):
    """Random settings (all flag sets, indel regimes, wildcard modes, with and without the
    on-device reverse complement); pairs are overlapping fragments with errors, unrelated
    sequences, and ragged / empty ones."""
    from atropos_amd.util import reverse_complement
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        flags = rng.randint(0, 15) if rng.random() < 0.4 else rng.choice([15, 9, 14, 11])
        e = rng.choice([0, 0.05, 0.1, 0.2, 0.2, 0.3])
        ic = rng.choice([1, 1, 1, 2, 3, 100000])
        mo = rng.choice([1, 1, 3, 20])
        wr, wq = (rng.random() < 0.2, rng.random() < 0.2)
        rc = rng.random() < 0.5
        top = rng.choice([12, 40, 100, 150, 255, max_len])
        npairs = rng.choice([1, 5, 64, 65, 130])
        refs, qrys = [], []
        for _p in range(npairs):
            m, n = rng.randint(0, top), rng.randint(0, top)
            alpha = "ACGT" if rng.random() < 0.8 else "ACGTN"
            frag = rseq(rng, m + n, alpha)
            kind = rng.random()
            if kind < 0.6:                                   # overlapping pair: both cut from one fragment
                off = rng.randint(0, max(0, m // 2))
                ref = frag[off:off + m]
                q0 = rng.randint(0, m + n - n) if n else 0
                qry = mutate(rng, frag[q0:q0 + n], rng.choice([0, 0.02, 0.1]), "ACGT")[:n]
            else:
                ref, qry = frag[:m], rseq(rng, n, alpha)
            refs.append(ref[:top])
            qrys.append(qry[:top])
        given = [reverse_complement(r) for r in refs] if rc else refs     # what the device is handed
        try:
            pa = PairAligner(e, flags, wr, wq, mo, ic, revcomp_ref=rc)
            got = pa.locate_batch(given, qrys).tuples()               # short batch: a wavefront per pair (<= 319 rows)
        except unsupported_exc:
            continue
        for path in ("full", "fast") + (("wave",) if top <= 319 else ()):
            assert pa.locate_batch(given, qrys, path=path).tuples() == got, (path, e, flags, wr, wq, mo, ic, rc, top)
        for ref, qry, g in zip(refs, qrys, got):
            assert g == oracle.locate(ref, qry, e, flags, wr, wq, mo, ic), (ref, qry, e, flags, wr, wq, mo, ic, rc, g)
            total += 1
    return total


# ---------------------------------------------------------------------------------------------
# MergeOverlapping against the reference's outputs (tests/golden/make_merge_golden.py)
def check_merge_golden(batch=True):
    from atropos_amd.modifiers import MergeOverlapping
    from atropos_amd.reads import Sequence
    cases = load_golden("merge_fuzz.json.gz")
    groups = {}
    for k, case in enumerate(cases):
        groups.setdefault(json.dumps(case["cfg"], sort_keys=True), []).append(k)
    done = 0
    for key, idxs in groups.items():
        cfg = json.loads(key)
        good = [k for k in idxs if "error" not in cases[k]]
        bad = [k for k in idxs if "error" in cases[k]]
        def mk(k):
            c = cases[k]
            a, b = Sequence("p/1", c["r1"], c["q1"]), Sequence("p/2", c["r2"], c["q2"])
            a.insert_overlap = b.insert_overlap = c["insert"]
            return a, b
        pairs = [mk(k) for k in good]
        if batch:
            # counters are per modifier in the reference fixtures: one modifier per pair for those,
            # one shared modifier for the batched call
            mod = MergeOverlapping(**cfg)
            outs = mod.call_batch([p[0] for p in pairs], [p[1] for p in pairs])
            assert mod.corrected_pairs == sum(cases[k]["out"]["corrected_pairs"] for k in good)
        else:
            outs = [MergeOverlapping(**cfg)(a, b) for a, b in pairs]
        for k, (o1, o2) in zip(good, outs):
            want = cases[k]["out"]
            got = dict(seq1=o1.sequence, qual1=o1.qualities, merged=bool(o1.merged), read2_none=o2 is None,
                       seq2=None if o2 is None else o2.sequence, qual2=None if o2 is None else o2.qualities,
                       corrected=[int(o1.corrected), 0 if o2 is None else int(o2.corrected)])
            for f in got:
                assert got[f] == want[f], (k, f, got[f], want[f], cases[k])
            done += 1
        for k in bad:                                         # the reference raises for these pairs
            a, b = mk(k)
            try:
                MergeOverlapping(**cfg)(a, b)
            except Exception as err:                          # noqa: BLE001
                assert type(err).__name__ in (cases[k]["error"], "ValueError"), (k, type(err).__name__, cases[k]["error"])
            else:
                raise AssertionError("pair %d: the reference raises %s" % (k, cases[k]["error"]))
            done += 1
    return done


def check_trim_golden_paired():
    """PairedTrimPipeline against the two output files of the reference's paired-end command."""
    import base64
    import hashlib
    from atropos_amd.trim import pipeline_from_args, PairedTrimPipeline
    doc = load_golden("trim_cases.json.gz")
    inputs = {k: base64.b64decode(v) for k, v in doc["inputs"].items()}
    done = 0
    for case in doc["paired"]:
        label = "%s: %s" % (case["input1"], case["args"])
        args = case["args"]
        for kind in case.get("aux", {}):
            args = args.replace("{%s}" % kind, kind + ".txt")
        pipe = pipeline_from_args(args, paired_input=True)
        assert isinstance(pipe, PairedTrimPipeline), label
        from atropos_amd.fastq import FastqBatch
        b1, _ = FastqBatch.from_bytes(inputs[case["input1"]], final=True)
        b2, _ = FastqBatch.from_bytes(inputs[case["input2"]], final=True)
        res = pipe.run(b1, b2)
        outs = list(res.text())
        if case.get("aux"):
            lines = [k for k in case["aux"] if k in ("info", "rest", "wildcard")]
            aux = res.aux_text(tuple(lines)) if lines else {}
            from atropos_amd.trim import DEST_NAMES
            for code, kind in DEST_NAMES.items():          # --too-short-output / --too-short-paired-output etc.
                if kind in case["aux"]:
                    aux[kind], aux[kind + "2"] = res.text(code)
            for kind, want in case["aux"].items():
                want = base64.b64decode(want)
                assert aux[kind] == want, (label, kind, _first_diff(aux[kind], want))
        if len(case["outputs"]) == 3:                        # -R: the --merged-output file
            outs.append(res.merged_text())
            assert res.counts()["merged"] == outs[2].count(b"\n") // 4 == pipe.merged_pairs, label
        for k, (out, want) in enumerate(zip(outs, case["outputs"])):
            text = base64.b64decode(want["text"])
            head = out[:len(text)] if want["size"] > 20000 else out
            assert head == text, (label, k, _first_diff(head, text))
            assert len(out) == want["size"], (label, k, len(out), want["size"])
            assert hashlib.sha256(out).hexdigest() == want["sha256"], (label, k)
        done += 1
    return done


def check_paired_file_chunking(tmp_path):
    """PairedTrimPipeline.trim_files in small lock-step chunks == one batch."""
    import base64
    import hashlib
    from atropos_amd.trim import pipeline_from_args
    doc = load_golden("trim_cases.json.gz")
    d1, d2 = (base64.b64decode(doc["inputs"][k]) for k in ("synth_pe.1.fastq", "synth_pe.2.fastq"))
    case = [c for c in doc["paired"] if c["input1"] == "synth_pe.1.fastq" and "-q 20 -m 30" in c["args"]][0]
    w1, w2 = pipeline_from_args(case["args"]).trim_bytes(d1, d2)
    paths = [tmp_path / n for n in ("a1.fastq", "a2.fastq", "o1.fastq", "o2.fastq")]
    paths[0].write_bytes(d1)
    paths[1].write_bytes(d2)
    counts = pipeline_from_args(case["args"]).trim_files(str(paths[0]), str(paths[1]), str(paths[2]), str(paths[3]),
                                                         chunk_bytes=30000)
    assert paths[2].read_bytes() == w1 and paths[3].read_bytes() == w2
    assert sum(counts.values()) == d1.count(b"\n") // 4
    # records of different sizes in the two files (longer descriptions in file 2): the file with the smaller
    # records carries surplus records from chunk to chunk; that surplus must stay bounded
    lines2 = d2.split(b"\n")
    for i in range(0, len(lines2) - 1, 4):
        lines2[i] += b" a much longer description than its mate has, to make the records unequal"
    d2_long = b"\n".join(lines2)
    want1, want2 = pipeline_from_args(case["args"]).trim_bytes(d1, d2_long)
    paths[1].write_bytes(d2_long)
    pipe = pipeline_from_args(case["args"])
    pipe.trim_files(str(paths[0]), str(paths[1]), str(paths[2]), str(paths[3]), chunk_bytes=12000)
    assert paths[2].read_bytes() == want1 and paths[3].read_bytes() == want2
    paths[1].write_bytes(d2)
    # with merging: a third file, written chunk by chunk
    case = [c for c in doc["paired"] if "--merge-min-overlap 20" in c["args"]][0]
    merged_path = tmp_path / "merged.fastq"
    mc = pipeline_from_args(case["args"]).trim_files(str(paths[0]), str(paths[1]), str(paths[2]), str(paths[3]),
                                                     chunk_bytes=30000, merged_out=str(merged_path))
    texts = [paths[2].read_bytes(), paths[3].read_bytes(), merged_path.read_bytes()]
    assert [len(t) for t in texts] == [o["size"] for o in case["outputs"]]
    assert [hashlib.sha256(t).hexdigest() for t in texts] == [o["sha256"] for o in case["outputs"]]
    assert mc["merged"] == texts[2].count(b"\n") // 4 and sum(mc.values()) == d1.count(b"\n") // 4
    # two part files per output: part i of both reads holds the same pairs, all parts together the single-file records
    mc2 = pipeline_from_args(case["args"]).trim_files(str(paths[0]), str(paths[1]), str(paths[2]), str(paths[3]),
                                                      chunk_bytes=30000, merged_out=str(merged_path), output_parts=2)
    assert mc2 == mc
    for whole_text, base in zip(texts, (paths[2], paths[3], merged_path)):
        parts = [(tmp_path / (base.name + ".part%d" % i)).read_bytes() for i in range(2)]
        assert sorted(b"".join(parts).split(b"\n")) == sorted(whole_text.split(b"\n"))
    names = [[ln.split()[0] for ln in (tmp_path / (b.name + ".part%d" % i)).read_bytes().split(b"\n")[0::4] if ln]
             for b in (paths[2], paths[3]) for i in range(2)]
    assert [n.rstrip(b"12").rstrip(b"/") for n in names[0]] == [n.rstrip(b"12").rstrip(b"/") for n in names[2]]
    assert [n.rstrip(b"12").rstrip(b"/") for n in names[1]] == [n.rstrip(b"12").rstrip(b"/") for n in names[3]]
    return counts


def check_fastq_reader_golden():
    """FastqBatch.from_bytes against the reference's FastqReader on fuzzed FASTQ texts: same record
    tuples, or the same FormatError (type, message, cause)."""
    import base64
    from atropos_amd.fastq import FastqBatch, FormatError
    cases = load_golden("fastq_fuzz.json.gz")
    errors = 0
    for k, case in enumerate(cases):
        text = base64.b64decode(case["text"])
        if "error" in case:
            try:
                FastqBatch.from_bytes(text, final=True)
            except FormatError as err:
                assert case["error"][0] == "FormatError", (k, case["error"])
                assert str(err) == case["error"][1], (k, str(err), case["error"][1])
                if len(case["error"]) > 2:
                    assert str(err.__cause__) == case["error"][2], (k, str(err.__cause__), case["error"][2])
                errors += 1
            else:
                raise AssertionError("text %d: the reference raises %s" % (k, case["error"]))
        else:
            batch, consumed = FastqBatch.from_bytes(text, final=True)
            got = [list(r) for r in batch.to_records()]
            assert got == case["records"], (k, got[:3], case["records"][:3])
    return len(cases), errors


def check_read2_validation(ia):
    """match_insert reverse-complements read 2 only as far as read 1 reaches (reference
    align/__init__.py:259-267): letters without a code beyond that point are none of its business,
    neither per pair nor in a batch (uint8 matrices and the FASTQ pipelines included)."""
    import pytest
    r1 = "ACGTTGCAAGGCTA"
    r2 = "TAGCCTTGCAACGT"
    want = ia.match_insert(r1, r2)
    assert ia.match_insert(r1, r2 + "xx..acgt") == ia.match_insert(r1, r2 + "GGTTACGT")
    got = ia.match_insert_batch([r1, r1, r1[:7]], [r2, r2 + "nn!?", r2[:7] + "acgtacg"]).results()
    assert got[0] == want and got[1] == ia.match_insert(r1, r2 + "AAAA") and got[2] == ia.match_insert(r1[:7], r2[:7])
    for at in (0, 5, len(r1) - 1):
        broken = r2[:at] + "x" + r2[at + 1:] + "ACGT"
        with pytest.raises(ValueError):
            ia.match_insert_batch([r1, r1], [r2, broken])
    # 33 .. 70 bases: the boundary inside and at the end of a 32-base chunk
    for n1 in (31, 32, 33, 63, 64, 65):
        a = ("ACGTTGCA" * 10)[:n1]
        b = ("TGCAACGT" * 10)[:n1]
        ia.match_insert_batch([a], [b + "?"])
        with pytest.raises(ValueError):
            ia.match_insert_batch([a], [b[:-1] + "?"])


def check_dpmatrix_golden(Aligner):
    """Aligner.enable_debug(): the printed DP matrix and the result of the same locate() call against what
    the reference prints (tests/golden/dpmatrix.json.gz: every cell it computed, blanks where it did not)."""
    done = 0
    for c in load_golden("dpmatrix.json.gz"):
        al = Aligner(c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"])
        al.indel_cost = c["ic"]
        assert al.dpmatrix is None
        al.enable_debug()
        res = al.locate(c["query"])
        assert (None if res is None else list(res)) == c["out"], c
        assert str(al.dpmatrix) == c["matrix"], (c, str(al.dpmatrix))
        done += 1
    return done


def check_long_reference(Aligner, oracle, AtroposHipError, batch_rounds=6):
    """Aligner with a reference of 129 .. 320 bases (no aligner handle: every read goes through the per-pair
    aligner with the same reference): the reference's own results (tests/golden/long_reference.json.gz),
    then batches of reads against the oracle."""
    done = 0
    for c in load_golden("long_reference.json.gz"):
        al = Aligner(c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"])
        al.indel_cost = c["ic"]
        assert oracle.locate(c["ref"], c["query"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]) == \
            (None if c["out"] is None else tuple(c["out"])), c
        # (queries of more than 320 bases, and sides of more than 255 without STOP_WITHIN_SEQ2, used to be refused: they
        # take the per-pair aligner's long path since round 4)
        res = al.locate(c["query"])
        assert (None if res is None else list(res)) == c["out"], (c, res)
        done += 1
    rng = random.Random(4242)
    for it in range(batch_rounds):
        m = rng.choice([129, 200, 256, 320])
        ref = rseq(rng, m)
        flags = rng.choice([14, 11, 15, 9])
        e, mo = rng.choice([0.05, 0.1]), rng.choice([1, 5])
        al = Aligner(ref, e, flags, False, False, mo)
        al.LONG_CHUNK = 50                                     # several chunks per call
        reads = []
        for _ in range(rng.choice([1, 64, 130])):
            n = rng.randint(0, 300)
            keep = rng.randint(0, min(m, n))
            reads.append(rng.choice([(rseq(rng, n) + ref[:keep])[-n:] if n else "", (ref[m - keep:] + rseq(rng, n))[:n],
                                     rseq(rng, n)]))
        got = al.locate_batch(reads).tuples()
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, False, False, mo, 1), (ref, q, e, flags, mo, g)
            done += 1
    assert Aligner("A" * 321, 0.1).locate("C" * 10 + "A" * 400) == oracle.locate("A" * 321, "C" * 10 + "A" * 400, 0.1, 15, False, False, 1, 1)
    try:
        Aligner("A" * 32737, 0.1)
        raise AssertionError("a 32 737-base reference should be outside the envelope")
    except AtroposHipError:
        pass
    return done


def check_ragged_tail_mode(Aligner, oracle, seed, nreads=20000, oracle_slice=1500):
    """Ragged batches at a size where the row-count bins fill whole waves (the window kernel's tail mode:
    columns counted from the read end): C2-like reads cut to random lengths -- including reads shorter than
    the adapter and empty ones -- for 3' adapters of several lengths and error rates; the filtered
    pipeline against the full sweep on every read and against the oracle on a slice."""
    import numpy as np
    import torch
    from atropos_amd import _lib, synth
    from atropos_amd.batch import ReadBatch
    rng = random.Random(seed)
    be = _lib.get_backend()
    w = synth.workload("C2", seed, nreads, device="cpu")
    ascii_t = w["reads"].to(be.device)
    total = 0
    for m, e, ic, flags in ((34, 0.1, 1, 14), (20, 0.15, 1, 14), (12, 0.2, 2, 14), (40, 0.12, 1, 10), (34, 0.1, 1, 15)):
        ref = w["adapter"][:m] if m <= len(w["adapter"]) else w["adapter"] + rseq(rng, m - len(w["adapter"]))
        al = Aligner(ref, e, flags, False, False, rng.choice([1, 3]), ic)
        g = torch.Generator().manual_seed(seed + m)
        lens = torch.randint(0, 151, (nreads,), generator=g, dtype=torch.int32)
        lens[torch.rand(nreads, generator=g) < 0.5] = 150                      # half of them full length
        lens[torch.rand(nreads, generator=g) < 0.1] += 0
        lens_d = lens.to(be.device)
        rb = ReadBatch.from_ascii(ascii_t, lens_d, 150, al.table_kind, al._table)
        got = al.locate_batch(rb).records.cpu()
        full = al.locate_batch(rb, filtered=False).records.cpu()
        assert torch.equal(got, full), (m, e, ic, flags, int((got != full).any(dim=1).sum()))
        reads = w["reads"].numpy()
        idx = [rng.randrange(nreads) for _ in range(oracle_slice)]
        tuples = LocateTuples(got)
        for i in idx:
            q = bytes(reads[i, :int(lens[i])]).decode("ascii")
            assert tuples[i] == oracle.locate(ref, q, e, flags, False, False, al.min_overlap, ic), (ref, q, e, flags)
        total += nreads
    return total


def LocateTuples(records):
    """int16 [n, 8] records -> list of the reference's 6-tuples / None."""
    out = []
    for row in records.tolist():
        out.append(None if row[1] < 0 else tuple(row[:6]))
    return out


def check_correct_errors_fixture():
    """atr_correct_errors_batch (the device twin of ErrorCorrectorMixin.correct_errors, byte walk) against
    correct_errors_fuzz.json.gz: 4 000 reference cases -- three actions, min_qual_difference, truncate_seqs,
    unequal lengths, index wrap-around and the three exceptions.  Cases are grouped by the per-call
    arguments (action, min_qual_difference, truncate_seqs, qualities present)."""
    import numpy as np
    import torch
    from atropos_amd import _lib
    from atropos_amd.modifiers import COMP_TABLE
    be = _lib.get_backend()
    cases = load_golden("correct_errors_fuzz.json.gz")
    assert len(cases) == 4000
    groups = {}
    for c in cases:
        has_q = bool(c["qual1"]) and bool(c["qual2"])
        groups.setdefault((c["action"], c["mqd"], c["truncate"], has_q), []).append(c)
    codes = {"KeyError": -1, "IndexError": -2, "ValueError": -3}
    actions = {"N": 0, "conservative": 1, "liberal": 2}
    done = 0
    for (action, mqd, truncate, has_q), group in sorted(groups.items(), key=repr):
        if not has_q and action != "N":
            # the reference raises ValueError before it looks at the reads (:244-248); the host wrappers raise too
            assert all(c["out"] == {"error": "ValueError"} for c in group)
            done += len(group)
            continue
        n = len(group)
        width = max(max(len(c["seq1"]), len(c["seq2"])) for c in group)

        def mat(key):
            m = np.zeros((n, width), dtype=np.uint8)
            for r, c in enumerate(group):
                b = c[key].encode("latin-1")
                m[r, :len(b)] = np.frombuffer(b, dtype=np.uint8)
            return torch.from_numpy(m).to(be.device)

        s1, s2 = mat("seq1"), mat("seq2")
        q1, q2 = (mat("qual1"), mat("qual2")) if has_q else (None, None)
        l1 = torch.tensor([len(c["seq1"]) for c in group], dtype=torch.int32, device=be.device)
        l2 = torch.tensor([len(c["seq2"]) for c in group], dtype=torch.int32, device=be.device)
        im = torch.tensor([c["im"] for c in group], dtype=torch.int16, device=be.device)
        changed, newlen = be.correct_errors_batch(s1, q1, l1, s2, q2, l2, im, None, actions[action], mqd, truncate, COMP_TABLE)
        changed, newlen = changed.cpu().numpy(), newlen.cpu().numpy()
        hs1, hs2 = s1.cpu().numpy(), s2.cpu().numpy()
        hq1, hq2 = (q1.cpu().numpy(), q2.cpu().numpy()) if has_q else (None, None)
        for r, c in enumerate(group):
            exp = c["out"]
            if "error" in exp:
                # the exception ends the reference's run; the callers of the batch raise it too (trim.py), so
                # what the in-place kernel had already rewritten of such a pair is never looked at
                assert changed[r, 0] == codes[exp["error"]], (c, changed[r])
                continue
            assert list(changed[r]) == exp["corrected"], (c, changed[r])
            assert newlen[r, 0] == len(exp["seq1"]) and newlen[r, 1] == len(exp["seq2"]), (c, newlen[r])
            assert bytes(hs1[r, :newlen[r, 0]]).decode("latin-1") == exp["seq1"], c
            assert bytes(hs2[r, :newlen[r, 1]]).decode("latin-1") == exp["seq2"], c
            if has_q:
                assert bytes(hq1[r, :len(exp["qual1"])]).decode("latin-1") == exp["qual1"], c
                assert bytes(hq2[r, :len(exp["qual2"])]).decode("latin-1") == exp["qual2"], c
        done += n
    return done


# ---------------------------------------------------------------------------------------------
# The fast pair pipeline (pairs_fast_core.hpp: bit-vector costs, threat analysis, banded payload) against the oracle
def check_pairs_fast(PairAligner, oracle, seed, rounds, top=150, npairs=96, adversarial=True):
    """Settings the fast pipeline takes (STOP_WITHIN_SEQ2, indel cost 1, literal compare): overlapping read
    pairs cut from one fragment (both overlap directions, substitutions and indels), unrelated pairs, and --
    adversarial -- low-complexity / tandem-repeat fragments, where many diagonals tie and the banded pass must
    either certify its payload or hand the pair to the full sweep."""
    from atropos_amd.util import reverse_complement
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        flags = rng.choice([15, 9, 15, 9, 8, 10, 12, 11, 13, 14])
        e = rng.choice([0.05, 0.1, 0.2, 0.2, 0.2, 0.3])
        mo = rng.choice([1, 1, 1, 3, 20])
        rc = rng.random() < 0.7
        ragged = rng.random() < 0.4
        refs, qrys = [], []
        for _p in range(npairs):
            m = rng.randint(max(1, top // 3), top) if ragged else top
            n = rng.randint(max(1, top // 3), top) if ragged else top
            kind = rng.random()
            if adversarial and kind < 0.25:
                unit = rseq(rng, rng.choice([1, 2, 3, 5, 7, 11]), "ACGT")
                frag = (unit * (2 * (m + n) // len(unit) + 2))[:2 * (m + n)]
                frag = mutate(rng, frag, rng.choice([0, 0.01, 0.05]), "ACGT")
            else:
                frag = rseq(rng, 2 * (m + n), "ACGT" if rng.random() < 0.85 else "ACGTN")
            if kind < 0.85:
                # read 1 = frag[a : a + n], the reference = frag[b : b + m]: any relative offset
                a = rng.randint(0, m + n - 1)
                b = rng.randint(max(0, a - m + 1), a + n - 1) if rng.random() < 0.85 else rng.randint(0, m + n - 1)
                p_err = rng.choice([0, 0.01, 0.02, 0.05, 0.1])
                qry = mutate(rng, frag[a:a + n], p_err, "ACGT")[:n]
                ref = mutate(rng, frag[b:b + m], p_err / 2, "ACGT")[:m]
            else:
                ref, qry = rseq(rng, m, "ACGT"), rseq(rng, n, "ACGT")
            refs.append(ref or "A")
            qrys.append(qry or "C")
        given = [reverse_complement(r) for r in refs] if rc else refs
        pa = PairAligner(e, flags, False, False, mo, 1, revcomp_ref=rc)
        # need = 1: every alignment matters (and reads of more than 160 bases take the fast pipeline as well)
        got = pa.locate_batch(given, qrys, need=[1] * len(refs), path="fast").tuples()
        nd = rng.choice([2, 10, top // 3, top // 2])
        part = pa.locate_batch(given, qrys, need=[nd] * len(refs), path="fast").tuples()
        if top <= 319:
            assert pa.locate_batch(given, qrys, path="wave").tuples() == got
        for ref, qry, g, h in zip(refs, qrys, got, part):
            exp = oracle.locate(ref, qry, e, flags, False, False, mo, 1)
            assert g == exp, (ref, qry, e, flags, mo, rc, g, exp)
            if exp is not None and exp[4] >= nd:
                assert h == exp, (ref, qry, e, flags, mo, rc, nd, h, exp)       # an alignment that reaches the bound is exact
            else:
                assert h is None or h[4] < nd, (nd, h, exp)
            total += 1
    return total


def check_long_reference_envelope(Aligner, PairAligner, oracle):
    """What the round-2 advisor found narrower on the long-reference path (129 .. 320 bases, per-pair aligner) than on
    aligner handles: adapters without indels (indel cost 100000) and reads with characters outside the IUPAC
    alphabet, which match nothing instead of raising; large indel costs on long pairs through every kernel family."""
    rng = random.Random(3)
    total = 0
    ref = rseq(rng, 150)
    for flags, ic in ((14, 1), (14, 100000), (11, 100000), (15, 2)):
        al = Aligner(ref, 0.1, flags, False, False, 3, ic)
        reads = [rseq(rng, 40) + ref[:rng.randint(5, 150)] for _ in range(20)]
        reads[3] = reads[3][:10] + "." + reads[3][11:]
        reads[7] = "#" + reads[7] + ".."
        reads[9] = reads[9].lower()
        got = al.locate_batch(reads).tuples()
        assert got == [oracle.locate(ref, q, 0.1, flags, False, False, 3, ic) for q in reads], (flags, ic)
        total += len(reads)
    for rnd in range(12):
        flags = rng.choice([15, 9, 14, 11, 8, 2, 0, 10])
        e, ic = rng.choice([0.1, 0.2, 0.3]), rng.choice([100000, 50, 7, 3])
        top = rng.choice([40, 150, 255, 320])
        if top > 255 and not (flags & 8):
            top = 255
        refs, qs = [], []
        for _ in range(16):
            m, n = rng.randint(0, top), rng.randint(0, top)
            frag = rseq(rng, m + n + 1)
            refs.append(frag[:m])
            qs.append(mutate(rng, frag[rng.randint(0, m):][:n], rng.choice([0, 0.05, 0.15]))[:n] if rng.random() < 0.7 else rseq(rng, n))
        pa = PairAligner(e, flags, False, False, rng.choice([1, 3]), ic)
        exp = [oracle.locate(r, q, e, flags, False, False, pa.min_overlap, ic) for r, q in zip(refs, qs)]
        for path in ("auto", "full", "fast") + (("wave",) if top <= 319 else ()):
            assert pa.locate_batch(refs, qs, path=path).tuples() == exp, (path, e, flags, ic, top)
        total += len(refs)
    return total


def check_multi_against_oracle(MultiAligner, oracle, seed, rounds, npairs=40, top=60):
    """MultiAligner.locate (no indels; up to max_matches hits) against the oracle: all 16 flag sets, error rates up to
    1, small alphabets (many hits: the max_matches cut and the perfect-hit break), unequal and empty sides."""
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        flags = rng.randint(0, 15) if rng.random() < 0.7 else 9
        e = rng.choice([0, 0.05, 0.1, 0.2, 0.3, 0.5, 1.0])
        mo, mm = rng.choice([1, 1, 3, 10]), rng.choice([1, 2, 5, 100])
        ma = MultiAligner(e, flags, mo)
        refs, qs = [], []
        for _p in range(npairs):
            m, n = rng.randint(0, top), rng.randint(0, top)
            alpha = rng.choice(["ACGT", "AC", "A", "ACGTN"])
            frag = rseq(rng, m + n + 2, alpha)
            kind = rng.random()
            if kind < 0.5:
                off = rng.randint(0, m)
                refs.append(frag[:m])
                qs.append(mutate(rng, frag[off:off + n], rng.choice([0, 0.05, 0.2]), alpha)[:n])
            elif kind < 0.7:
                refs.append(frag[:m])
                qs.append(frag[:m][:n] if rng.random() < 0.5 else frag[:m] * 2)
            else:
                refs.append(rseq(rng, m, alpha))
                qs.append(rseq(rng, n, alpha))
        got = ma.locate_batch(refs, qs, mm)
        for r, q, g in zip(refs, qs, got):
            assert g == oracle.multi_locate(r, q, e, flags, mo, mm), (r, q, e, flags, mo, mm, g)
            total += 1
    return total


def long_read_case(c):
    """The read of a tests/golden/long_reads.json.gz case: c["n"] random bases (seeded) with the pieces laid over."""
    rng = random.Random(c["seed"])
    q = list(rseq(rng, c["n"], c["alpha"]))
    for pos, text in c["pieces"]:
        pos = max(0, pos)
        for i, ch in enumerate(text):
            if pos + i < c["n"]:
                q[pos + i] = ch
    return "".join(q)


def check_golden_long_reads(Aligner, unsupported_exc, batch=True):
    """The reference's answers on reads of 737 .. 2 600 bases (long_reads.json.gz), per read and as batches."""
    cases = load_golden("long_reads.json.gz")
    checked = 0
    groups = {}
    for c in cases:
        groups.setdefault((c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]), []).append(c)
    for key, group in groups.items():
        try:
            al = Aligner(*key)
        except unsupported_exc:
            continue
        reads = [long_read_case(c) for c in group]
        got = al.locate_batch(reads).tuples() if batch else [al.locate(q) for q in reads]
        for c, g in zip(group, got):
            assert g == tup(c["out"]), (c, g)
            checked += 1
        assert al.locate(reads[0]) == tup(group[0]["out"])
    return checked


def check_long_reads(Aligner, oracle, unsupported_exc, seed, rounds, count=24, max_n=4000):
    """Reads beyond the batch pipelines' 736 bases (the reference has no length limit, _align.pyx:266-291): the
    full sweep with a rolling origin base.  Adapters whole, edited and cut, placed at the columns where the base
    moves (multiples of 256), at the read's ends and twice; every flag set; long and short reads in one list."""
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        m = rng.choice([rng.randint(1, 12), rng.randint(8, 40), rng.randint(30, 128)])
        ref = rseq(rng, m, "ACGT" if rng.random() < 0.8 else "ACGTN")
        flags = rng.choice([14, 14, 11, 15, 10, 6, 9, 1, 2, 4, 8, 0, 12, 3, 5, 7, 13])
        e = rng.choice([0, 0.05, 0.1, 0.1, 0.2, 0.3])
        ic = rng.choice([1, 1, 2, 100000])
        mo = rng.choice([1, 3, 5])
        wr, wq = rng.random() < 0.2, rng.random() < 0.2
        try:
            al = Aligner(ref, e, flags, wr, wq, mo, ic)
        except unsupported_exc:
            continue
        reads = []
        for _ in range(count):
            n = rng.choice([737, 767, 768, 769, 1023, 1024, 1025, rng.randint(737, 1300), rng.randint(737, max_n)])
            w = rng.random()
            part = mutate(rng, ref, rng.choice([0, 0, 0.04, 0.1, 0.2]))
            if w < 0.3:                                      # around a step of the base
                pos = rng.choice([256, 512, 768, 1024, 1280]) + rng.randint(-m - 3, 3)
                q = rseq(rng, max(0, pos)) + part + rseq(rng, n)
            elif w < 0.45:                                   # cut by the read end
                q = rseq(rng, n) + part[:rng.randint(1, len(part))] if part else rseq(rng, n)
                q = q[len(q) - n:]
            elif w < 0.55:                                   # at (or before) the read start
                q = part[rng.randint(0, max(0, len(part) - 1)):] + rseq(rng, n)
            elif w < 0.7:                                    # twice, the better one second or first
                a, b = mutate(rng, ref, 0.1), mutate(rng, ref, rng.choice([0, 0.1]))
                if rng.random() < 0.5:
                    a, b = b, a
                q = rseq(rng, rng.randint(0, n - 2 * m - 10)) + a + rseq(rng, rng.randint(0, 600)) + b + rseq(rng, n)
            elif w < 0.8:
                q = rseq(rng, rng.randint(0, n)) + part + rseq(rng, n)
            else:
                q = rseq(rng, n, "ACGTN" if rng.random() < 0.3 else "ACGT")
            reads.append(q[:n])
        if rng.random() < 0.5:                               # a list of long and short reads (split by the host)
            reads += [mutate(rng, rseq(rng, rng.randint(0, 200)) + ref, 0.05) for _ in range(6)]
            rng.shuffle(reads)
        got = al.locate_batch(reads).tuples()
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, e, flags, wr, wq, mo, ic), (ref, len(q), q, e, flags, wr, wq, mo, ic, g)
            total += 1
        q = reads[0]
        assert al.locate(q) == got[0]
    return total


def long_pair_case(c):
    """Reference and query of a tests/golden/long_pairs.json.gz case, rebuilt from its seed."""
    rng = random.Random(c["seed"])
    ref = rseq(rng, c["m"], c["alpha"])
    kind, n = c["kind"], c["n"]
    if kind == 0:
        q = (rseq(rng, c["a"]) + mutate(rng, ref, c["rate"]) + rseq(rng, n))[:n]
    elif kind == 1:
        q = mutate(rng, ref[c["a"]:], c["rate"])[:n] + rseq(rng, c["b"])
    elif kind == 2:
        q = (rseq(rng, n) + mutate(rng, ref, c["rate"])[:max(1, c["a"])])[-n:] if n else ""
    else:
        q = rseq(rng, n)
    return ref, q


def check_golden_long_pairs(Aligner, PairAligner, unsupported_exc):
    """The reference's answers for references of 321 .. 1 500 bases (long_pairs.json.gz) through Aligner (one long
    reference, a batch of reads) and through PairAligner (a reference per pair)."""
    cases = load_golden("long_pairs.json.gz")
    checked = 0
    by_setting = {}
    for c in cases:
        by_setting.setdefault((c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]), []).append(c)
    for (e, flags, wr, wq, mo, ic), group in by_setting.items():
        pairs = [long_pair_case(c) for c in group]
        got = PairAligner(e, flags, wr, wq, mo, ic).locate_batch([p[0] for p in pairs], [p[1] for p in pairs]).tuples()
        for c, g in zip(group, got):
            assert g == tup(c["out"]), (c, g)
            checked += 1
        ref, q = pairs[0]
        assert Aligner(ref, e, flags, wr, wq, mo, ic).locate(q) == tup(group[0]["out"])
    return checked


def check_long_pairs(Aligner, PairAligner, oracle, seed, rounds):
    """Pairs with a side beyond 320 bases (atr_locate_pairs_long_batch) and Aligner with a long reference against
    the oracle: all flag sets, indel costs, wildcard modes, ragged lengths on both sides."""
    rng = random.Random(seed)
    total = 0
    for _ in range(rounds):
        flags = rng.choice([15, 9, 14, 11, 8, 2, 0, 5, 10, 6, 3, 12, 7, 13, 1, 4])
        e = rng.choice([0, 0.05, 0.1, 0.2, 0.3])
        ic, mo = rng.choice([1, 1, 2, 100000]), rng.choice([1, 3, 10])
        wr, wq = rng.random() < 0.2, rng.random() < 0.2
        refs, qs = [], []
        for _ in range(12):
            m = rng.choice([rng.randint(321, 700), rng.randint(1, 320), rng.randint(256, 400)])
            n = rng.choice([rng.randint(321, 900), rng.randint(0, 320), rng.randint(600, 1500)])
            ref = rseq(rng, m, "ACGT" if rng.random() < 0.8 else "ACGTN")
            w = rng.random()
            if w < 0.4:
                q = (rseq(rng, rng.randint(0, 300)) + mutate(rng, ref, rng.choice([0, 0.03, 0.1])) + rseq(rng, n))[:n]
            elif w < 0.6:
                q = mutate(rng, ref[rng.randint(0, m - 1):], 0.05)[:n] + rseq(rng, rng.randint(0, 50))
            elif w < 0.8:
                q = (rseq(rng, n) + mutate(rng, ref, 0.05)[:rng.randint(1, m)])[-n:] if n else ""
            else:
                q = rseq(rng, n)
            refs.append(ref)
            qs.append(q)
        if max(max(map(len, refs)), max(map(len, qs))) <= 320:
            continue
        got = PairAligner(e, flags, wr, wq, mo, ic).locate_batch(refs, qs).tuples()
        for r, q, g in zip(refs, qs, got):
            assert g == oracle.locate(r, q, e, flags, wr, wq, mo, ic), (len(r), len(q), e, flags, wr, wq, mo, ic, g)
            total += 1
    for _ in range(max(1, rounds // 5)):
        m = rng.randint(321, 1200)
        ref = rseq(rng, m)
        flags = rng.choice([14, 15, 11, 10])
        al = Aligner(ref, 0.1, flags, False, False, 3, 1)
        reads = [(rseq(rng, rng.randint(0, 100)) + mutate(rng, ref, 0.05)[rng.randint(0, m // 2):rng.randint(m // 2, m)]
                  + rseq(rng, 100))[:rng.randint(50, 700)] for _ in range(10)]
        got = al.locate_batch(reads).tuples()
        for q, g in zip(reads, got):
            assert g == oracle.locate(ref, q, 0.1, flags, False, False, 3, 1), (m, len(q), g)
            total += 1
        assert al.locate(reads[0]) == got[0]
    return total


def check_insert_list_cap(oracle):
    """Eight-chunk reads list 12 overlap lengths per pair (round 6: the LDS of a fourth block per CU), five-chunk reads 8
    (a sixth), the others 16; a pair with more takes the ordered redo.  Periodic reads (a short unit repeated, a few substitutions) pass the probe at
    every multiple of the period: pairs with 5 .. 40 listed lengths, reads of 225 .. 256 bases and, for the other
    cap, 129 .. 160 -- the same records as the checker either way."""
    import random
    from atropos_amd import synth
    from atropos_amd.align import InsertAligner
    from atropos_amd.util import reverse_complement
    rng = random.Random(4)
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    orc = oracle.InsertOracle(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    total = 0
    for n in (250, 256, 225, 150, 160):
        r1s, r2s = [], []
        for period in (6, 9, 13, 17, 21, 29, 40, 55):
            for _ in range(4):
                unit = "".join(rng.choice("ACGT") for _ in range(period))
                frag = (unit * (n // period + 2))[:n]
                r1 = "".join(rng.choice("ACGT") if rng.random() < 0.02 else c for c in frag)
                r2 = "".join(rng.choice("ACGT") if rng.random() < 0.02 else c for c in reverse_complement(frag))
                r1s.append(r1)
                r2s.append(r2)
        got = ia.match_insert_batch(r1s, r2s).results()
        for x, y, g in zip(r1s, r2s, got):
            exp = orc.match_insert(x, y)
            exp = None if exp is None else [list(exp[0]), None if exp[1] is None else list(exp[1]),
                                            None if exp[2] is None else list(exp[2])]
            assert norm_insert(g) == exp, (n, x, y)
            total += 1
    return total
