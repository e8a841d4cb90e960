#!/usr/bin/env python3
"""Round-2 golden vectors, generated from the REFERENCE itself (build container only: needs
/root/reference; the reference package is copied to a scratch directory and its Cython modules are
built there by tests/golden/make_golden.build_reference -- nothing of it enters this repository).

    python tests/golden/make_round2_golden.py

Fixtures (inputs are generated here; expected values are what the reference returned):
    linked_fuzz.json.gz    sets of linked adapters (`-a ^FRONT...BACK`): per read what
                           AdapterCutter._best_match / LinkedAdapter.match_to do -- which adapter, how many
                           5' parts match (> 1: the reference raises AttributeError), both Match records
    info_records.json.gz   AdapterCutter runs: the MatchInfo rows (Match.get_info_record) of every read
    dpmatrix.json.gz       str(aligner.dpmatrix) after enable_debug() + locate(), with the result tuple
    insert_long.json.gz    InsertAligner.match_insert on MiSeq-length pairs (2 x 257 .. 320 bp)
    long_reference.json.gz Aligner.locate with references of 129 .. 320 bases (all adapter flag sets)
    c5_head.json.gz        the first pairs of BASELINE config C5 (2 x 250 bp, qualities) through
                           InsertAdapterCutter(mismatch_action='liberal', read wildcards): full outputs of
                           the first pairs, a digest of every pair
While doing so it pins the oracle's C restatement of Adapter.match_to / LinkedAdapter.match_to
(oracle.match_to, oracle.linked_many): every case must agree, or the script aborts.
"""
import hashlib
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import build_reference, dump, match_fields  # noqa: E402


def pair_digest(state):
    text = repr(state).encode()
    return hashlib.sha256(text).hexdigest()[:16]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. "insert_long": just that fixture
    build_reference("/tmp/atropos_ref_build")
    import numpy as np
    from atropos.adapters import Adapter, LinkedAdapter, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE
    from atropos.commands.trim.modifiers import AdapterCutter, InsertAdapterCutter
    from atropos.io.seqio import Sequence
    from atropos.util import RandomMatchProbability
    from atropos_amd import synth
    from oracle import oracle as O
    O.build(force=True)
    rng = random.Random(20260928)

    def rseq(n, alpha="ACGT"):
        return "".join(rng.choice(alpha) for _ in range(n))

    def mutate(s, p, alpha="ACGT"):
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

    # ------------------------------------------------------------------ linked adapter sets
    def linked_case(n_reads):
        na = rng.randint(1, 4)
        e = rng.choice([0.05, 0.1, 0.12, 0.2])
        kw = dict(max_error_rate=e, min_overlap=rng.choice([1, 3, 5]), indel_cost=rng.choice([1, 1, 2]),
                  read_wildcards=rng.random() < 0.15)
        with_n = rng.random() < 0.2
        alpha = "ACGTN" if with_n else "ACGT"
        fl = rng.randint(6, 24)
        fronts = [rseq(fl if rng.random() < 0.6 else rng.randint(6, 24), alpha) for _ in range(na)]
        backs = [rseq(rng.choice([rng.randint(8, 32), rng.randint(33, 40), rng.randint(41, 64)]), alpha) for _ in range(na)]
        n = rng.randint(40, 160)
        reads = []
        for _ in range(n_reads):
            a = rng.randrange(na)
            w = rng.random()
            head = (fronts[a] if w < 0.35 else mutate(fronts[a], rng.choice([0.03, 0.08, 0.15])) if w < 0.7 else
                    rseq(rng.randint(0, 3)) + fronts[a] if w < 0.75 else fronts[a] + fronts[(a + 1) % na] if w < 0.8 else "")
            head = head.replace("N", "A")
            b = backs[a if rng.random() < 0.85 else rng.randrange(na)].replace("N", "C")
            v = rng.random()
            tail = (b if v < 0.3 else mutate(b, rng.choice([0.03, 0.08, 0.15])) if v < 0.6 else
                    b[:rng.randint(1, len(b))] if v < 0.8 else "")
            q = (head + rseq(rng.randint(0, n), "ACGT" if rng.random() < 0.9 else "ACGTN") + tail + rseq(rng.randint(0, 12)))[:n + 30]
            if rng.random() < 0.1:
                q = q.lower()
            reads.append(q or "A")
        return dict(fronts=fronts, backs=backs, kw=kw, reads=reads)

    def run_linked(c):
        las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, **c["kw"]) for f, b in zip(c["fronts"], c["backs"])]
        outs = []
        for q in c["reads"]:
            which, count, fm, bm = -1, 0, None, None
            for a, la in enumerate(las):
                lm = la.match_to(Sequence(name="r", sequence=q))
                if lm is None:
                    continue
                count += 1
                if which < 0:
                    which, fm, bm = a, match_fields(lm.front_match), match_fields(lm.back_match)
            outs.append([which, count, fm, bm])
        return outs

    def check_oracle_linked(c, outs):
        kw = c["kw"]
        width = max(len(q) for q in c["reads"])
        mat = np.zeros((len(c["reads"]), width), np.uint8)
        for i, q in enumerate(c["reads"]):
            mat[i, :len(q)] = np.frombuffer(q.encode(), np.uint8)
        lens = np.array([len(q) for q in c["reads"]], np.int32)
        wh, f, b = O.linked_many(c["fronts"], c["backs"], mat, lens, kw["max_error_rate"], kw["min_overlap"], kw["indel_cost"],
                                 True, kw["read_wildcards"], 2)
        for i, (which, count, fm, bm) in enumerate(outs):
            got = [int(wh[i, 0]), int(wh[i, 1]), None if f[i, 1] < 0 else [int(v) for v in f[i]],
                   None if b[i, 1] < 0 else [int(v) for v in b[i]]]
            assert got == [which, count, fm, bm], (c["fronts"], c["backs"], kw, c["reads"][i], got, [which, count, fm, bm])

    cases = []
    for _ in range(160 if only is None else 0):
        c = linked_case(24)
        c["out"] = run_linked(c)
        check_oracle_linked(c, c["out"])
        cases.append(c)
    if only is None:
        dump("linked_fuzz.json.gz", cases)
    extra = 0
    for _ in range(1200 if only is None else 0):                                  # uncommitted: oracle vs reference only
        c = linked_case(40)
        check_oracle_linked(c, run_linked(c))
        extra += len(c["reads"])
    print("oracle.linked_many == reference on %d committed + %d further reads" % (sum(len(c["reads"]) for c in cases), extra))

    # ------------------------------------------------------------------ info records
    info_cases = []
    for _ in range(120 if only is None else 0):
        nad = rng.choice([1, 2, 3])
        specs = []
        for a in range(nad):
            specs.append(dict(seq=rseq(rng.randint(6, 34), "ACGT" if rng.random() < 0.8 else "ACGTN"),
                              where=rng.choice([BACK, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE]), name="ad%d" % a))
        kw = dict(max_error_rate=rng.choice([0.1, 0.12, 0.2]), min_overlap=rng.choice([1, 3, 5]))
        times, action = rng.choice([1, 1, 2, 3]), rng.choice(['trim', 'trim', 'mask', None])
        reads = []
        for k in range(10):
            sp = rng.choice(specs)
            a = mutate(sp["seq"].replace("N", "C"), rng.choice([0, 0, 0.05, 0.1]))
            body = rseq(rng.randint(20, 90))
            q = body + a + rseq(rng.randint(0, 12)) if sp["where"] in (BACK, SUFFIX, ANYWHERE) and rng.random() < 0.7 else a + body
            if rng.random() < 0.3:
                q += rng.choice(specs)["seq"].replace("N", "G")
            qual = None if rng.random() < 0.2 else "".join(chr(rng.randint(35, 73)) for _ in q)
            reads.append(["read%d" % k, q, qual])
        cutter = AdapterCutter([Adapter(sp["seq"], sp["where"], name=sp["name"], **kw) for sp in specs], times=times, action=action)
        outs = []
        for name, q, qual in reads:
            r = cutter(Sequence(name=name, sequence=q, qualities=qual))
            outs.append(None if not r.match_info else [list(info) for info in r.match_info])
        info_cases.append(dict(specs=specs, kw=kw, times=times, action=action, reads=reads, out=outs))
    if only is None:
        dump("info_records.json.gz", info_cases)

    # ------------------------------------------------------------------ Aligner.enable_debug() / dpmatrix
    if only in (None, "dpmatrix"):
        from atropos.align import Aligner
        dp_cases = []
        for it in range(90):
            m = rng.randint(1, 18)
            ref = rseq(m, "ACGT" if rng.random() < 0.7 else "ACGTNRY")
            flags = rng.randint(0, 15) if rng.random() < 0.5 else rng.choice([14, 11, 15, 9, 1, 8])
            e = rng.choice([0, 0.1, 0.2, 0.25, 0.34, 0.5])
            wr, wq = rng.random() < 0.25, rng.random() < 0.25
            mo, ic = rng.choice([1, 1, 3]), rng.choice([1, 1, 1, 2, 3])
            kind = rng.random()
            query = (rseq(rng.randint(0, 12)) + mutate(ref.replace("N", "A").replace("R", "G").replace("Y", "C"), 0.15) +
                     rseq(rng.randint(0, 12))) if kind < 0.6 else rseq(rng.randint(0, 30))
            if kind > 0.9:
                query = rseq(rng.randint(0, 5)) + ref[:rng.randint(1, m)]
            al = Aligner(ref, e, flags, wr, wq, mo)
            al.indel_cost = ic
            al.enable_debug()
            res = al.locate(query)
            dp_cases.append(dict(ref=ref, query=query, e=e, flags=flags, wr=wr, wq=wq, mo=mo, ic=ic,
                                 out=None if res is None else list(res), matrix=str(al.dpmatrix)))
        dump("dpmatrix.json.gz", dp_cases)

    # ------------------------------------------------------------------ match_insert on 2 x 300 bp pairs
    if only in (None, "insert_long"):
        from atropos.align import InsertAligner
        from atropos.util import reverse_complement
        A1, A2 = synth.PE_ADAPTER1, synth.PE_ADAPTER2

        def noise(s, p):
            return "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in s)

        long_cases, checked = [], 0
        cfgs = [dict(), dict(max_insert_mismatch_frac=0.1, max_adapter_mismatch_frac=0.1), dict(read_wildcards=True),
                dict(adapter_wildcards=False), dict(min_insert_overlap=5, min_adapter_overlap=3),
                dict(insert_max_rmp=1e-3, adapter_max_rmp=1e-2)]
        for cfg in cfgs:
            ref = InsertAligner(A1, A2, **cfg)
            orc = O.InsertOracle(A1, A2, **cfg)
            for it in range(160):
                n = rng.choice([257, 288, 289, 300, 300, 319, 320])
                f = rng.randint(0, int(1.5 * n))
                F = rseq(f)
                r1 = (F + A1 + rseq(n))[:rng.choice([n, n, n - 3, n - 40])]
                r2 = (reverse_complement(F) + A2 + rseq(n))[:rng.choice([n, n, n - 1, n - 70])]
                p = rng.choice([0, 0.01, 0.03, 0.1])
                r1, r2 = noise(r1, p), noise(r2, p)
                if rng.random() < 0.08:
                    r1 = rseq(n)
                if rng.random() < 0.03:
                    r1, r2 = "A" * len(r1), "T" * len(r2)
                res = ref.match_insert(r1, r2)
                out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
                mine = orc.match_insert(r1, r2)
                mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                                  None if mine[2] is None else list(mine[2])]
                assert mine == out, (cfg, r1, r2, mine, out)
                checked += 1
                if it < 40:
                    long_cases.append(dict(a1=A1, a2=A2, r1=r1, r2=r2, kw=cfg, out=out))
        # adapters of more than 64 bases (an indexed TruSeq adapter has 66): overhangs compared beyond base 64
        B1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC" + "ATCACGAT" + "ATCTCGTATGCCGTCTTCTGCTTG"
        B2 = A2 + rseq(42)
        assert len(B1) == 66 and len(B2) == 100
        for cfg in cfgs[:4]:
            ref = InsertAligner(B1, B2, **cfg)
            orc = O.InsertOracle(B1, B2, **cfg)
            for it in range(120):
                n = rng.choice([100, 150, 150, 250, 300])
                f = rng.randint(0, n)                       # overhangs of up to n bases
                F = rseq(f)
                t1, t2 = mutate(B1, rng.choice([0, 0.03, 0.1])), mutate(B2, rng.choice([0, 0.03, 0.1]))
                r1 = (F + t1 + rseq(n))[:n]
                r2 = (reverse_complement(F) + t2 + rseq(n))[:rng.choice([n, n, n - 2])]
                p = rng.choice([0, 0.01, 0.03])
                r1, r2 = noise(r1, p), noise(r2, p)
                res = ref.match_insert(r1, r2)
                out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
                mine = orc.match_insert(r1, r2)
                mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                                  None if mine[2] is None else list(mine[2])]
                assert mine == out, (cfg, r1, r2, mine, out)
                checked += 1
                if it < 30:
                    long_cases.append(dict(a1=B1, a2=B2, r1=r1, r2=r2, kw=cfg, out=out))
        # soft-masked reads: the insert compare tells the cases apart, the adapter compares fold them when a
        # wildcard flag is set (compare_prefixes translates), and compare literally when none is
        def soften(s):
            kind = rng.random()
            if kind < 0.3 or not s:
                return s
            if kind < 0.4:
                return s.lower()
            a = rng.randrange(len(s)); b = rng.randint(a, len(s))
            return s[:a] + s[a:b].lower() + s[b:]
        soft_cfgs = [dict(), dict(read_wildcards=True), dict(adapter_wildcards=False),
                     dict(adapter_wildcards=False, read_wildcards=True), dict(max_insert_mismatch_frac=0.3, min_adapter_overlap=3)]
        for cfg in soft_cfgs:
            ref = InsertAligner(A1, A2, **cfg)
            orc = O.InsertOracle(A1, A2, **cfg)
            for it in range(150):
                n = rng.choice([40, 75, 100, 150])
                f = rng.randint(0, int(1.3 * n))
                F = rseq(f)
                r1 = (F + A1 + rseq(n))[:n]
                r2 = (reverse_complement(F) + A2 + rseq(n))[:rng.choice([n, n, n - 4])]
                p = rng.choice([0, 0.01, 0.05])
                r1, r2 = noise(r1, p), noise(r2, p)
                if rng.random() < 0.5:                     # the same stretch of the insert lower-case in both reads
                    a = rng.randint(0, max(0, min(f, n) - 1)); b = rng.randint(a, min(f, n))
                    r1 = r1[:a] + r1[a:b].lower() + r1[b:]
                    lo, hi = max(0, f - b), max(0, f - a)
                    r2 = r2[:lo] + r2[lo:hi].lower() + r2[hi:]
                r1, r2 = soften(r1), soften(r2)
                res = ref.match_insert(r1, r2)
                out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
                mine = orc.match_insert(r1, r2)
                mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                                  None if mine[2] is None else list(mine[2])]
                assert mine == out, (cfg, r1, r2, mine, out)
                checked += 1
                if it < 40:
                    long_cases.append(dict(a1=A1, a2=A2, r1=r1, r2=r2, kw=cfg, out=out))
        dump("insert_long.json.gz", long_cases)
        print("oracle.match_insert == reference on %d pairs (257 .. 320 bp reads; adapters of 66 / 100 bases; soft-masked reads)" % checked)
    # ------------------------------------------------------------------ Aligner with a reference of 129 .. 320 bases
    if only in (None, "long_reference"):
        from atropos.align import Aligner
        lr_cases = []
        for it in range(220):
            m = rng.choice([129, 130, 150, 160, 200, 255, 256, 257, 300, 320]) if it % 3 else rng.randint(129, 320)
            alpha = "ACGT" if rng.random() < 0.7 else "ACGTN"
            ref = rseq(m, alpha)
            flags = rng.choice([14, 11, 15, 8, 2, 9, 15, 14])
            e = rng.choice([0, 0.02, 0.05, 0.1, 0.2])
            wr, wq = rng.random() < 0.3, rng.random() < 0.3
            if m > 255 and not flags & 8:
                flags |= 8                                   # the device envelope above 255 rows: STOP_WITHIN_SEQ2
            mo, ic = rng.choice([1, 1, 3, 20]), rng.choice([1, 1, 1, 2])
            n = rng.choice([0, 30, 100, 150, 250, 300, 320])
            kind = rng.random()
            core = ref.replace("N", "A")
            if kind < 0.35:                                  # the reference's head at the read end (3' adapter)
                keep = rng.randint(1, min(m, n)) if n else 0
                query = (rseq(n) + mutate(core[:keep], 0.03))[-n:] if n else ""
            elif kind < 0.6:                                 # its tail at the read start (5' adapter)
                keep = rng.randint(1, min(m, n)) if n else 0
                query = (mutate(core[m - keep:], 0.03) + rseq(n))[:n]
            elif kind < 0.8:                                 # the read inside the reference
                at = rng.randint(0, max(0, m - n))
                query = mutate(core[at:at + n], 0.04)
            else:
                query = rseq(n, "ACGTN" if wq else "ACGT")
            al = Aligner(ref, e, flags, wr, wq, mo)
            al.indel_cost = ic
            res = al.locate(query)
            lr_cases.append(dict(ref=ref, query=query, e=e, flags=flags, wr=wr, wq=wq, mo=mo, ic=ic,
                                 out=None if res is None else list(res)))
        dump("long_reference.json.gz", lr_cases)
        print("long_reference: %d cases, %d with a match" % (len(lr_cases), sum(c["out"] is not None for c in lr_cases)))
    if only is not None:
        return

    # ------------------------------------------------------------------ C5 head through InsertAdapterCutter
    count, full = 2048, 96
    w = synth.workload("C5", 0, count)

    def rows(t):
        return [bytes(x.tolist()).decode("ascii") for x in t]

    rmp = RandomMatchProbability()
    akw = dict(max_error_rate=0.2, min_overlap=1, indel_cost=3, match_probability=rmp, max_rmp=1e-6, read_wildcards=True)
    cutter = InsertAdapterCutter(Adapter(synth.PE_ADAPTER1, BACK, name="a1", **akw), Adapter(synth.PE_ADAPTER2, BACK, name="a2", **akw),
                                 action='trim', mismatch_action='liberal', read_wildcards=True)
    states, digests, text_digests = [], [], []
    for r1, q1, r2, q2 in zip(rows(w["reads1"]), rows(w["quals1"]), rows(w["reads2"]), rows(w["quals2"])):
        a, b = cutter(Sequence(name="p", sequence=r1, qualities=q1), Sequence(name="p", sequence=r2, qualities=q2))
        st = [[x.sequence, x.qualities, int(x.corrected), bool(x.insert_overlap),
               match_fields(x.match) if x.match is not None else None] for x in (a, b)]
        states.append(st)
        digests.append(pair_digest(st))
        text_digests.append(pair_digest([st[0][0], st[0][1], st[1][0], st[1][1]]))      # what the output files hold
    dump("c5_head.json.gz", dict(count=count, adapter_kw={k: v for k, v in akw.items() if k != "match_probability"},
                                 full=states[:full], digests=digests, text_digests=text_digests, with_adapters=list(cutter.with_adapters),
                                 corrected_pairs=cutter.corrected_pairs, corrected_bp=list(cutter.corrected_bp)))


if __name__ == "__main__":
    main()
