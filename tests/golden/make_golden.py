#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container, where /root/reference exists: it copies the
reference's Python package into a scratch directory, builds its three Cython
modules there (nothing of the reference is written into this repository), imports
it, and records inputs + the reference's outputs as small data fixtures.  While
doing so it also pins the CPU oracle (oracle/align_oracle.c): every case - the
committed ones and a much larger uncommitted fuzz - must agree tuple-for-tuple,
otherwise the script aborts.

    python tests/golden/make_golden.py            # regenerate all fixtures

Fixtures (all inputs are generated here or transcribed as data from the
reference's own known-answer tests, tests/test_align.py, tests/test_adapters.py,
tests/test_modifiers.py):
    kats.json              known answers the reference's tests assert, re-evaluated
    locate_fuzz.json.gz    Aligner.locate over all flag sets / error rates / indel costs
    multi_fuzz.json.gz     MultiAligner.locate
    prefix_fuzz.json.gz    compare_prefixes / compare_suffixes
    insert_fuzz.json.gz    InsertAligner.match_insert
    match_to_fuzz.json.gz  Adapter.match_to (boundary object, incl. exact shortcut)
    synth_heads.json.gz    first reads of workloads C1..C5 with reference outputs
    cutter_fuzz.json.gz    AdapterCutter (several adapters, times, trim/mask/None)
    insert_cutter_fuzz.json.gz  InsertAdapterCutter incl. error correction (N/liberal/conservative)
    caller_kats.json       known answers of the reference's caller tests
    rmp.json               RandomMatchProbability values
"""
import argparse
import gzip
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference"


def build_reference(scratch):
    if not os.path.exists(os.path.join(scratch, "atropos", "align")) or not any(
            f.startswith("_align.") and f.endswith(".so") for f in os.listdir(os.path.join(scratch, "atropos", "align"))):
        os.makedirs(scratch, exist_ok=True)
        for item in ("atropos", "setup.py", "versioneer.py", "setup.cfg", "README.md"):
            src = os.path.join(REF_SRC, item)
            dst = os.path.join(scratch, item)
            if os.path.isdir(src):
                shutil.copytree(src, dst, dirs_exist_ok=True)
            else:
                shutil.copy(src, dst)
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "-i"], cwd=scratch,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)


def dump(name, obj):
    path = os.path.join(HERE, name)
    data = json.dumps(obj, separators=(",", ":")).encode()
    if name.endswith(".gz"):
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(data)
    else:
        with open(path, "wb") as fh:
            fh.write(data)
    print("%-24s %8d bytes" % (name, os.path.getsize(path)))


def match_fields(m):
    return None if m is None else [m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/atropos_ref_build")
    ap.add_argument("--big-fuzz", type=int, default=60000, help="uncommitted oracle-vs-reference cases")
    args = ap.parse_args()
    build_reference(args.scratch)
    sys.path.insert(0, ROOT)

    from atropos.align import (Aligner, MultiAligner, compare_prefixes, compare_suffixes,
                               InsertAligner, locate)
    from atropos.adapters import Adapter, LinkedAdapter, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE
    from atropos.io.seqio import Sequence
    from atropos.util import RandomMatchProbability, reverse_complement
    from oracle import oracle as O
    from atropos_amd import synth

    rng = random.Random(0xA7205)

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

    # ------------------------------------------------------------------ KATs
    kats = {"locate": [], "compare_prefixes": [], "compare_suffixes": [], "multi_locate": [],
            "match_insert": [], "match_to": []}

    def kat_locate(ref, query, e, flags, wr=False, wq=False, mo=1, expect="?"):
        got = locate(ref, query, e, flags, wr, wq, mo)
        if expect != "?":
            assert got == expect, (ref, query, got, expect)
        assert O.locate(ref, query, e, flags, wr, wq, mo, 1) == got
        kats["locate"].append(dict(ref=ref, query=query, e=e, flags=flags, wr=wr, wq=wq, mo=mo, ic=1,
                                   out=got))

    WS = ["CCCATTGATC", "CCCRTTRATC", "YCCATYGATC", "CSSATTSATC", "CCCWWWGATC", "CCCATKKATC", "CCMATTGMTC",
          "BCCATTBABC", "BCCATTBABC", "CCCDTTDADC", "CHCATHGATC", "CVCVTTVATC", "CCNATNGATC", "CCCNTTNATC"]
    # reference tests/test_align.py:24-29, :98-132 and SURVEY section 8(c) probes
    kat_locate("A" * 17, "ACAG" + "A" * 42, 0.0, BACK, expect=(0, 17, 4, 21, 17, 0))
    kat_locate("CTCCAGCTTAGACATATC", "CC", 0.1, BACK)
    kat_locate("GCTTAGACATATC", "CAA", 1.0, BACK)
    rd = "CATCTGTCC" + WS[0] + "GCCAGGGTTGATTCGGCTGATCTGGCCG"
    for a in WS:
        kat_locate(a, rd, 0.0, BACK, wr=True, expect=(0, 10, 9, 19, 10, 0))
    kat_locate("CCCXTTXATC", rd, 0.0, BACK, wr=True, expect=None)
    for s in WS:
        kat_locate(WS[0], "CATCTGTCC" + s + "GCCAGGGTTGATTCGGCTGATCTGGCCG", 0.0, BACK, wq=True,
                   expect=(0, 10, 9, 19, 10, 0))
    for a in WS:
        for s in WS:
            kat_locate(a, "CATCTGTCC" + s + "GCCAGGGTTGATTCGGCTGATCTGGCCG", 0.0, BACK, wr=True, wq=True,
                       expect=(0, 10, 9, 19, 10, 0))
    kat_locate("CTGATCTGGCCG", "AAAAGGG", 0.1, BACK, expect=None)
    kat_locate("TCGTATGCCGTCTTC", "TCGTATGCCCTCC", 0.2, BACK, expect=(0, 15, 0, 12, 12, 3))
    kat_locate("GAGATTGCCA", "TTGCCAACGTACGT", 0.1, FRONT, expect=(4, 10, 0, 6, 6, 0))
    kat_locate("FRONTADAPT", "FRONTADAPXTTTT", 0.1, PREFIX, expect=(0, 10, 0, 11, 10, 1))
    kat_locate("BACKADAPTER", "CCCCBACKADAPTE", 0.1, SUFFIX, expect=(0, 11, 4, 14, 10, 1))
    kat_locate("TTAGACATATCTCCGTCG", "ATCTCCGTCGAAAAAAA", 0.1, ANYWHERE, expect=(8, 18, 0, 10, 10, 0))

    def kat_cp(fn, name, a, b, wr=False, wq=False, expect="?"):
        got = fn(a, b, wr, wq)
        if expect != "?":
            assert got == expect, (name, a, b, got, expect)
        assert getattr(O, name)(a, b, wr, wq) == got
        kats[name].append(dict(ref=a, query=b, wr=wr, wq=wq, out=got))

    # reference tests/test_align.py:55-95
    kat_cp(compare_prefixes, "compare_prefixes", "AAXAA", "AAAAATTTTTTTTT", expect=(0, 5, 0, 5, 4, 1))
    kat_cp(compare_prefixes, "compare_prefixes", "AANAA", "AACAATTTTTTTTT", wr=True, expect=(0, 5, 0, 5, 5, 0))
    kat_cp(compare_prefixes, "compare_prefixes", "XAAAAA", "AAAAATTTTTTTTT", expect=(0, 6, 0, 6, 4, 2))
    for s in WS:
        r = s + "GCCAGGGTTGATTCGGCTGATCTGGCCG"
        kat_cp(compare_prefixes, "compare_prefixes", WS[0], r, wq=True, expect=(0, 10, 0, 10, 10, 0))
        kat_cp(compare_prefixes, "compare_prefixes", r, WS[0], wr=True, expect=(0, 10, 0, 10, 10, 0))
        kat_cp(compare_prefixes, "compare_prefixes", s, s + "GCCAGGG", expect=(0, 10, 0, 10, 10, 0))
        kat_cp(compare_prefixes, "compare_prefixes", s + "GCCAGGG", s, wr=True, wq=True,
               expect=(0, 10, 0, 10, 10, 0))
    for wr in (False, True):
        for wq in (False, True):
            kat_cp(compare_prefixes, "compare_prefixes", "CCCXTTXATC", WS[0] + "GCCAGG", wr, wq,
                   expect=(0, 10, 0, 10, 8, 2))
    kat_cp(compare_prefixes, "compare_prefixes", "TTNGACATAT", "TTAGACATATGG", True, False,
           expect=(0, 10, 0, 10, 10, 0))
    kat_cp(compare_suffixes, "compare_suffixes", "AAXAA", "TTTTTTTAAAAA", expect=(0, 5, 7, 12, 4, 1))
    kat_cp(compare_suffixes, "compare_suffixes", "AANAA", "TTTTTTTAACAA", wr=True, expect=(0, 5, 7, 12, 5, 0))
    kat_cp(compare_suffixes, "compare_suffixes", "AAAAAX", "TTTTTTTAAAAA", expect=(0, 6, 6, 12, 4, 2))

    def kat_multi(ref, query, e, flags, mo, mx=100, expect_len=None):
        got = MultiAligner(e, flags, mo).locate(ref, query, mx)
        if expect_len is not None:
            assert len(got) == expect_len
        assert O.multi_locate(ref, query, e, flags, mo, mx) == got, (ref, query)
        kats["multi_locate"].append(dict(ref=ref, query=query, e=e, flags=flags, mo=mo, mx=mx, out=got))
        return got

    # reference tests/test_align.py:195-234
    g = kat_multi("AGAGATCAGATGACAGATC", "GATCA", 0, 15, 3, expect_len=2)
    assert sorted(g, key=lambda x: -x[4]) == [(3, 8, 0, 5, 5, 0), (15, 19, 0, 4, 4, 0)]
    g = kat_multi("GATATCAGATGACAGATCAGAGATCAGAT", "GAGATCAGATGA", 0.1, 15, 10, expect_len=2)
    assert sorted(g, key=lambda x: x[5]) == [(19, 29, 0, 10, 10, 0), (0, 12, 0, 12, 11, 1)]
    # duplicate emission of the full-length hit with the insert aligner's flags
    g = kat_multi("ACGTACGTTGCAATC", "ACGTACGTTGCAATG", 0.2, 9, 1)
    assert g.count(g[-1]) == 2 and g[-1][3] == 15
    kat_multi("ACGTACGTTGCAATC", "ACGTACGTTGCAATC", 0.2, 9, 1, expect_len=1)

    def kat_insert(a1, a2, r1, r2, **kw):
        res = InsertAligner(a1, a2, **kw).match_insert(r1, r2)
        out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
        mine = O.InsertOracle(a1, a2, **kw).match_insert(r1, r2)
        mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                          None if mine[2] is None else list(mine[2])]
        assert mine == out, (r1, r2, mine, out)
        kats["match_insert"].append(dict(a1=a1, a2=a2, r1=r1, r2=r2, kw=kw, out=out))
        return res

    # reference tests/test_align.py:156-191, tests/test_modifiers.py:295-324,444-471
    res = kat_insert("TTAGACATATGG", "CAGTGGAGTATA", "AGTCGAGCCCATTGCAGACT" + "TTAGACATAT",
                     "AGTCTGCAATGGGCTCGACT" + "CAGTGGAGTA")
    assert (res[1].rstart, res[1].length, res[2].rstart, res[2].length) == (20, 10, 20, 10)
    res = kat_insert("TTAGACATAT", "CAGTGGAGTA", "GACAGGCCGTTTGAATGTTGACGGGATGTT", "CATCCCGTCAACATTCAAACGGCCTGTCCA")
    assert (res[1].rstart, res[1].length, res[2].rstart, res[2].length) == (28, 2, 28, 2)
    A1 = synth.PE_ADAPTER1
    A2 = synth.PE_ADAPTER2
    kat_insert(A1, A2,
               "TTTGCAGCTTTTGTAGACAAGTGCTGTGCAGCTGATGTCAAAGAGACCTGCTTTGCTCTGGAGGGTCCAAAACTTGTAGCCTCAACCCGAGAAGCCATAGCCTAA",
               "ATAGGCTATGGCTTCTCGAGTTGAAGCTACAAGTTTTGGACCCTCCAGAGCAAAGCAGGTCTCTTTGACATCAGCTGCACAGCACTTGTCTACAAAAGCTGCAAAAGATCGGAAGAGCGTCTCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTCGACGTATCATTAAAAAAAAAAACACATCACATCAACAAGATAACACGACTTCTCCATCCACAGTACCGATGACCTCAACATTAGT")
    kat_insert(A1, A2,
               "CTGGGCTGGGATGCCTATCCCTCAGTTGAGGCTTACACATTTATTTTCATGTATTGTGGTATTACTTCGCTGTGTATAAAGTATAGATCGGAAGAGCACACGTCTGAACTCCAGTCACTGACCAATCTCGT",
               "ATACTTTATACACAGCGAAGTAATACCACAATACATGAAAATAAATGTGTAAGCCTCAACTGAGGGATAGGCATCCCAGCCCAGAGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTC")

    def kat_match_to(seq, where, read, **kw):
        ad = Adapter(seq, where, **kw)
        m = ad.match_to(Sequence(name="r", sequence=read))
        kats["match_to"].append(dict(seq=seq, where=where, read=read, kw=kw, out=match_fields(m)))
        return m

    # reference tests/test_adapters.py:44-68
    m = kat_match_to("TCGTATGCCGTCTTC", BACK, "TCGTATGCCCTCC", max_error_rate=0.2, min_overlap=3,
                     read_wildcards=False, adapter_wildcards=False)
    assert (m.errors, m.astart, m.astop) == (3, 0, 15)
    kat_match_to("ACGT", BACK, "TTACGT", max_error_rate=0.1)
    dump("kats.json", kats)

    # ------------------------------------------------------------------ locate fuzz
    FLAGS = [14, 11, 8, 2, 15, 9, 0, 5, 10, 1, 4, 3, 6, 7, 12, 13]

    def gen_locate_case():
        m = rng.randint(1, 64) if rng.random() < 0.8 else rng.randint(1, 8)
        ref = rseq(m, "ACGT" if rng.random() < 0.7 else "ACGTNRYKMSWBDHVX")
        flags = rng.choice(FLAGS[:5]) if rng.random() < 0.7 else rng.choice(FLAGS)
        e = rng.choice([0, 0.05, 0.1, 0.12, 0.2, 0.3, 1 / 3, 0.5])
        ic = rng.choice([1, 1, 1, 2, 3, 100000])
        mo = rng.choice([1, 3, 5, 10])
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        mode = rng.random()
        if mode < 0.65:
            a = mutate(ref, rng.choice([0, 0.03, 0.08, 0.15]))
            cut, pos = rng.randint(0, len(a)), rng.randint(0, 120)
            w = rng.random()
            q = (rseq(pos) + a + rseq(rng.randint(0, 30)) if w < 0.4 else
                 a[cut:] + rseq(pos) if w < 0.7 else rseq(pos) + a[:cut])
            q = q[:rng.choice([60, 100, 150, 260])]
        else:
            q = rseq(rng.randint(0, 160), "ACGTN")
        if rng.random() < 0.05:
            q = q.lower()
        return dict(ref=ref, query=q, e=e, flags=flags, wr=wr, wq=wq, mo=mo, ic=ic)

    def run_locate_case(c):
        ref_out = Aligner(c["ref"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"]).locate(c["query"])
        orc_out = O.locate(c["ref"], c["query"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        assert ref_out == orc_out, (c, ref_out, orc_out)
        return ref_out

    cases = []
    for _ in range(6000):
        c = gen_locate_case()
        c["out"] = run_locate_case(c)
        cases.append(c)
    dump("locate_fuzz.json.gz", cases)
    for _ in range(args.big_fuzz):
        run_locate_case(gen_locate_case())
    print("oracle == reference on %d further locate cases" % args.big_fuzz)

    # ------------------------------------------------------------------ multi fuzz
    def gen_multi_case():
        m = rng.randint(1, 60)
        ref = rseq(m, "ACGTN")
        if rng.random() < 0.7:
            q = mutate(ref, 0.1, "ACGTN")
            sh = rng.randint(0, m)
            q = (rseq(rng.randint(0, 5)) + q) if rng.random() < 0.5 else q[sh:] + rseq(sh)
            if rng.random() < 0.5:
                q = q[:m]
        else:
            q = rseq(rng.randint(0, 60), "ACGTN")
        if rng.random() < 0.05:
            q = ref
        flags = rng.choice([9, 9, 9, 14, 11, 8, 2, 15, 0, 5, 10, 1])
        mx = rng.choice([100, 100, 1, 2, 3])
        if flags & 4:
            mx = 200   # the reference's match array overflows for small max_matches with STOP_WITHIN_SEQ1
        return dict(ref=ref, query=q, e=rng.choice([0, 0.1, 0.2, 0.3, 0.5]), flags=flags,
                    mo=rng.choice([1, 3, 5]), mx=mx)

    def run_multi_case(c):
        ref_out = MultiAligner(c["e"], c["flags"], c["mo"]).locate(c["ref"], c["query"], c["mx"])
        assert ref_out == O.multi_locate(c["ref"], c["query"], c["e"], c["flags"], c["mo"], c["mx"]), c
        return ref_out

    cases = []
    for _ in range(2000):
        c = gen_multi_case()
        c["out"] = run_multi_case(c)
        cases.append(c)
    dump("multi_fuzz.json.gz", cases)
    for _ in range(args.big_fuzz // 3):
        run_multi_case(gen_multi_case())

    # ------------------------------------------------------------------ prefixes / suffixes
    cases = []
    for _ in range(1500):
        alpha = "ACGTNRYKMXacgtn"
        a = rseq(rng.randint(0, 40), alpha)
        b = mutate(a, 0.1, alpha) if rng.random() < 0.5 else rseq(rng.randint(0, 40), alpha)
        wr, wq = rng.random() < 0.5, rng.random() < 0.5
        p, s = compare_prefixes(a, b, wr, wq), compare_suffixes(a, b, wr, wq)
        assert p == O.compare_prefixes(a, b, wr, wq) and s == O.compare_suffixes(a, b, wr, wq)
        cases.append(dict(ref=a, query=b, wr=wr, wq=wq, prefix=p, suffix=s))
    dump("prefix_fuzz.json.gz", cases)

    # ------------------------------------------------------------------ insert fuzz
    def noise(s, p):
        return "".join((rng.choice("ACGTN") if rng.random() < p else c) for c in s)

    cfgs = [dict(), dict(max_insert_mismatch_frac=0.1, max_adapter_mismatch_frac=0.1), dict(read_wildcards=True),
            dict(adapter_wildcards=False), dict(min_insert_overlap=5, min_adapter_overlap=3),
            dict(insert_max_rmp=1e-3, adapter_max_rmp=1e-2), dict(adapter_check_cutoff=3),
            dict(base_probs=dict(match_prob=0.33, mismatch_prob=0.67))]
    cases = []
    nbig = 0
    for ci, cfg in enumerate(cfgs):
        a1, a2 = A1, A2
        if ci % 2:
            a1 = a1[:20].replace("G", "N", 1)
        ref = InsertAligner(a1, a2, **cfg)
        orc = O.InsertOracle(a1, a2, **cfg)
        for it in range(250 + args.big_fuzz // 40):
            n = rng.choice([20, 50, 100, 150, 250])
            f = rng.randint(0, int(1.6 * n))
            F = rseq(f)
            r1 = (F + a1.replace("N", "A") + rseq(n))[:rng.choice([n, n, n, n - 3])]
            r2 = (reverse_complement(F) + a2 + rseq(n))[:n]
            p = rng.choice([0, 0.01, 0.03, 0.1])
            r1, r2 = noise(r1, p), noise(r2, p)
            if rng.random() < 0.1:
                r1 = rseq(n)
            res = ref.match_insert(r1, r2)
            out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
            mine = orc.match_insert(r1, r2)
            mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                              None if mine[2] is None else list(mine[2])]
            assert mine == out, (cfg, r1, r2, mine, out)
            nbig += 1
            if it < 250:
                cases.append(dict(a1=a1, a2=a2, r1=r1, r2=r2, kw=cfg, out=out))
    dump("insert_fuzz.json.gz", cases)
    print("oracle == reference on %d match_insert cases" % nbig)

    # ------------------------------------------------------------------ Adapter.match_to
    rmp_obj = RandomMatchProbability()
    cases = []
    for it in range(3000):
        m = rng.randint(3, 40)
        seq = rseq(m, "ACGT" if rng.random() < 0.75 else "ACGTN")
        where = rng.choice([BACK, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE])
        kw = dict(max_error_rate=rng.choice([0.1, 0.12, 0.2, 0.29, 0.35]), min_overlap=rng.choice([1, 3, 5]),
                  read_wildcards=rng.random() < 0.2, adapter_wildcards=rng.random() < 0.8,
                  indels=rng.random() < 0.8, indel_cost=rng.choice([1, 1, 3]))
        use_rmp = rng.random() < 0.3
        a = mutate(seq.replace("N", "A"), rng.choice([0, 0, 0.05, 0.12]))
        cut, pos = rng.randint(0, len(a)), rng.randint(0, 100)
        w = rng.random()
        if where in (BACK, SUFFIX):
            read = rseq(pos) + (a if w < 0.5 else a[:cut]) + (rseq(rng.randint(0, 20)) if where == BACK and w < 0.4 else "")
        elif where in (FRONT, PREFIX):
            read = (a if w < 0.5 else a[cut:]) + rseq(pos)
        else:
            read = rseq(pos) + a + rseq(rng.randint(0, 30)) if w < 0.5 else a[cut:] + rseq(pos)
        if rng.random() < 0.1:
            read = rseq(rng.randint(0, 100), "ACGTN")
        if rng.random() < 0.1:
            read = read.lower()
        ad = Adapter(seq, where, match_probability=rmp_obj if use_rmp else None,
                     max_rmp=1e-6 if use_rmp else None, **kw)
        mt = ad.match_to(Sequence(name="r", sequence=read))
        cases.append(dict(seq=seq, where=where, read=read, kw=kw, use_rmp=use_rmp, out=match_fields(mt)))
    dump("match_to_fuzz.json.gz", cases)

    # ------------------------------------------------------------------ synthetic workload heads
    heads = {}

    def rows(t):
        return [bytes(x.tolist()).decode("ascii") for x in t]

    for name in ("C1", "C2"):
        w = synth.workload(name, 0, 768)
        al = Aligner(w["adapter"], w["max_error_rate"], BACK, False, False, w["min_overlap"], w["indel_cost"])
        outs = []
        for q in rows(w["reads"]):
            o = al.locate(q)
            assert o == O.locate(w["adapter"], q, w["max_error_rate"], BACK, False, False, w["min_overlap"], 1)
            outs.append(o)
        heads[name] = dict(count=768, out=outs)
    for name in ("C3", "C5"):
        cnt = 384 if name == "C3" else 256
        w = synth.workload(name, 0, cnt)
        kw = dict(read_wildcards=True) if name == "C5" else {}
        ia = InsertAligner(w["adapter1"], w["adapter2"], **kw)
        orc = O.InsertOracle(w["adapter1"], w["adapter2"], **kw)
        outs = []
        for r1, r2 in zip(rows(w["reads1"]), rows(w["reads2"])):
            res = ia.match_insert(r1, r2)
            out = None if res is None else [list(res[0]), match_fields(res[1]), match_fields(res[2])]
            mine = orc.match_insert(r1, r2)
            mine = None if mine is None else [list(mine[0]), None if mine[1] is None else list(mine[1]),
                                              None if mine[2] is None else list(mine[2])]
            assert mine == out
            outs.append(out)
        heads[name] = dict(count=cnt, kw=kw, out=outs)
    w = synth.workload("C4", 0, 512)
    linked = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                            min_overlap=w["min_overlap"], indel_cost=w["indel_cost"])
              for f, b in zip(w["fronts"], w["backs"])]
    outs = []
    for q in rows(w["reads"]):
        per = []
        for la in linked:
            lm = la.match_to(Sequence(name="r", sequence=q))
            per.append(None if lm is None else [match_fields(lm.front_match), match_fields(lm.back_match)])
        outs.append(per)
    heads["C4"] = dict(count=512, out=outs)
    dump("synth_heads.json.gz", heads)

    # ------------------------------------------------------------------ callers (cutters)
    from atropos.commands.trim.modifiers import AdapterCutter, InsertAdapterCutter, ErrorCorrectorMixin
    from atropos.adapters import AdapterParser

    def read_state(r):
        return dict(seq=r.sequence, qual=r.qualities, corrected=r.corrected, overlap=bool(r.insert_overlap),
                    match=match_fields(r.match) if r.match is not None and hasattr(r.match, "astart") else None,
                    adapter=(r.match.adapter.name if r.match is not None and hasattr(r.match, "astart") else None),
                    n_info=(len(r.match_info) if r.match_info else 0))

    cutter_cases = []
    for it in range(260):
        nad = rng.choice([1, 2, 3])
        specs = []
        for a in range(nad):
            m = rng.randint(6, 34)
            seq = rseq(m, "ACGT" if rng.random() < 0.8 else "ACGTN")
            where = rng.choice([BACK, BACK, FRONT, PREFIX, SUFFIX, ANYWHERE])
            specs.append(dict(seq=seq, where=where, name="ad%d" % a))
        kw = dict(max_error_rate=rng.choice([0.1, 0.12, 0.2]), min_overlap=rng.choice([1, 3, 5]),
                  indels=rng.random() < 0.85)
        times = rng.choice([1, 1, 2, 3])
        action = rng.choice(['trim', 'trim', 'mask', None])
        reads = []
        for _ in range(12):
            body = rseq(rng.randint(20, 90))
            sp = rng.choice(specs)
            a = mutate(sp["seq"].replace("N", "C"), rng.choice([0, 0, 0.05, 0.1]))
            w = rng.random()
            if sp["where"] in (BACK, SUFFIX):
                q = body + (a if w < 0.6 else a[:rng.randint(0, len(a))]) + (rseq(rng.randint(0, 15)) if sp["where"] == BACK and w < 0.3 else "")
            elif sp["where"] in (FRONT, PREFIX):
                q = (a if w < 0.6 else a[rng.randint(0, len(a)):]) + body
            else:
                q = body + a + rseq(rng.randint(0, 10)) if w < 0.5 else a + body
            if rng.random() < 0.3:
                sp2 = rng.choice(specs)
                q = q + sp2["seq"].replace("N", "G")[:rng.randint(3, len(sp2["seq"]))]
            if rng.random() < 0.15:
                q = rseq(rng.randint(0, 60), "ACGTN")
            qual = "".join(chr(33 + rng.randint(2, 40)) for _ in q) if rng.random() < 0.7 else None
            reads.append((q, qual))
        ads = [Adapter(sp["seq"], sp["where"], name=sp["name"], **kw) for sp in specs]
        cutter = AdapterCutter(ads, times=times, action=action)
        outs = []
        for q, qual in reads:
            outs.append(read_state(cutter(Sequence(name="r", sequence=q, qualities=qual))))
        cutter_cases.append(dict(specs=specs, kw=kw, times=times, action=action, reads=reads, out=outs,
                                 with_adapters=cutter.with_adapters))
    dump("cutter_fuzz.json.gz", cutter_cases)

    icut_cases = []
    parser_kw = dict(max_error_rate=0.2, min_overlap=1, indel_cost=3, match_probability=RandomMatchProbability(),
                     max_rmp=1e-6)
    for it in range(130):
        action = rng.choice([None, None, 'liberal', 'conservative', 'N'])
        kwc = dict(mismatch_action=action)
        if rng.random() < 0.3:
            kwc.update(max_insert_mismatch_frac=0.3, max_adapter_mismatch_frac=0.3)
        if rng.random() < 0.2:
            kwc.update(read_wildcards=True)
        if rng.random() < 0.2:
            kwc.update(symmetric=False)
        trim_action = rng.choice(['trim', 'trim', 'mask', None])
        ad1 = Adapter(A1, BACK, name="a1", **parser_kw)
        ad2 = Adapter(A2, BACK, name="a2", **parser_kw)
        cutter = InsertAdapterCutter(ad1, ad2, action=trim_action, **kwc)
        pairs, outs = [], []
        for _ in range(10):
            n = rng.choice([40, 75, 100, 150])
            f = rng.randint(5, int(1.6 * n))
            F = rseq(f)
            r1 = (F + A1 + rseq(n))[:rng.choice([n, n, n, n - 4])]
            r2 = (reverse_complement(F) + A2 + rseq(n))[:rng.choice([n, n, n - 2])]
            p = rng.choice([0, 0.01, 0.03, 0.08])
            r1, r2 = noise(r1, p), noise(r2, p)
            if rng.random() < 0.15:
                r2 = (rseq(rng.randint(0, n)) + A2)[:n].ljust(n, "A")   # no overlap: adapter-only fallback
            q1 = "".join(chr(33 + rng.randint(2, 40)) for _ in r1)
            q2 = "".join(chr(33 + rng.randint(2, 40)) for _ in r2)
            if rng.random() < 0.3:
                q1, q2 = "I" * len(r1), "I" * len(r2)             # equal qualities: the 'liberal' mean rule
            if action in (None, 'N') and rng.random() < 0.3:
                q1 = q2 = None
            try:
                a, b = cutter(Sequence(name="p", sequence=r1, qualities=q1),
                              Sequence(name="p", sequence=r2, qualities=q2))
            except (ValueError, IndexError):
                # the reference itself fails on some synthesized overlaps of unequal-length
                # reads (empty quality slice in correct_errors); such pairs are not recorded
                continue
            pairs.append((r1, q1, r2, q2))
            outs.append([read_state(a), read_state(b)])
        icut_cases.append(dict(kw=kwc, trim_action=trim_action, pairs=pairs, out=outs,
                               with_adapters=list(cutter.with_adapters), corrected_pairs=cutter.corrected_pairs,
                               corrected_bp=list(cutter.corrected_bp)))
    dump("insert_cutter_fuzz.json.gz", icut_cases)

    # known answers of the reference's own caller tests (tests/test_modifiers.py:295-324, :444-471,
    # tests/test_trim.py:17-28)
    caller_kats = {}
    r1 = 'TTGTTTTTATGGAGAGAGTTTTAAGGTTTATTTTAGTTTTAAAGGATATTGTAGGTTAGAGGGAAAGTGTATGATGAAGGTATATATTGGTAGATCGGAAGAGCACACGTCTGAACTTCAGTCAC'
    r2 = 'ACCAATATTTTACTCCATCATACACTTACCCTCTAAACTATAATAACTTTTTTATCTATACTTAACCTTTATTTTCAACTCATCACAATAAAGATCCGAAGAGAGACGTGAAGGGAAAGAACATA'
    a1 = "GATCGGAAGAGCACACGTCTGAACTCCAGTCACCAGATCATCTCGTATGCCGTCTTCTGCTTG"
    parser = AdapterParser()
    cutter = InsertAdapterCutter(parser.parse_from_spec(a1), parser.parse_from_spec(A2),
                                 max_insert_mismatch_frac=0.3, max_adapter_mismatch_frac=0.3)
    n1, n2 = cutter(Sequence('foo', r1, '#' * 125), Sequence('foo', r2, '#' * 125))
    assert len(n1) == 91 and len(n2) == 91
    caller_kats["mismatched_adapter_overlaps"] = dict(r1=r1, r2=r2, a1=a1, a2=A2, out=[read_state(n1), read_state(n2)])
    e1 = Sequence('read1', 'TTTGCAGCTTTTGTAGACAAGTGCTGTGCAGCTGATGTCAAAGAGACCTGCTTTGCTCTGGAGGGTCCAAAACTTGTAGCCTCAACCCGAGAAGCCATAGCCTAA',
                  'CCCCCFCGGGGGBFFAFC<?BEADCCF<FFFFGFFDFDFFGGGGCFGGC?DFFFEC;,===??DG==DDDFFFFG8DDD7+5;;DF*=)))10885D**58>6=0')
    e2 = Sequence('read1', 'ATAGGCTATGGCTTCTCGAGTTGAAGCTACAAGTTTTGGACCCTCCAGAGCAAAGCAGGTCTCTTTGACATCAGCTGCACAGCACTTGTCTACAAAAGCTGCAAAAGATCGGAAGAGCGTCTCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTCGACGTATCATTAAAAAAAAAAACACATCACATCAACAAGATAACACGACTTCTCCATCCACAGTACCGATGACCTCAACATTAGT',
                  'CCCCCG@FCFGGCFGGGGFEFGFGGFCFGGGFGFGGGGGGGGGGGGGGGGGGGGGGGGGGG9FGGGGGGGFGDFFGGGGGGGGGGGGGGGGG8;>@?@FEGGGGGGGGGGGGGGGGGGGGG=DDFAEFFFGF>B>EA):DFFBDFFB6CDEDDD9=99DD>55)580:A5)*)*;DD>**51:0118):)4))1***0:*)*)((***0*.(((((*)/.)1/(6((()1.)(((6).-----8<:C<73')
    caller_kats["unequal_lengths_in"] = dict(s1=e1.sequence, q1=e1.qualities, s2=e2.sequence, q2=e2.qualities)
    im, _, _ = InsertAligner(A1, A2).match_insert(e1.sequence, e2.sequence)
    ec = ErrorCorrectorMixin('N')
    ec.correct_errors(e1, e2, im, truncate_seqs=True)
    assert e1.corrected == 3 and e2.corrected == 3
    caller_kats["unequal_lengths_out"] = dict(insert=list(im), r1=read_state(e1), r2=read_state(e2))
    ad = Adapter('CCCC', BACK, 0.1)
    t = AdapterCutter([ad], times=3)(Sequence('name', 'AAAACCCCAAAA'))
    caller_kats["statistics"] = dict(out=read_state(t), lengths_back=dict(ad.lengths_back))
    dump("caller_kats.json", caller_kats)

    # ------------------------------------------------------------------ RMP values
    R = RandomMatchProbability()
    vals = []
    for size in list(range(0, 40)) + [50, 64, 100, 125, 150, 171, 200, 250, 300]:
        for k in sorted(set([0, 1, size // 3, size // 2, max(0, size - 2), max(0, size - 1), size])):
            if k <= size:
                v = R(k, size)
                assert v == O.rmp(k, size)
                vals.append([k, size, 0.25, 0.75, repr(v)])
                v = RandomMatchProbability()(k, size, 0.33, 0.67)
                assert v == O.rmp(k, size, 0.33, 0.67)
                vals.append([k, size, 0.33, 0.67, repr(v)])
    assert R(3, 5) == 0.103515625 and R.factorial(27) == 10888869450418352160768000000
    dump("rmp.json", vals)
    print("all golden fixtures written; oracle pinned")


if __name__ == "__main__":
    main()
