#!/usr/bin/env python3
"""Differential fuzz of the device trim pipelines against the reference's `atropos trim` command
(development tool: needs the reference built in --scratch, so it only runs in the build
container; the kernels are driven through the CPU emulation).  Prints every disagreement.
usage: tools/fuzz_trim_vs_reference.py [--cases N] [--seed S] [--paired]"""
import argparse
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rseq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def mutate(rng, s, p):
    out = []
    for ch in s:
        r = rng.random()
        if r < p * 0.7:
            out.append(rng.choice("ACGT"))
        elif r < p * 0.85:
            continue
        elif r < p:
            out.append(ch + rng.choice("ACGT"))
        else:
            out.append(ch)
    return "".join(out)


def make_reads(rng, n, adapters, L):
    recs = []
    for i in range(n):
        kind = rng.random()
        body = rseq(rng, rng.randint(0, L), "ACGT" if rng.random() < 0.85 else "ACGTN")
        if adapters and kind < 0.6:
            ad = mutate(rng, rng.choice(adapters), rng.choice([0, 0, 0.05, 0.15]))
            where = rng.random()
            if where < 0.5:
                seq = body + ad[:rng.randint(1, len(ad))] if rng.random() < 0.4 else body + ad + rseq(rng, rng.randint(0, 20))
            elif where < 0.8:
                seq = ad[rng.randint(0, len(ad) - 1):] + body
            else:
                seq = body[:len(body) // 2] + ad + body[len(body) // 2:]
        else:
            seq = body
        seq = seq[:L + 40]
        if rng.random() < 0.1:
            seq = "N" * rng.randint(1, 4) + seq + "N" * rng.randint(0, 4)
        if rng.random() < 0.05:
            seq = seq.lower()
        q = "".join(chr(33 + max(2, min(40, int(38 - k * rng.uniform(0, 0.5) + rng.randint(-4, 4))))) for k in range(len(seq)))
        recs.append("@r%d extra\n%s\n+\n%s\n" % (i, seq, q))
    return "".join(recs).encode()


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def make_pairs(rng, n, a1, a2, L):
    """Read pairs cut from fragments (shorter or longer than the reads) with the adapters behind them."""
    out1, out2 = [], []
    for i in range(n):
        frag = rseq(rng, rng.randint(5, 2 * L + 20))
        rc = "".join(COMP[c] for c in reversed(frag))
        l1 = L if rng.random() < 0.8 else rng.randint(max(1, L - 25), L)
        l2 = L if rng.random() < 0.8 else rng.randint(max(1, L - 25), L)
        p = rng.choice([0, 0, 0.01, 0.04])
        r1 = mutate(rng, (frag + a1 + rseq(rng, L))[:l1], p)
        r2 = mutate(rng, (rc + a2 + rseq(rng, L))[:l2], p)
        if rng.random() < 0.08 and r1:
            k = rng.randrange(len(r1))
            r1 = r1[:k] + "N" + r1[k + 1:]
        soft = rng.random()
        if soft < 0.06:                                       # soft-masked: a stretch of one read, or both reads whole
            a = rng.randrange(len(r1) + 1)
            r1 = r1[:a] + r1[a:].lower()
        elif soft < 0.1:
            r1, r2 = r1.lower(), r2.lower()
        elif soft < 0.14 and r2:
            a = rng.randrange(len(r2))
            r2 = r2[:a] + r2[a:a + 12].lower() + r2[a + 12:]
        for r, out, tag in ((r1, out1, 1), (r2, out2, 2)):
            q = "".join(chr(33 + rng.randint(2, 40)) for _ in r)
            out.append("@p%d/%d\n%s\n+\n%s\n" % (i, tag, r, q))
    return "".join(out1).encode(), "".join(out2).encode()


def insert_args(rng):
    a1, a2 = rseq(rng, rng.randint(12, 40)), rseq(rng, rng.randint(12, 40))
    args = ["--aligner", "insert", "-a", a1, "-A", a2]
    if rng.random() < 0.4:
        args += ["-e", str(rng.choice([0.1, 0.15, 0.25]))]
    if rng.random() < 0.3:
        args += ["--insert-match-error-rate", str(rng.choice([0.1, 0.3]))]
    if rng.random() < 0.2:
        args += ["--insert-match-adapter-error-rate", str(rng.choice([0.1, 0.3]))]
    if rng.random() < 0.2:
        args += ["--insert-max-rmp", str(rng.choice([1e-4, 1e-9]))]
    if rng.random() < 0.5:
        args += ["--correct-mismatches", rng.choice(["liberal", "conservative", "N"])]
    if rng.random() < 0.2:
        args += ["--match-read-wildcards"]
    if rng.random() < 0.15:
        args += ["-N"]
    if rng.random() < 0.25:
        args += ["-q", rng.choice(["10", "20", "15,15"])]
    if rng.random() < 0.2:
        args += ["-u", str(rng.choice([2, -3]))]
    if rng.random() < 0.2:
        args += ["-U", str(rng.choice([3, -2]))]
    if rng.random() < 0.3:
        args += ["--trim-n"]
    if rng.random() < 0.5:
        args += ["-m", str(rng.choice([1, 15, 40]))]
    if rng.random() < 0.2:
        args += ["--max-n", str(rng.choice([1, 0.1]))]
    r = rng.random()
    if r < 0.1:
        args += ["--discard-trimmed"]
    elif r < 0.2:
        args += ["--discard-untrimmed"]
    r = rng.random()
    if r < 0.12:
        args += ["--mask-adapter"]
    elif r < 0.2:
        args += ["--no-trim"]
    if rng.random() < 0.3:
        args += ["--pair-filter", rng.choice(["any", "both"])]
    return a1, a2, args


def random_args(rng, paired):
    adapters, args = [], []
    nad = rng.choice([0, 1, 1, 2, 3])
    linked = rng.random() < 0.1 and nad >= 1
    for k in range(nad):
        ad = rseq(rng, rng.randint(6, 30))
        adapters.append(ad)
        if linked:
            ad2 = rseq(rng, rng.randint(8, 20))
            adapters.append(ad2)
            args += ["-a", "%s...%s" % (ad, ad2)]
            break
        flag = rng.choice(["-a", "-a", "-g", "-b"])
        spec = ad
        if flag == "-g" and rng.random() < 0.3:
            spec = "^" + ad
        elif flag == "-a" and rng.random() < 0.2:
            spec = ad + "$"
        if rng.random() < 0.2:
            spec = "name%d=%s" % (k, spec)
        args += [flag, spec]
    if paired:
        ad = rseq(rng, rng.randint(8, 30))
        adapters.append(ad)
        args += ["-A", ad]
    if rng.random() < 0.5:
        args += ["-e", str(rng.choice([0.05, 0.1, 0.15, 0.2]))]
    if rng.random() < 0.5:
        args += ["-O", str(rng.choice([1, 3, 5, 8]))]
    if rng.random() < 0.3 and not linked:
        args += ["-n", str(rng.choice([2, 3]))]
    if rng.random() < 0.2:
        args += ["-N"]
    if rng.random() < 0.2:
        args += ["--match-read-wildcards"]
    if rng.random() < 0.2:
        args += ["--no-indels"]                    # with anchored adapters: the compare_prefixes / compare_suffixes branch
    if rng.random() < 0.5:
        args += ["-q", rng.choice(["10", "20", "15,20", "25,5", "0,30"])]
    if rng.random() < 0.3:
        args += ["-u", str(rng.choice([1, 3, 7, -2, -5]))]
        if rng.random() < 0.3:
            args += ["-u", str(-4 if int(args[-1]) > 0 else 4)]
    if paired and rng.random() < 0.3:
        args += ["-U", str(rng.choice([2, -3]))]
    if rng.random() < 0.15:
        args += ["--nextseq-trim", str(rng.choice([10, 20]))]
    if rng.random() < 0.4:
        args += ["--trim-n"]
    if rng.random() < 0.5:
        args += ["-m", str(rng.choice([1, 10, 25, 40]))]
    if rng.random() < 0.2:
        args += ["-M", str(rng.choice([30, 60, 100]))]
    if rng.random() < 0.25:
        args += ["--max-n", str(rng.choice([0, 1, 3, 0.05, 0.2]))]
    r = rng.random()
    if r < 0.1:
        args += ["--discard-trimmed"]
    elif r < 0.2:
        args += ["--discard-untrimmed"]
    r = rng.random()
    if r < 0.12:
        args += ["--mask-adapter"]
    elif r < 0.2:
        args += ["--no-trim"]
    if rng.random() < 0.1:
        order = list("CGQAW")
        rng.shuffle(order)
        args += ["--op-order", "".join(order)]
    if paired and rng.random() < 0.3:
        args += ["--pair-filter", rng.choice(["any", "both"])]
    if not linked and rng.random() < 0.25:
        plain = "--mask-adapter" not in args and "--no-trim" not in args
        kinds = ["rrbs", "truseq", "%d,%d,0,%d" % (rng.randint(0, 6), rng.randint(0, 6), rng.randint(0, 1))]
        if plain:
            kinds += ["non-directional", "non-directional-rrbs", "%d,%d,1,%d" % (rng.randint(0, 8), rng.randint(0, 8), rng.randint(0, 1))]
        if paired:
            kinds += ["swift", "2,0,0,0;0,4,0,1"]
        args += ["--bisulfite", rng.choice(kinds)]
    if "--mask-adapter" not in args and "--no-trim" not in args and rng.random() < 0.2:
        args += ["--cut-min", str(rng.choice([3, 10, 30, -4, -15]))]
        if rng.random() < 0.3:
            args += ["--cut-min", str(-6 if int(args[-1]) > 0 else 6)]
        if paired and rng.random() < 0.5:
            args += ["--cut-min2", str(rng.choice([5, -7]))]
    return adapters, args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--insert", action="store_true", help="paired-end with the insert aligner (+ error correction)")
    ap.add_argument("--long", action="store_true", help="reads of 150 .. 300 bases (MiSeq lengths)")
    ap.add_argument("--merge", action="store_true", help="paired-end modes: add --merge-overlapping options (third output file)")
    ap.add_argument("--aux", action="store_true", help="adapter aligner: add --info-file / --rest-file / --wildcard-file, the "
                    "read-name modifiers, -z and the extra output files, and compare every file")
    ap.add_argument("--dump", type=int, default=-1, help="write the inputs and the command line of this case to /tmp/fuzz_case.*")
    ap.add_argument("--scratch", default="/tmp/oracle_ref")
    o = ap.parse_args()
    sys.path.insert(0, o.scratch)
    from atropos.commands import get_command
    from emu.backend import EmuBackend
    from atropos_amd import _lib
    _lib.set_backend(EmuBackend(), _test_double=True)
    from atropos_amd.trim import pipeline_from_args
    rng = random.Random(o.seed)
    bad = skipped = 0
    with tempfile.TemporaryDirectory() as tmp:
        o.paired = o.paired or o.insert
        for case in range(o.cases):
            L = rng.choice([30, 60, 100, 150] if not o.long else [150, 250, 300])
            n = rng.choice([1, 20, 70, 130])
            if o.insert:
                a1, a2, args = insert_args(rng)
                data1, data2 = make_pairs(rng, n, a1, a2, L)
            else:
                adapters, args = random_args(rng, o.paired)
                data1 = make_reads(rng, n, adapters, L)
                data2 = make_reads(rng, n, adapters, L) if o.paired else None
            paths = [os.path.join(tmp, x) for x in ("i1.fq", "i2.fq", "o1.fq", "o2.fq")]
            open(paths[0], "wb").write(data1)
            merging = o.merge and o.paired and rng.random() < 0.8
            if merging:
                args = list(args) + ["-R", "--merge-min-overlap", str(rng.choice([0.3, 0.5, 0.9, 8, 20]))]
                if rng.random() < 0.5:
                    args += ["--merge-error-rate", str(rng.choice([0.05, 0.1, 0.2]))]
                if "--correct-mismatches" not in args and rng.random() < 0.5:
                    args += ["--correct-mismatches", rng.choice(["liberal", "conservative", "N"])]
                paths.append(os.path.join(tmp, "merged.fq"))
            aux = {}
            if o.aux and not o.insert and not merging:
                # adapters get names (the reference numbers unnamed ones with a process-wide counter)
                args = list(args)
                for i, a in enumerate(args):
                    if i and args[i - 1] in ("-a", "-g", "-b", "-A") and "=" not in a and "..." not in a:
                        args[i] = "ad%d=%s" % (i, a)
                if rng.random() < 0.5:
                    data1 = data1.replace(b" extra\n", b" length=77 x\n")
                    open(paths[0], "wb").write(data1)
                for kind, flag in (("info", "--info-file"), ("rest", "--rest-file"), ("wildcard", "--wildcard-file")):
                    if rng.random() < 0.6 and not any("..." in a for a in args):
                        aux[kind] = os.path.join(tmp, kind + ".txt")
                        args += [flag, aux[kind]]
                if rng.random() < 0.4:
                    args += ["--length-tag", "length="]
                if rng.random() < 0.3:
                    args += ["--strip-suffix", rng.choice([" x", "extra", "7 x"])]
                if rng.random() < 0.4 and not any("..." in a for a in args):
                    args += ["-x", rng.choice(["p_", "{name}:"])]
                if rng.random() < 0.3 and not any("..." in a for a in args):
                    args += ["-y", rng.choice(["_s", "/{name}"])]
                if rng.random() < 0.2:
                    args += ["-z"]
                for kind in aux:
                    if os.path.exists(aux[kind]):
                        os.remove(aux[kind])
            params = list(args)
            if o.paired:
                open(paths[1], "wb").write(data2)
                params += ["-pe1", paths[0], "-pe2", paths[1], "-o", paths[2], "-p", paths[3]]
                if merging:
                    params += ["--merged-output", paths[4]]
            else:
                params += ["-se", paths[0], "-o", paths[2]]
            params += ["--quiet", "--no-default-adapters", "--no-cache-adapters"]
            if case == o.dump:
                open("/tmp/fuzz_case.1.fq", "wb").write(data1)
                if data2 is not None:
                    open("/tmp/fuzz_case.2.fq", "wb").write(data2)
                open("/tmp/fuzz_case.args", "w").write(" ".join(args))
            for p in paths[2:]:
                if os.path.exists(p):
                    os.remove(p)
            try:
                rc, _ = get_command("trim").execute(params)
            except SystemExit:
                rc = 2
            if rc != 0:
                # the reference rejects / fails on this command line: so must the pipeline
                try:
                    pipe = pipeline_from_args(args)
                    pipe.trim_bytes(data1, data2) if o.paired else pipe.trim_bytes(data1)
                except (Exception, SystemExit):   # noqa: BLE001
                    skipped += 1
                else:
                    bad += 1
                    print("CASE %d: the reference fails, the pipeline does not\n  args: %s" % (case, " ".join(args)))
                continue
            # (the reference creates an output file with the first record that goes there: no merged read, no file)
            want = [open(p, "rb").read() if os.path.exists(p) else b"" for p in (paths[2:] if o.paired else paths[2:3])]
            want += [open(p, "rb").read() if os.path.exists(p) else b"" for p in aux.values()]
            try:
                pipe = pipeline_from_args(args)
                if aux:
                    from atropos_amd.fastq import FastqBatch
                    if o.paired:
                        res = pipe.run(FastqBatch.from_bytes(data1, final=True)[0], FastqBatch.from_bytes(data2, final=True)[0])
                        got = tuple(res.text())
                    else:
                        res = pipe.run(FastqBatch.from_bytes(data1, final=True)[0])
                        got = (res.text(),)
                    texts = res.aux_text(tuple(aux))
                    got += tuple(texts[k] for k in aux)
                elif merging:
                    from atropos_amd.fastq import FastqBatch
                    res = pipe.run(FastqBatch.from_bytes(data1, final=True)[0], FastqBatch.from_bytes(data2, final=True)[0])
                    got = res.text() + (res.merged_text(),)
                else:
                    got = pipe.trim_bytes(data1, data2) if o.paired else (pipe.trim_bytes(data1),)
            except NotImplementedError:
                skipped += 1
                continue
            except Exception as err:              # noqa: BLE001
                bad += 1
                print("CASE %d RAISED %r\n  args: %s" % (case, err, " ".join(args)))
                continue
            if list(got) != want:
                bad += 1
                print("CASE %d DIFFERS\n  args: %s\n  reads: %d x <=%d" % (case, " ".join(args), n, L))
                for g, w in zip(got, want):
                    if g != w:
                        gl, wl = g.split(b"\n"), w.split(b"\n")
                        for i in range(min(len(gl), len(wl))):
                            if gl[i] != wl[i]:
                                print("   line %d: got %r\n            want %r" % (i, gl[i][:120], wl[i][:120]))
                                break
                        else:
                            print("   lengths differ: got %d want %d lines" % (len(gl), len(wl)))
    print("cases %d, skipped %d, disagreements %d" % (o.cases, skipped, bad))


if __name__ == "__main__":
    main()
