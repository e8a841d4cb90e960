#!/usr/bin/env python3
"""Generates tests/golden/trim_cases.json.gz: FASTQ inputs and the exact output text of the
REFERENCE's `atropos trim` command for each of a list of command lines.  Run in this container
only (the reference is imported from a scratch build, see make_golden.py --scratch); the
committed file holds data only: input text, argument strings, expected output text / error.

usage: python tests/golden/make_trim_golden.py [--scratch /tmp/oracle_ref]
"""
import argparse
import base64
import gzip
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def synth_fastq(nreads, seed, dos=False, repeat_name=False):
    """C2-like reads (adapter-containing 100 bp reads) with qualities that decay towards the
    3' end, some low-quality 5' starts, N ends and lower-case stretches."""
    sys.path.insert(0, ROOT)
    from atropos_amd import synth
    w = synth.workload("C1", 0, nreads, device="cpu")
    reads = w["reads"].numpy()
    rng = np.random.RandomState(seed)
    eol = "\r\n" if dos else "\n"
    out = []
    for i in range(nreads):
        seq = bytes(reads[i]).decode("ascii")
        n = len(seq)
        if rng.rand() < 0.1:
            n = int(rng.randint(0, n + 1))                      # ragged lengths, including empty reads
        seq = seq[:n]
        q = 38 - (np.arange(n) * rng.uniform(0.0, 0.45)).astype(int) + rng.randint(-3, 4, size=n)
        if rng.rand() < 0.15:
            q[:rng.randint(1, 8)] = rng.randint(2, 12)
        if rng.rand() < 0.2:
            k = rng.randint(1, 6)
            seq = "N" * min(k, n) + seq[k:]
        if rng.rand() < 0.2:
            k = rng.randint(1, 6)
            seq = seq[:max(0, n - k)] + "N" * min(k, n)
        if rng.rand() < 0.05:
            seq = seq.lower()
        if rng.rand() < 0.05 and n > 20:
            seq = seq[:n - 9] + "GGGGGGGGG"                     # NextSeq dark cycles
        q = np.clip(q, 2, 40)
        qual = "".join(chr(int(v) + 33) for v in q)
        name = "read%d/1 comment=%d" % (i, rng.randint(0, 1000))
        plus = "+" + (name if (repeat_name or rng.rand() < 0.05) else "")
        out.append("@%s%s%s%s%s%s%s%s" % (name, eol, seq, eol, plus, eol, qual, eol))
    return "".join(out)


TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ANCHOR5, ANCHOR3 = "ACGTACGTTGCA", "GGCATTCAGGAC"


def synth_anchor_fastq(nreads, seed):
    """Reads that start with (a damaged copy of) ANCHOR5 and end with (a damaged copy of) ANCHOR3:
    substitutions, the odd insertion/deletion, N, truncations -- the input of the anchored
    adapter cases, without indels (compare_prefixes / compare_suffixes) and with them."""
    rng = np.random.RandomState(seed)

    def damage(s):
        out = []
        for c in s:
            r = rng.rand()
            if r < 0.08:
                out.append("ACGTN"[rng.randint(0, 5)])
            elif r < 0.10:
                continue
            elif r < 0.12:
                out.append(c + "ACGT"[rng.randint(0, 4)])
            else:
                out.append(c)
        return "".join(out)

    out = []
    for i in range(nreads):
        head = damage(ANCHOR5) if rng.rand() < 0.8 else ""
        tail = damage(ANCHOR3) if rng.rand() < 0.8 else ""
        if rng.rand() < 0.15:
            head = head[rng.randint(0, 4):]
        if rng.rand() < 0.15:
            tail = tail[:max(0, len(tail) - rng.randint(0, 4))]
        body = "".join("ACGT"[v] for v in rng.randint(0, 4, size=rng.randint(0, 60)))
        seq = head + body + tail
        if rng.rand() < 0.05:
            seq = seq.lower()
        if rng.rand() < 0.03:
            seq = seq[:rng.randint(0, 8)]
        qual = "".join(chr(int(v) + 33) for v in rng.randint(20, 41, size=len(seq)))
        out.append("@anchor%d\n%s\n+\n%s\n" % (i, seq, qual))
    return "".join(out)

def synth_longmix_fastq(nreads, seed):
    """Short C1-like reads with a few records of 800 .. 3000 bases among them (one over-long record must not abort
    the chunk): the adapter whole or damaged somewhere in the long read, cut by its end, or absent."""
    import random
    rng = random.Random(seed)
    base = synth_fastq(nreads, seed).split("\n")
    out = []
    for i in range(nreads):
        out += base[4 * i:4 * i + 4]
        if i % 9 == 4:
            n = rng.choice([737, 768, 800, rng.randint(800, 3000)])
            seq = [rng.choice("ACGT") for _ in range(n)]
            w = rng.random()
            ad = list(TRUSEQ)
            for _ in range(rng.choice([0, 0, 1, 2, 3])):
                ad[rng.randrange(len(ad))] = rng.choice("ACGT")
            if w < 0.5:
                pos = rng.randint(0, n - 1)
                seq[pos:pos + len(ad)] = ad
            elif w < 0.75:
                cut = rng.randint(3, len(ad))
                seq[n - cut:] = ad[:cut]
            seq = "".join(seq)[:n]
            qual = "".join(chr(33 + max(2, 38 - j // 90 + rng.randint(-2, 2))) for j in range(len(seq)))
            out += ["@long%d" % i, seq, "+", qual]
    return "\n".join(out) + "\n"


LONGAD = ("GATCGGAAGAGCACACGTCTGAACTCCAGTCACTTAGGCATCTCGTATGCCGTCTTCTGCTTGAAAAAGGCCTTAACCGGTTACGTACGATTAGCCATGCATGCCAAT"
          "TGGCCAATCGTAGCTAGCTAACGGATCCGATTACAGGCATT")


def synth_longad_fastq(nreads, seed):
    """Reads of 50 .. 260 bases that run into a 150-base adapter at a random place (whole, cut by the read end,
    absent), with substitutions."""
    import random
    rng = random.Random(seed)
    out = []
    for i in range(nreads):
        n = rng.randint(50, 260)
        frag = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, n + 40)))
        seq = list((frag + LONGAD + "".join(rng.choice("ACGT") for _ in range(300)))[:n])
        for j in range(len(seq)):
            if rng.random() < 0.01:
                seq[j] = rng.choice("ACGTN")
        seq = "".join(seq)
        out.append("@la%d\n%s\n+\n%s\n" % (i, seq, "".join(chr(33 + rng.randint(20, 40)) for _ in seq)))
    return "".join(out)


# (input name, argument string) -- the reference's own CLI tests (tests/test_atropos.py) that
# take FASTQ input and stay inside the device pipeline's envelope, plus parameter sweeps on
# synthetic reads
CASES = [
    ("small.fastq", "-b TTAGACATATCTCCGTCG"),
    ("small.fastq", "-a TTAGACATATCTCCGTCG"),
    ("empty.fastq", "-a TTAGACATATCTCCGTCG"),
    ("dos.fastq", "-e 0.12 -b TTAGACATATCTCCGTCG"),
    ("small.fastq", "-b ttagacatatctccgtcg"),
    ("small.fastq", "-b TTAGACATATCTCCGTCG --discard"),
    ("small.fastq", "-b CAAGAT --discard-untrimmed"),
    ("plus.fastq", "-e 0.12 -b TTAGACATATCTCCGTCG"),
    ("lowqual.fastq", "-q 10 -a XXXXXX"),
    ("illumina64.fastq", "-q 10 --quality-base 64 -a XXXXXX"),
    ("illumina64.fastq", "-q 10 --quality-base 64"),
    ("anywhere_repeat.fastq", "-b CAAG -n 3 --mask-adapter"),
    ("illumina.fastq", "-a VCCGAMCYUCKHRKDCUBBCNUWNSGHCGU"),
    ("illumina.fastq", "-a GCCGAACTTCTTAGACTGCCTTAAGGACGT"),
    ("illumina.fastq", "-a GCCGAACUUCUUAGACUGCCUUAAGGACGU"),
    ("illumina.fastq", "--times 2 -a adapt=GCCGAACTTCTTA -a TTAGACTGCC"),
    ("small.fastq", "--no-trim --discard-untrimmed -a CCCTAGTTAAAC"),
    ("small.fastq", "-u 5"),
    ("small.fastq", "-u -5"),
    ("small.fastq", "-u -5 -u 5"),
    ("small.fastq", "-q 10 --trim-n -m 5 -a TTAGACATATCTCCGTCG"),
    ("nextseq.fastq", "--nextseq-trim 22"),
    ("illumina5.fastq", "-a CCGCCTTGGCCGT -m 10 -q 5,20"),
    ("synth.fastq", "-a " + TRUSEQ),
    ("synth.fastq", "-a " + TRUSEQ + " -q 20 -m 20"),
    ("synth.fastq", "-a " + TRUSEQ + " -q 15,25 --trim-n -m 30 -M 90"),
    ("synth.fastq", "-b " + TRUSEQ + " -e 0.2 -O 5 -n 2"),
    ("synth.fastq", "-g " + TRUSEQ[:20] + " -a " + TRUSEQ + " --trim-n"),
    ("synth.fastq", "-a " + TRUSEQ + " --mask-adapter --trim-n"),
    ("synth.fastq", "-a " + TRUSEQ + " --mask-adapter --max-n 0.3"),
    ("synth.fastq", "-a " + TRUSEQ + " --no-trim --discard"),
    ("synth.fastq", "-a " + TRUSEQ + " --discard-untrimmed -u 3 -u -2"),
    ("synth.fastq", "-a " + TRUSEQ + " --match-read-wildcards -N"),
    ("synth.fastq", "-a " + TRUSEQ + " -N --no-indels"),
    ("synth.fastq", "--nextseq-trim 20 -a " + TRUSEQ + " -m 1"),
    ("synth.fastq", "--max-n 2 --trim-n"),
    ("synth.fastq", "--max-n 0.05"),
    ("synth.fastq", "-a " + TRUSEQ + " --op-order AQCGW -q 20 -u 4"),
    ("synth.fastq", "-a ^ACGTACGT..." + TRUSEQ + " -e 0.15"),
    ("synth.fastq", "-a " + TRUSEQ + "$ -q 12"),
    ("synth.fastq", "-g ^" + "ACGTAC" + " -O 4 -e 0.2"),
    # anchored adapters, with and without indels (Adapter.match_to's compare_prefixes / compare_suffixes branch)
    ("anchor.fastq", "-g ^" + ANCHOR5 + " -e 0.2"),
    ("anchor.fastq", "-g ^" + ANCHOR5 + " -e 0.2 --no-indels"),
    ("anchor.fastq", "-a " + ANCHOR3 + "$ -e 0.2 --no-indels"),
    ("anchor.fastq", "-a " + ANCHOR3 + "$ -e 0.1 --no-indels -O 5 --discard-untrimmed"),
    ("anchor.fastq", "-a ^" + ANCHOR5 + "..." + ANCHOR3 + " -e 0.2 --no-indels"),
    ("anchor.fastq", "-g ^" + ANCHOR5 + " -a " + ANCHOR3 + "$ -e 0.25 --no-indels --match-read-wildcards -n 2"),
    ("anchor.fastq", "-g ^ACGTNCGTTGCA -e 0.2 --no-indels -N"),
    ("anchor.fastq", "-g ^ACGTNCGTTGCA -e 0.2 --no-indels --mask-adapter"),
    ("synth.fastq", "-a " + TRUSEQ + "$ --no-indels -e 0.15"),
    ("synth_dos.fastq", "-a " + TRUSEQ + " -q 20 -m 20"),
    ("synth_name2.fastq", "-a " + TRUSEQ + " --trim-n"),
    ("nofinalnewline.fastq", "-a TTAGACATATCTCCGTCG"),
    # --info-file / --rest-file / --wildcard-file (writers.py:193-222): "{info}" etc. stand for a path; the files'
    # text is recorded next to the main output
    ("small.fastq", "-a ad=TTAGACATATCTCCGTCG --info-file {info} --rest-file {rest} --wildcard-file {wildcard}"),
    ("small.fastq", "-g ad=TTAGACATATCTCCGTCG --info-file {info} --rest-file {rest}"),
    ("anywhere_repeat.fastq", "-b rep=CAAG -n 3 --mask-adapter --info-file {info} --rest-file {rest}"),
    ("illumina.fastq", "-a wild=GCCGAACTTCTTAGACTNCCTTAAGGACNT --info-file {info} --wildcard-file {wildcard}"),
    ("illumina.fastq", "--times 2 -a adapt=GCCGAACTTCTTA -a second=TTAGACTGCC --info-file {info} --rest-file {rest}"),
    ("synth.fastq", "-a tru=" + TRUSEQ + " -q 15,25 --trim-n -m 30 --info-file {info} --rest-file {rest} --wildcard-file {wildcard}"),
    ("synth.fastq", "-b tru=" + TRUSEQ + " -g head=" + TRUSEQ[:20] + " -e 0.2 -O 5 -n 2 --discard-trimmed --info-file {info} --rest-file {rest}"),
    ("synth.fastq", "-a tru=" + TRUSEQ + " --no-trim --info-file {info} --wildcard-file {wildcard} --match-read-wildcards"),
    # an adapter of 150 bases (beyond the aligner handles' 128 rows: the per-pair aligner with one reference for all)
    ("longad.fastq", "-a long=" + LONGAD + " -e 0.1"),
    ("longad.fastq", "-b long=" + LONGAD + " -e 0.15 -O 10 -n 2 --info-file {info}"),
    # the filtered reads into files of their own (trim/__init__.py:580-630)
    ("synth.fastq", "-a tru=" + TRUSEQ + " -q 20 -m 40 -M 95 --too-short-output {too_short} --too-long-output {too_long} --untrimmed-output {untrimmed}"),
    ("synth.fastq", "-a tru=" + TRUSEQ + " -m 30 --max-n 1 --mask-adapter --too-short-output {too_short} --untrimmed-output {untrimmed}"),
    # --bisulfite: RRBS / non-directional / custom MinCutter parameters (modifiers.py:786-860)
    ("synth.fastq", "-a tru=" + TRUSEQ + " --bisulfite rrbs -q 20 -m 10"),
    ("bisulf.fastq", "-a tru=" + TRUSEQ + " --bisulfite non-directional-rrbs --trim-n"),
    ("bisulf.fastq", "-a tru=" + TRUSEQ + " -g head=" + TRUSEQ[:20] + " -n 2 --bisulfite 4,6,1,0 -u 2"),
    ("bisulf.fastq", "-b tru=" + TRUSEQ + " --bisulfite 3,5,0,1 -q 15 --op-order CAGQW"),
    ("synth.fastq", "-a tru=" + TRUSEQ + " --bisulfite non-directional --mask-adapter" if False else "-a tru=" + TRUSEQ + " --bisulfite truseq -m 20"),
    # --cut-min: at least that many bases gone from an end, whatever removed them (MinCutter, modifiers.py:587-650)
    ("synth.fastq", "-a tru=" + TRUSEQ + " -q 15,20 --trim-n --cut-min 6 --cut-min -12 -m 10"),
    ("synth.fastq", "-g head=" + TRUSEQ[:20] + " -a tru=" + TRUSEQ + " -u 3 --cut-min 25 -n 2"),
    # read-name modifiers and the quality cap (modifiers.py:652-720)
    ("small.fastq", "-a ad=TTAGACATATCTCCGTCG -x pre_{name}_ -y _suf --strip-suffix /1 --strip-suffix _573"),
    ("synth.fastq", "-a tru=" + TRUSEQ + " -q 15 --length-tag comment= -y :{name} --strip-suffix 7"),
    ("lengthtag.fastq", "-a tru=" + TRUSEQ + " --trim-n --length-tag length= -x {name}: --info-file {info}"),
    ("lowqual64.fastq", "-z --quality-base 64 -q 3"),
    ("synth.fastq", "-b tru=" + TRUSEQ + " -n 2 --mask-adapter -z -y /{name} --length-tag comment="),
    # records beyond the batch pipelines' 736 bases among ordinary ones (the reference has no length limit)
    ("longmix.fastq", "-a " + TRUSEQ),
    ("longmix.fastq", "-b " + TRUSEQ + " -n 2 -e 0.12"),
    ("longmix.fastq", "-a " + TRUSEQ + " -q 20 -m 20 --trim-n"),
    ("longmix.fastq", "-g " + TRUSEQ[:20] + " -a " + TRUSEQ + " --mask-adapter"),
    # malformed inputs: the reference raises FormatError
    ("bad_at.fastq", "-a ACGT"),
    ("bad_plus.fastq", "-a ACGT"),
    ("bad_name2.fastq", "-a ACGT"),
    ("bad_length.fastq", "-a ACGT"),
    ("bad_truncated.fastq", "-a ACGT"),
]


PE1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCACACAGTGATCTCGTATGCCGTCTTCTGCTTG"
PE2 = "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTCGCCGTATCATT"

# paired-end: (input 1, input 2, argument string); outputs are the -o and -p files
PAIRED_CASES = [
    ("paired.1.fastq", "paired.2.fastq", "--aligner insert -a TTAGACATATCTCCGTCG -A CAGTGGAGTA"),
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -A CAGTGGAGTA -m 14"),
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -A CAGTGGAGTA -q 10 --pair-filter both -m 20 --trim-n"),
    ("paired.1.fastq", "paired.2.fastq", "--aligner insert -a TTAGACATATCTCCGTCG -A CAGTGGAGTA --mask-adapter -U 2"),
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -A CAGTGGAGTA --discard-untrimmed"),
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -A CAGTGGAGTA --discard-trimmed --pair-filter both"),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s -q 20 -m 30" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --mask-adapter --trim-n --max-n 0.3" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --no-trim --discard-untrimmed" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --insert-match-error-rate 0.1 -e 0.15 -u 3 -U -4" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --match-read-wildcards -q 10,15 --pair-filter both -m 60" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --correct-mismatches liberal" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --correct-mismatches conservative -q 15 -m 25" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --correct-mismatches N --match-read-wildcards --trim-n" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --correct-mismatches liberal --mask-adapter" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -q 20 -m 40 -M 140 --trim-n" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -n 2 --discard-trimmed" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s --adapter-max-rmp 0.001 --nextseq-trim 20" % (PE1, PE2)),
    # MergeOverlapping (-R): a third output, the --merged-output file
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -A CAGTGGAGTA -R --merge-min-overlap 0.5"),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s -R" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s -R --merge-min-overlap 0.5 --correct-mismatches liberal" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s -R --merge-min-overlap 30 --merge-error-rate 0.05 -q 20 -m 30" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -R --merge-min-overlap 20 --merge-error-rate 0.15 --correct-mismatches conservative" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -R --merge-min-overlap 0.3 --correct-mismatches N -q 15 -m 20 --pair-filter both" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -R --merge-min-overlap 0.25 --correct-mismatches liberal -e 0.2" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-R --merge-min-overlap 12 -u 2 -U 3 --trim-n"),
    # legacy mode (-p without -A / -G / -B / -U, -q, --trim-n ...: only read 1 is modified and filtered, cli.py:630-648)
    ("paired.1.fastq", "paired.2.fastq", "-a TTAGACATAT -m 14"),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -e 0.12 -n 2 -m 40 -M 140 --discard-untrimmed -u 2" % PE1),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-b %s --mask-adapter --max-n 3" % PE1[:25]),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -q 20 --cut-min 5 --cut-min2 -8 -m 20" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "--aligner insert -a %s -A %s --cut-min -10 --cut-min2 4 --cut-min2 -4" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s --bisulfite swift -m 20" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s --bisulfite 0,3,0,1;5,0,1,0 -q 15" % (PE1, PE2)),
    # the filtered pairs into files of their own
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s -q 20 -m 60 --too-short-output {too_short} --too-short-paired-output {too_short2} --untrimmed-output {untrimmed} --untrimmed-paired-output {untrimmed2}" % (PE1, PE2)),
    # masked adapters and merging: MergeOverlapping sees the reads with their N's
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s --mask-adapter -R --merge-min-overlap 0.3" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a %s -A %s --mask-adapter -R --merge-min-overlap 20 --correct-mismatches liberal -q 15 --trim-n" % (PE1, PE2)),
    # info / rest / wildcard files and read-name modifiers with paired-end input (one line group per read, read 1 first)
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a a1=%s -A a2=%s -q 15 --trim-n --info-file {info} --rest-file {rest}" % (PE1, PE2)),
    ("synth_pe.1.fastq", "synth_pe.2.fastq", "-a a1=%s -A a2=%s -n 2 -m 20 -x {name}_ --strip-suffix /x -z --wildcard-file {wildcard} --info-file {info}" % (PE1[:30] + "N" + PE1[31:], PE2)),
    # reads of 420 bases: MergeOverlapping's aligner beyond the per-pair pipeline's 320 (the reference has no limit)
    ("long_pe.1.fastq", "long_pe.2.fastq", "-R --merge-min-overlap 0.3 -u 1 -U 2 --trim-n"),
    ("long_pe.1.fastq", "long_pe.2.fastq", "-a %s -A %s -R --merge-min-overlap 40 --correct-mismatches liberal -q 15" % (PE1, PE2)),
    # soft-masked reads: the merge aligner compares characters, a != A
    ("soft_pe.1.fastq", "soft_pe.2.fastq", "-a %s -A %s -R --merge-min-overlap 0.4 --correct-mismatches liberal" % (PE1, PE2)),
    ("soft_pe.1.fastq", "soft_pe.2.fastq", "-R --merge-min-overlap 15 --merge-error-rate 0.1 -u 1 -U 1"),
    # ... and the insert aligner on them: characters in the insert compare, case folded in the adapter compares
    ("soft_pe.1.fastq", "soft_pe.2.fastq", "--aligner insert -a %s -A %s" % (PE1, PE2)),
    ("soft_pe.1.fastq", "soft_pe.2.fastq", "--aligner insert -a %s -A %s --correct-mismatches liberal -R --merge-min-overlap 0.5" % (PE1, PE2)),
    ("soft_pe.1.fastq", "soft_pe.2.fastq", "--aligner insert -a %s -A %s -N --correct-mismatches N -m 20" % (PE1, PE2)),
]


def synth_pairs(npairs, seed):
    """C3-like pairs (2 x 150 bp, fragment shorter or longer than the reads) with qualities."""
    sys.path.insert(0, ROOT)
    from atropos_amd import synth
    w = synth.workload("C3", 0, npairs, device="cpu")
    rng = np.random.RandomState(seed)
    texts = []
    for reads in (w["reads1"].numpy(), w["reads2"].numpy()):
        out = []
        for i in range(npairs):
            seq = bytes(reads[i]).decode("ascii")
            n = len(seq)
            if rng.rand() < 0.08:
                n = int(rng.randint(0, n + 1))
                seq = seq[:n]
            q = np.clip(38 - (np.arange(n) * rng.uniform(0.0, 0.3)).astype(int) + rng.randint(-3, 4, size=n), 2, 40)
            if rng.rand() < 0.1 and n > 8:
                seq = seq[:n - 4] + "NNNN"
            out.append("@pair%d/x\n%s\n+\n%s\n" % (i, seq, "".join(chr(int(v) + 33) for v in q)))
        texts.append("".join(out).encode())
    return texts


def synth_long_pairs(npairs, seed, n=420):
    """Pairs of 2 x 420 bp (beyond the per-pair aligner's 320): fragments shorter and longer than the reads."""
    sys.path.insert(0, ROOT)
    from atropos_amd import synth
    r1, r2 = synth.paired_end(0, npairs, n, synth.PE_ADAPTER1, synth.PE_ADAPTER2, synth.SEEDS["C3"] + seed, "cpu")[:2]
    rng = np.random.RandomState(seed)
    texts = []
    for reads in (r1.numpy(), r2.numpy()):
        out = []
        for i in range(npairs):
            seq = bytes(reads[i]).decode("ascii")
            q = np.clip(38 - (np.arange(n) * rng.uniform(0.0, 0.08)).astype(int) + rng.randint(-3, 4, size=n), 2, 40)
            out.append("@lp%d/x\n%s\n+\n%s\n" % (i, seq, "".join(chr(int(v) + 33) for v in q)))
        texts.append("".join(out).encode())
    return texts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/oracle_ref")
    args = ap.parse_args()
    sys.path.insert(0, args.scratch)
    from atropos.commands import get_command
    data_dir = os.path.join(args.scratch, "tests", "data")
    if not os.path.isdir(data_dir):
        data_dir = "/root/reference/tests/data"                    # the reference's own test inputs (build container)

    inputs = {}
    for name in sorted({c[0] for c in CASES}):
        path = os.path.join(data_dir, name)
        if os.path.exists(path):
            inputs[name] = open(path, "rb").read()
        elif os.path.exists(path + ".gz"):
            inputs[name] = gzip.open(path + ".gz", "rb").read()
    inputs["synth.fastq"] = synth_fastq(1200, 11).encode()
    inputs["anchor.fastq"] = synth_anchor_fastq(600, 14).encode()
    inputs["longmix.fastq"] = synth_longmix_fastq(150, 19).encode()
    inputs["longad.fastq"] = synth_longad_fastq(300, 29).encode()
    bis = synth_fastq(300, 33).split("\n")                        # reads that start with CAA / CGA (MspI sites) and others
    for i in range(1, len(bis), 4):
        if (i // 4) % 3 == 0 and len(bis[i]) > 6:
            bis[i] = ("CAA" if (i // 4) % 2 else "CGA") + bis[i][3:]
    inputs["bisulf.fastq"] = "\n".join(bis).encode()
    tagged = synth_fastq(200, 31).split("\n")                    # names with a length field (and one that only looks like it)
    for i in range(0, len(tagged) - 1, 4):
        n = len(tagged[i + 1])
        tagged[i] += " length=%d" % n if i % 12 else " xlength=%d length=" % n
        if tagged[i + 2] != "+":
            tagged[i + 2] = "+" + tagged[i][1:]
    inputs["lengthtag.fastq"] = "\n".join(tagged).encode()
    low = synth_fastq(120, 32).split("\n")                       # phred+64 qualities with characters below the base
    for i in range(3, len(low), 4):
        low[i] = "".join(chr(min(126, ord(c) + 31 - (7 if (k % 5 == 0) else 0) - (40 if k % 11 == 3 else 0))) for k, c in enumerate(low[i]))
    inputs["lowqual64.fastq"] = "\n".join(low).encode()
    inputs["synth_dos.fastq"] = synth_fastq(100, 12, dos=True).encode()
    inputs["synth_name2.fastq"] = synth_fastq(100, 13, repeat_name=True).encode()
    small = inputs["small.fastq"]
    inputs["nofinalnewline.fastq"] = small.rstrip(b"\n")
    lines = small.split(b"\n")
    inputs["bad_at.fastq"] = b"\n".join(lines[:4] + [b"X" + lines[4][1:]] + lines[5:])
    inputs["bad_plus.fastq"] = b"\n".join(lines[:6] + [b"-"] + lines[7:])
    inputs["bad_name2.fastq"] = b"\n".join(lines[:2] + [b"+other name"] + lines[3:])
    inputs["bad_length.fastq"] = b"\n".join(lines[:7] + [lines[7][:-3]] + lines[8:])
    inputs["bad_truncated.fastq"] = b"\n".join(lines[:6]) + b"\n"

    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for idx, (name, argstr) in enumerate(CASES):
            in_path = os.path.join(tmp, "in_%d.fastq" % idx)
            out_path = os.path.join(tmp, "out_%d.fastq" % idx)
            open(in_path, "wb").write(inputs[name])
            aux_paths = {kind: os.path.join(tmp, "%s_%d.txt" % (kind, idx)) for kind in ("info", "rest", "wildcard", "too_short", "too_long", "untrimmed")
                         if "{%s}" % kind in argstr}
            filled = argstr
            for kind, path in aux_paths.items():
                filled = filled.replace("{%s}" % kind, path)
            params = filled.split() + ["-se", in_path, "-o", out_path, "--quiet", "--no-default-adapters",
                                                            "--no-cache-adapters"]
            retcode, summary = get_command("trim").execute(params)
            case = dict(input=name, args=argstr, output=None, error=None)
            if aux_paths:
                assert retcode == 0, argstr
                # (a file nothing was written to is not created)
                case["aux"] = {kind: base64.b64encode(open(path, "rb").read() if os.path.exists(path) else b"").decode()
                               for kind, path in aux_paths.items()}
            if retcode != 0:
                # the command only logs the exception: re-raise it by iterating the reference's reader
                from atropos.io._seqio import FastqReader
                try:
                    list(FastqReader(in_path))
                    raise AssertionError("reference failed outside the reader: " + argstr)
                except Exception as err:                                  # noqa: BLE001 (recorded, not handled)
                    if isinstance(err, AssertionError):
                        raise
                    case["error"] = [type(err).__name__, str(err)]
                    if err.__cause__ is not None:
                        case["error"].append(str(err.__cause__))
            else:
                text = open(out_path, "rb").read()
                case["size"] = len(text)
                case["sha256"] = hashlib.sha256(text).hexdigest()
                # full text for small outputs; the head (for a readable diff) for the large ones
                case["output"] = base64.b64encode(text if len(text) <= 20000 else text[:3000]).decode()
            cases.append(case)
            print("%-24s %-70s -> %s" % (name, argstr, "ERROR " + case["error"][1][:50] if case["error"] else
                                         "%d bytes" % case["size"]))
    for name in ("paired.1.fastq", "paired.2.fastq"):
        inputs[name] = open(os.path.join(data_dir, name), "rb").read()
    inputs["synth_pe.1.fastq"], inputs["synth_pe.2.fastq"] = synth_pairs(500, 21)
    inputs["long_pe.1.fastq"], inputs["long_pe.2.fastq"] = synth_long_pairs(160, 27)
    soft = []
    rng = np.random.RandomState(22)
    for text in synth_pairs(300, 23):
        lines = text.decode().split("\n")
        soft.append(lines)
    for r in range(300):                               # lower-case stretches: one mate, both mates, whole reads
        kind = rng.rand()
        for k, lines in enumerate(soft):
            seq = lines[4 * r + 1]
            if kind < 0.15 or (kind < 0.3 and k == 0):
                a = rng.randint(0, max(1, len(seq)))
                b = rng.randint(a, len(seq) + 1)
                seq = seq[:a] + seq[a:b].lower() + seq[b:]
            elif kind < 0.36:
                seq = seq.lower()
            lines[4 * r + 1] = seq
    inputs["soft_pe.1.fastq"], inputs["soft_pe.2.fastq"] = ("\n".join(lines).encode() for lines in soft)
    paired = []
    with tempfile.TemporaryDirectory() as tmp:
        for idx, (n1, n2, argstr) in enumerate(PAIRED_CASES):
            paths = [os.path.join(tmp, "pe_%d_%s.fastq" % (idx, t)) for t in ("in1", "in2", "out1", "out2")]
            open(paths[0], "wb").write(inputs[n1])
            open(paths[1], "wb").write(inputs[n2])
            aux_paths = {kind: os.path.join(tmp, "pe_%s_%d.txt" % (kind, idx))
                         for kind in ("info", "rest", "wildcard", "too_short", "too_short2", "untrimmed", "untrimmed2", "too_long", "too_long2")
                         if "{%s}" % kind in argstr}
            filled = argstr
            for kind, path in aux_paths.items():
                filled = filled.replace("{%s}" % kind, path)
            params = filled.split() + ["-pe1", paths[0], "-pe2", paths[1], "-o", paths[2], "-p", paths[3], "--quiet",
                                       "--no-default-adapters", "--no-cache-adapters"]
            if "-R" in argstr.split():
                paths.append(os.path.join(tmp, "pe_%d_merged.fastq" % idx))
                params += ["--merged-output", paths[4]]
            retcode, _summary = get_command("trim").execute(params)
            assert retcode == 0, (argstr, retcode)
            case = dict(input1=n1, input2=n2, args=argstr, outputs=[])
            if aux_paths:
                case["aux"] = {kind: base64.b64encode(open(path, "rb").read() if os.path.exists(path) else b"").decode()
                               for kind, path in aux_paths.items()}
            for path in paths[2:]:
                text = open(path, "rb").read()
                case["outputs"].append(dict(size=len(text), sha256=hashlib.sha256(text).hexdigest(),
                                            text=base64.b64encode(text if len(text) <= 20000 else text[:3000]).decode()))
            paired.append(case)
            print("%-18s %-90s -> %d + %d bytes" % (n1, argstr[:90], case["outputs"][0]["size"], case["outputs"][1]["size"]))
    doc = dict(inputs={k: base64.b64encode(v).decode() for k, v in inputs.items()}, cases=cases, paired=paired)
    out = os.path.join(HERE, "trim_cases.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps(doc, sort_keys=True).encode())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
