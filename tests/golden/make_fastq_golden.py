#!/usr/bin/env python3
"""Generates tests/golden/fastq_fuzz.json.gz: random FASTQ texts (odd names, empty reads,
repeated descriptions, CRLF files, missing final newline, malformed records) and what the
REFERENCE's FastqReader (atropos/io/_seqio.pyx:163-245) makes of them -- the record tuples or the
FormatError.  Run in this container only; the committed file is data.
usage: python tests/golden/make_fastq_golden.py [--scratch /tmp/oracle_ref]"""
import argparse
import base64
import gzip
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_name(rng):
    alphabet = "abcXYZ019:/_-.# @+"
    return "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 40)))


def make_text(rng):
    eol = "\r\n" if rng.random() < 0.25 else "\n"
    nrec = rng.choice([0, 1, 2, 3, 5, 9, 17, 65])
    recs = []
    for _ in range(nrec):
        n = rng.choice([0, 1, 7, 16, 33]) if rng.random() < 0.5 else rng.randint(0, 60)
        name = rand_name(rng)
        seq = "".join(rng.choice("ACGTNacgtn") for _ in range(n))
        qual = "".join(chr(rng.randint(33, 74)) for _ in range(n))
        plus = "+" + (name if rng.random() < 0.2 else "")
        recs.append([("@" + name), seq, plus, qual])
    kind = rng.random()
    if nrec and kind < 0.30:                                   # break something
        r = rng.randrange(nrec)
        what = rng.choice(["at", "plus", "name2", "qlen", "trunc", "mixed_eol", "blank"])
        if what == "at":
            recs[r][0] = rng.choice(["X", ">", ""]) + recs[r][0][1:]
        elif what == "plus":
            recs[r][2] = rng.choice(["-", "", "x+"]) + recs[r][2][1:]
        elif what == "name2":
            recs[r][2] = "+" + recs[r][0][1:] + "x"
        elif what == "qlen":
            recs[r][3] = recs[r][3] + "I" if rng.random() < 0.5 or not recs[r][3] else recs[r][3][:-1]
        elif what == "trunc":
            recs = recs[:r] + [recs[r][:rng.randint(1, 3)]]
        elif what == "mixed_eol":
            recs[r][rng.randrange(4)] += "\r"
        elif what == "blank":
            recs[r].insert(rng.randrange(4), "")
    text = eol.join(eol.join(lines) for lines in recs)
    if recs and rng.random() < 0.8:
        text += eol
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/oracle_ref")
    args = ap.parse_args()
    sys.path.insert(0, args.scratch)
    from atropos.commands import get_command
    get_command("trim")                                    # import order: resolves the io package's import cycle
    from atropos.io.seqio import FastqReader
    rng = random.Random(20260928)
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.fastq")
        for _ in range(300):
            text = make_text(rng).encode("ascii")
            open(path, "wb").write(text)
            case = dict(text=base64.b64encode(text).decode())
            try:
                case["records"] = [[r.name, r.sequence, r.qualities, r.name2] for r in FastqReader(path)]
            except Exception as err:                              # noqa: BLE001 (recorded, not handled)
                case["error"] = [type(err).__name__, str(err)] + ([str(err.__cause__)] if err.__cause__ else [])
            cases.append(case)
    out = os.path.join(HERE, "fastq_fuzz.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps(cases, sort_keys=True).encode())
    print("wrote", out, os.path.getsize(out), "bytes;", len(cases), "texts,", sum("error" in c for c in cases), "errors")


if __name__ == "__main__":
    main()
