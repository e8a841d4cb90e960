#!/usr/bin/env python3
"""Token-run similarity of the product's Python files against the reference tree
(container only: needs /root/reference).  A token of a repo file counts as "copied" when it
lies inside a run of >= RUN identical tokens (comments, docstrings and layout removed) that
also occurs somewhere in the reference's Python / Cython sources.  Prints the share per file.

    python tools/copycheck.py [--run 12] [--ref /root/reference/atropos] [files ...]
"""
import argparse
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tokens_of(path):
    """Significant tokens of a Python (or Cython, best effort) source file."""
    with open(path, "rb") as fh:
        src = fh.read().decode("utf-8", "replace")
    out = []
    try:
        prev_sig = None
        for tok in tokenize.generate_tokens(io.StringIO(src).readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                            tokenize.ENCODING, tokenize.ENDMARKER):
                if tok.type == tokenize.NEWLINE:
                    prev_sig = None
                continue
            if tok.type == tokenize.STRING and prev_sig is None:
                continue                      # a statement that starts with a string: docstring
            out.append(tok.string)
            prev_sig = tok.string
    except (tokenize.TokenError, IndentationError, SyntaxError):
        # Cython sources do not always tokenize as Python: fall back to a crude splitter
        import re
        src = re.sub(r'"""(?:.|\n)*?"""', "", src)
        src = re.sub(r"#[^\n]*", "", src)
        out = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+(?:\.\d+)?|[^\sA-Za-z_0-9]", src)
    return out


def reference_grams(ref_dir, run):
    grams = set()
    for dirpath, _, files in os.walk(ref_dir):
        for name in files:
            if name.endswith((".py", ".pyx", ".pxd")):
                t = tokens_of(os.path.join(dirpath, name))
                for i in range(len(t) - run + 1):
                    grams.add(hash(tuple(t[i:i + run])))
    return grams


def share(path, grams, run, show=False):
    t = tokens_of(path)
    hit = [False] * len(t)
    for i in range(len(t) - run + 1):
        if hash(tuple(t[i:i + run])) in grams:
            for k in range(i, i + run):
                hit[k] = True
    if show:
        i = 0
        while i < len(t):
            if hit[i]:
                j = i
                while j < len(t) and hit[j]:
                    j += 1
                print("    [%d tokens] %s" % (j - i, " ".join(t[i:j])[:300]))
                i = j
            else:
                i += 1
    return (sum(hit) / len(t) if t else 0.0), len(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", type=int, default=12)
    ap.add_argument("--ref", default="/root/reference/atropos")
    ap.add_argument("--show", action="store_true", help="print the matching token runs")
    ap.add_argument("files", nargs="*")
    args = ap.parse_args()
    files = args.files
    if not files:
        for dirpath, _, names in os.walk(os.path.join(ROOT, "atropos_amd")):
            files += [os.path.join(dirpath, n) for n in names if n.endswith(".py")]
    grams = reference_grams(args.ref, args.run)
    worst = 0.0
    for f in sorted(files):
        s, n = share(f, grams, args.run, args.show)
        worst = max(worst, s)
        print("%5.1f %%  %6d tokens  %s" % (100 * s, n, os.path.relpath(f, ROOT)))
    print("worst: %.1f %%" % (100 * worst))
    return 0


if __name__ == "__main__":
    sys.exit(main())
