"""GPU: first mismatch of the two-pass pre-pass against the full sweep on check_piece_pipeline's cases."""
import random, sys
sys.path.insert(0, ".")
import numpy as np
from atropos_amd import _lib
from atropos_amd.align import Aligner
from tests import _cases
seed, rounds, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = random.Random(seed)
bad = 0
for rnd in range(rounds):
    m = rng.randint(20, 40)
    ref = _cases.rseq(rng, m, "ACGT" if rng.random() < 0.9 else "ACGTN")
    flags = rng.choice([14, 14, 14, 10]); e = rng.choice([0, 0.03, 0.05, 0.08, 0.1, 0.1, 0.1, 0.12])
    ic = rng.choice([1, 1, 1, 2, 100000]); mo = rng.choice([1, 3, 3, 5, 12, 25])
    wr, wq = rng.random() < 0.2, rng.random() < 0.25
    try:
        al = Aligner(ref, e, flags, wr, wq, mo, ic)
    except _lib.AtroposHipError:
        continue
    n = rng.choice((70, 100, 128, 150, 150, 160, 180, 250, 300))
    reads = _cases.piece_reads(rng, ref, n, count, e)
    mat = np.frombuffer("".join(reads).encode(), np.uint8).reshape(len(reads), n).copy()
    try:
        planes = al.pack(mat, layout="plane64")
    except _lib.AtroposHipError:
        continue
    got = al.locate_batch(planes).tuples()
    exp = al.locate_batch(al.pack(mat, layout="tile64"), path="full").tuples()
    diff = [i for i in range(len(reads)) if got[i] != exp[i]]
    if diff:
        bad += 1
        print("round", rnd, "ref", ref, "m", m, "e", e, "flags", flags, "ic", ic, "mo", mo, "wr", wr, "wq", wq, "n", n, "ndiff", len(diff))
        for i in diff[:4]:
            print("  ", i, reads[i], got[i], exp[i])
        if bad >= 3:
            break
print("bad rounds", bad)
