#!/usr/bin/env python3
"""GPU-vs-oracle fuzz campaign (run on the GPU box): every batched kernel path against
oracle/align_oracle.c on random cases -- the filtered pipeline (all adapter types, the 32-row NARROW
pre-pass, equal-length batches with partial overlaps), the full sweep over all 16 flag sets and the wavefront-per-read
kernel (every batch goes through all three families and the automatic choice), the pair aligner (sides of up to 320
bases: full sweep, cost / threat / band pipeline with and without `need`, wavefront per pair), the insert aligner
(reads of up to 320 bases, probed sweep), MultiAligner (a wavefront per pair), the fused linked-adapter
pipeline (and, round 6, the grouped form on the sets inside its envelope), the plane-guided error correction and the fused match + correction call against the two calls, ragged batches at wave-filling size (tail-mode window sweep), the
two-pass pre-pass on bit planes (equal-length and ragged batches against full sweep, one-pass pipeline and oracle) and
reads of 737 .. 4 000 bases (rolling origin base), pairs / references beyond 320 bases (64-bit cells).
usage: tools/gpu_fuzz.py [first_seed] [seeds]   (the log of the round's last run is kept under profiles/)"""
import sys, time
sys.path.insert(0, '.')
from atropos_amd import _lib
_lib.set_backend(_lib.HipBackend(0))
from atropos_amd.align import Aligner, PairAligner, InsertAligner, MultiAligner
from oracle import oracle
from tests import _cases
t0 = time.time()
tot = 0


def rng_top(seed):
    return (150, 100, 250, 300, 64)[seed % 5]     # read lengths of the pair-pipeline rounds (Myers words 5 / 8 / 10)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for seed in range(first, first + (int(sys.argv[2]) if len(sys.argv) > 2 else 4)):
    tot += _cases.check_filtered_pipeline(Aligner, oracle, _lib.AtroposHipError, seed, 500)
    tot += _cases.check_filtered_pipeline(Aligner, oracle, _lib.AtroposHipError, seed + 100, 300, (33, 40), (14, 10, 6, 14))
    tot += _cases.check_uniform_partial_overlaps(Aligner, oracle, _lib.AtroposHipError, seed + 200, 150)
    tot += _cases.check_batches_against_oracle(Aligner, oracle, _lib.AtroposHipError, seed + 300, 300)
    tot += _cases.check_pairs_against_oracle(PairAligner, oracle, _lib.AtroposHipError, seed + 400, 150)
    tot += _cases.check_pairs_fast(PairAligner, oracle, seed + 450, 12, top=rng_top(seed), npairs=96)
    tot += _cases.check_insert_batches_against_oracle(InsertAligner, oracle, seed + 500, 60)
    tot += _cases.check_multi_against_oracle(MultiAligner, oracle, seed + 550, 40)
    tot += _cases.check_linked_sets_against_oracle(oracle, seed + 600, 150, reads_per_round=(1, 64, 65, 200, 700))[0]
    tot += _cases.check_plane_guided_correction(n=20_000, seed=seed + 700)
    tot += _cases.check_fused_match_correct(n=8_192, seed=seed + 720)
    tot += _cases.check_ragged_tail_mode(Aligner, oracle, seed + 800, nreads=40_000, oracle_slice=600)
    tot += _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, seed + 900, 120, 300)[0]
    # round 6: adapters of 41 .. 64 bases, START_WITHIN_SEQ1 (flags 11 / 15), the certificates on long and hostile adapters
    tot += _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, seed + 910, 60, 300, mrange=(41, 64))[0]
    tot += _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, seed + 920, 60, 300, mrange=(20, 64), flag_choices=(11, 15))[0]
    tot += _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, seed + 930, 6, 1500, mrange=(20, 64))
    tot += _cases.check_long_reads(Aligner, oracle, _lib.AtroposHipError, seed + 950, 40)
    tot += _cases.check_long_pairs(Aligner, PairAligner, oracle, seed + 980, 30)
    print(seed, tot, "%.0f s" % (time.time() - t0), "grouped linked sets so far in this seed:", _cases.check_linked_sets_against_oracle.grouped_sets[0], flush=True)
print("cases", tot)
