"""CPU check of the gfx950 locate kernel's arithmetic and control flow through the
lock-step emulation in tests/emu (same per-lane source as the GPU build), against
the golden vectors and the oracle.  The GPU parity tests proper are in
test_gpu_locate.py (-m gpu)."""
import pytest

from . import _cases
from .conftest import load_golden, tup


def test_golden_locate(emu_backend):
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    checked, unsupported = _cases.check_golden_locate(Aligner, _lib.AtroposHipError)
    assert checked > 6000
    assert unsupported < 200          # only ">15 distinct symbols compared literally"


def test_batches_all_flags(emu_backend, oracle):
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_batches_against_oracle(Aligner, oracle, _lib.AtroposHipError, 11, 150) > 10000


def test_filtered_pipeline(emu_backend, oracle):
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_filtered_pipeline(Aligner, oracle, _lib.AtroposHipError, 3, 250) > 20000


def test_filtered_pipeline_narrow(emu_backend, oracle):
    """33..40-base adapters without START_WITHIN_SEQ1: the pre-pass sweeps 32 rows only
    (filter_core.hpp, NARROW mode)."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_filtered_pipeline(Aligner, oracle, _lib.AtroposHipError, 5, 60, (33, 40), (14, 10, 6, 14)) > 5000


def test_long_reads(emu_backend, oracle):
    from atropos_amd.align import Aligner
    from atropos_amd import _lib
    assert _cases.check_golden_long_reads(Aligner, _lib.AtroposHipError) > 600          # the reference's answers
    assert _cases.check_long_reads(Aligner, oracle, _lib.AtroposHipError, 29, 15) > 300


def test_long_pairs(emu_backend, oracle):
    """References / per-pair sides beyond 320 bases: the 64-bit cell path (pairs_long_core.hpp)."""
    from atropos_amd.align import Aligner, PairAligner
    from atropos_amd import _lib
    assert _cases.check_golden_long_pairs(Aligner, PairAligner, _lib.AtroposHipError) == 400
    assert _cases.check_long_pairs(Aligner, PairAligner, oracle, 37, 10) > 100


def test_piece_pipeline(emu_backend, oracle):
    """The two-pass pre-pass on plane64 reads (piece_core.hpp): pieces, read-end conditions, windows, the full
    sweep of the reads that need more than a window -- the per-lane source of the GPU build against the oracle."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 17, 120)
    assert total > 12000 and refused < 60


def test_piece_pipeline_long_adapters(emu_backend, oracle):
    """Round 6: adapters of 41 .. 64 bases in the two-pass pre-pass -- pass B sweeps their first 32 rows (extended NARROW
    mode, the tail compare over up to four dwords), up to eight body pieces, k <= 6, 96-column windows."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 101, 110, 150, mrange=(41, 64))
    assert total > 10000 and refused < 50


def test_piece_pipeline_start_within_seq1(emu_backend, oracle):
    """Round 6: START_WITHIN_SEQ1 in the two-pass pre-pass (the 5' and "anywhere" adapter types, flags 11 / 15; adapters of
    up to 32 bases): read-start conditions mirrored from the read end, windows from column 0 swept from the all-zero column,
    short reads of ragged batches through the full sweep."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 301, 110, 150, mrange=(20, 64),
                                                 flag_choices=(11, 15, 11, 15))
    assert total > 9000 and refused < 60


def test_uniform_partial_overlaps(emu_backend, oracle):
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_uniform_partial_overlaps(Aligner, oracle, _lib.AtroposHipError, 7, 25, 200) > 4000


def test_synthetic_heads(emu_backend):
    from atropos_amd import synth
    from atropos_amd.align import Aligner
    heads = load_golden("synth_heads.json.gz")
    for name in ("C1", "C2"):
        w = synth.workload(name, 0, heads[name]["count"])
        al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
        for path in ("auto", "filtered", "full", "wave"):
            assert al.locate_batch(w["reads"], path=path).tuples() == [tup(x) for x in heads[name]["out"]], path


def test_api_surface(emu_backend, oracle):
    from atropos_amd import align
    # reference tests/test_align.py:13-22 (smoke calls, incl. a 100 % error rate)
    a = align.Aligner("CTCCAGCTTAGACATATC", 0.1, flags=14)
    assert a.locate("CC") == oracle.locate("CTCCAGCTTAGACATATC", "CC", 0.1, 14)
    assert align.Aligner("GCTTAGACATATC", 1.0, flags=14).locate("CAA") == oracle.locate("GCTTAGACATATC", "CAA", 1.0, 14)
    with pytest.raises(ValueError):
        align.Aligner("ACGT", 0.1, min_overlap=0)
    with pytest.raises(ValueError):
        align.Aligner("ACGT", 0.1, indel_cost=0)
    a = align.Aligner("ACGT", 0.1)
    with pytest.raises(ValueError):
        a.min_overlap = 0
    with pytest.raises(ValueError):
        a.indel_cost = 0
    with pytest.raises(UnicodeEncodeError):
        a.locate("ACé")
    a.min_overlap = 3
    assert a.min_overlap == 3
    import pickle
    b = pickle.loads(pickle.dumps(align.Aligner("TCGTATGCCGTCTTC", 0.2, 14, False, False, 3, 1)))
    assert b.locate("TCGTATGCCCTCC") == (0, 15, 0, 12, 12, 3)
    assert align.locate("TCGTATGCCGTCTTC", "TCGTATGCCCTCC", 0.2, 14) == (0, 15, 0, 12, 12, 3)
    assert align.Aligner("", 0.1).locate("ACGT") is None
    assert align.Aligner("ACGT", 0.1).locate("") is None
    assert a.reference == b"ACGT"
    assert align.Aligner("ACGN", 0.1, wildcard_ref=True).reference == bytes([1, 2, 4, 15])
    # changing the indel cost changes the result (issue #80 alignment needs indels)
    c = align.Aligner("TCGTATGCCGTCTTC", 0.2, 14, min_overlap=3)
    c.indel_cost = 100000
    assert c.locate("TCGTATGCCCTCC") != (0, 15, 0, 12, 12, 3)
    # batches packed for one table are rejected by an aligner that needs another
    batch = a.pack(["ACGT", "TTTT"])
    with pytest.raises(ValueError):
        align.Aligner("ACGN", 0.1, wildcard_ref=True).locate_batch(batch)
    assert len(a.locate_batch([])) == 0


def test_pair_aligner_against_oracle(emu_backend, oracle):
    from atropos_amd.align import PairAligner
    from atropos_amd._lib import AtroposHipError
    assert _cases.check_pairs_against_oracle(PairAligner, oracle, AtroposHipError, seed=77, rounds=60) > 1500


def test_dpmatrix_debug(emu_backend):
    from atropos_amd.align import Aligner
    assert _cases.check_dpmatrix_golden(Aligner) == 90


def test_ragged_tail_mode(emu_backend, oracle):
    from atropos_amd.align import Aligner
    assert _cases.check_ragged_tail_mode(Aligner, oracle, 3, nreads=3000, oracle_slice=400) == 15000


def test_pair_aligner_lds_column_fallback(emu_backend, oracle):
    """The LDS-column pair kernel only runs where stream-ordered allocation is missing; keep its per-lane code
    (locate_pair_one) under test as well."""
    import ctypes as C
    from atropos_amd._lib import AtroposHipError
    from atropos_amd.align import PairAligner
    flag = C.c_int.in_dll(emu_backend.lib, "emu_pairs_use_lds_column")
    flag.value = 1
    try:
        assert _cases.check_pairs_against_oracle(PairAligner, oracle, AtroposHipError, seed=79, rounds=25) > 500
    finally:
        flag.value = 0


def test_long_reference(emu_backend):
    """References of 129 .. 320 bases: Aligner.locate through the per-pair aligner."""
    from atropos_amd.align import Aligner
    from atropos_amd._lib import AtroposHipError
    from oracle import oracle
    assert _cases.check_long_reference(Aligner, oracle, AtroposHipError) > 400


def test_pairs_fast_pipeline(emu_backend, oracle):
    """pairs_fast_core.hpp (bit-vector costs, threat analysis, banded payload DP) through the CPU twin: the settings
    the fast pipeline takes, with and without a bound on the matches, three mask-word counts, and -- widen -- the
    banded pass run wider and longer than a pair needs, as in a wave whose other lanes need more."""
    import ctypes as C
    from atropos_amd.align import PairAligner
    stats = (C.c_longlong * 4).in_dll(emu_backend.lib, "emu_pairs_fast_stats")
    widen = C.c_int.in_dll(emu_backend.lib, "emu_pairs_fast_widen")
    for i in range(4):
        stats[i] = 0
    try:
        for w, (seed, rounds, top, npairs) in enumerate(((5, 25, 150, 64), (6, 8, 250, 48), (7, 6, 320, 40), (8, 30, 40, 96))):
            widen.value = w % 2
            assert _cases.check_pairs_fast(PairAligner, oracle, seed, rounds, top=top, npairs=npairs) == rounds * npairs
    finally:
        widen.value = 0
    none, band, full, _ = list(stats)
    assert band > 2000 and none > 1000 and full < band          # the banded pass is what most pairs with an alignment take


def test_long_reference_envelope(emu_backend, oracle):
    from atropos_amd.align import Aligner, PairAligner
    assert _cases.check_long_reference_envelope(Aligner, PairAligner, oracle) > 250


def test_certificates_against_oracle(emu_backend, oracle):
    """The pre-pass's DP-free decisions (perfect-overlap and single-substitution certificates) on adapters and flanks
    built to break them: the CPU twin of the two-pass pipeline against the oracle."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, 31, 30) > 9000
    # round 6: adapters of 41 .. 64 bases (two-word sweep, filter_overlap_certificates64), perfect partial adapters behind hostile flanks
    assert _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, 61, 25, 300, mrange=(41, 64)) > 5000
