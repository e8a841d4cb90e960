"""CPU check (lock-step emulation, same per-lane source as the GPU build) of the insert
aligner kernel, the general MultiAligner kernel and compare_prefixes/suffixes."""
from . import _cases
from .conftest import load_golden


def test_golden_insert(emu_backend):
    from atropos_amd.align import InsertAligner
    assert _cases.check_golden_insert(InsertAligner) > 2000


def test_golden_multi_and_compare(emu_backend):
    from atropos_amd import align
    assert _cases.check_golden_multi_compare(align) > 3000


def test_insert_batches(emu_backend, oracle):
    from atropos_amd.align import InsertAligner
    assert _cases.check_insert_batches_against_oracle(InsertAligner, oracle, 5, 27) > 1000


def test_synthetic_heads_c3_c5(emu_backend):
    from atropos_amd import synth
    from atropos_amd.align import InsertAligner
    heads = load_golden("synth_heads.json.gz")
    for name in ("C3", "C5"):
        w = synth.workload(name, 0, heads[name]["count"])
        ia = InsertAligner(w["adapter1"], w["adapter2"], **heads[name]["kw"])
        res = ia.match_insert_batch(w["reads1"], w["reads2"]).results()
        assert [_cases.norm_insert(r) for r in res] == heads[name]["out"]
        assert sum(r is not None for r in res) > heads[name]["count"] // 4


def test_insert_errors(emu_backend):
    import pytest
    from atropos_amd import _lib
    from atropos_amd.align import InsertAligner
    ia = InsertAligner("TTAGACATATGG", "CAGTGGAGTATA")
    with pytest.raises(KeyError):
        ia.match_insert("ACGTACGT", "ACGTXCGT")          # no complement for X (reference: KeyError)
    assert ia.match_insert("ACGTACGT", "acgtacgt") is None      # soft-masked read 2: characters compare as they are
    with pytest.raises(ValueError):
        ia.match_insert_batch(["ACGT"], ["ACXT"])
    with pytest.raises(_lib.AtroposHipError):
        InsertAligner("A" * 129, "ACGT")
    assert ia.match_insert("", "") is None
    assert ia.match_insert("ACGT", "") is None
    _cases.check_read2_validation(ia)


def test_plane_guided_correction(emu_backend):
    assert _cases.check_plane_guided_correction(n=320) == 7 * 320


def test_fused_match_correct_contract(emu_backend):
    """the call's contract on the test double (which composes it from the two steps): shapes, arguments, the method"""
    assert _cases.check_fused_match_correct(n=192) == 13 * 192


def test_correct_errors_fixture(emu_backend):
    assert _cases.check_correct_errors_fixture() == 4000


def test_multi_aligner_against_oracle(emu_backend, oracle):
    from atropos_amd.align import MultiAligner
    assert _cases.check_multi_against_oracle(MultiAligner, oracle, 21, 60) == 2400


def test_insert_list_cap_eight_chunks(emu_backend, oracle):
    from . import _cases
    assert _cases.check_insert_list_cap(oracle) == 160
