"""The C-ABI library loads (no GPU needed for dlopen) and exports every symbol the
headers under include/ declare."""
import ctypes
import os
import re

from .conftest import ROOT


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names.update(re.findall(r"\b(atr_[a-z0-9_]+)\s*\(", text))
    return names


def test_exports_every_declared_symbol():
    from atropos_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in sorted(declared):
        assert hasattr(lib, name), "libatropos_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))


def test_host_only_entry_points():
    """Entry points that do no device work behave without a GPU."""
    from atropos_amd import _lib
    lib = _lib.load_library()
    assert lib.atr_version() >= 100
    assert lib.atr_packed_bytes(65, 150) == 2 * 5 * 64 * 16
    buf = ctypes.create_string_buffer(256)
    assert lib.atr_translate_table(_lib.TABLE_IUPAC, buf) == 0
    assert buf.raw[ord("N")] == 15 and buf.raw[ord("n")] == 15 and buf.raw[ord("X")] == 0
    assert lib.atr_translate_table(_lib.TABLE_ACGT, buf) == 0
    assert buf.raw[ord("U")] == 8 and buf.raw[ord("N")] == 0
    assert lib.atr_translate_table(_lib.TABLE_DNA15, buf) == 0
    assert buf.raw[ord("a")] == 0 and buf.raw[ord("A")] == 1
    h = ctypes.c_void_p()
    assert lib.atr_aligner_create(b"ACGT", 4, 0.1, 14, 0, 0, 0, 1, ctypes.byref(h)) == -1      # min_overlap < 1
    assert lib.atr_aligner_create(b"ACGT", 4, 0.1, 14, 0, 0, 1, 0, ctypes.byref(h)) == -1      # indel_cost < 1
    assert lib.atr_aligner_create(b"A" * 129, 129, 0.1, 14, 0, 0, 1, 1, ctypes.byref(h)) == -2
    assert lib.atr_aligner_create(b"ACGT", 4, 0.1, 14, 0, 0, 1, 1, ctypes.byref(h)) == 0
    assert lib.atr_aligner_query_table(h, buf) == _lib.TABLE_DNA15
    assert lib.atr_aligner_set_min_overlap(h, 0) == -1
    lib.atr_aligner_destroy(h)
    assert lib.atr_aligner_create(b"ACGTN", 5, 0.1, 14, 1, 0, 1, 1, ctypes.byref(h)) == 0
    assert lib.atr_aligner_query_table(h, buf) == _lib.TABLE_ACGT
    lib.atr_aligner_destroy(h)
    assert lib.atr_aligner_create(b"FRONTADAPT", 10, 0.1, 8, 0, 0, 1, 1, ctypes.byref(h)) == 0
    assert lib.atr_aligner_query_table(h, buf) == _lib.TABLE_CUSTOM
    assert buf.raw[ord("F")] == 1 and buf.raw[ord("Z")] == 0
    lib.atr_aligner_destroy(h)


def test_argument_validation_without_device():
    """Every batch entry point validates its arguments before it touches the device, so bad
    calls return ATR_ERR_INVALID / ATR_ERR_UNSUPPORTED even on a machine without a GPU."""
    from atropos_amd import _lib
    lib = _lib.load_library()
    INVALID, UNSUPPORTED = -1, -2
    tab = ctypes.create_string_buffer(256)
    # nothing to do is fine, without pointers
    assert lib.atr_pack_reads(None, 0, None, None, 0, 100, tab, None, None, None) == 0
    assert lib.atr_pack_planes(None, 0, None, None, 0, 100, tab, None, None, None) == 0
    assert lib.atr_locate_pairs_batch(None, None, 150, 0, None, None, 150, 0, 0.2, 15, 0, 0, 1, 1, None, None) == 0
    # negative sizes, missing tables, sizes beyond the envelope
    assert lib.atr_pack_reads(None, 0, None, None, -1, 100, tab, None, None, None) == INVALID
    assert lib.atr_pack_reads(None, 0, None, None, 1, 100, None, None, None, None) == INVALID
    assert lib.atr_pack_reads(None, 0, None, None, 1, _lib.MAX_READ_LEN + 1, tab, None, None, None) == INVALID
    assert lib.atr_locate_batch(None, None, None, 1, 100, None, None, None) == INVALID
    assert lib.atr_insert_match_batch(None, None, None, None, None, 1, 100, None, None) == INVALID
    assert lib.atr_locate_pairs_batch(None, None, 321, 0, None, None, 150, 1, 0.2, 15, 0, 0, 1, 1, None, None) == UNSUPPORTED
    # above 255 bases the cell counts mismatches, which needs STOP_WITHIN_SEQ2 (flag 8)
    assert lib.atr_locate_pairs_batch(None, None, 300, 0, None, None, 150, 1, 0.2, 7, 0, 0, 1, 1, None, None) == UNSUPPORTED
    assert lib.atr_locate_pairs_batch(None, None, 300, 0, None, None, 300, 1, 0.2, 15, 0, 0, 1, 1, None, None) == INVALID
    assert lib.atr_locate_pairs_batch(None, None, 150, 0, None, None, 150, 1, 0.2, 16, 0, 0, 1, 1, None, None) == INVALID
    assert lib.atr_locate_pairs_batch(None, None, 150, 0, None, None, 150, 1, 0.2, 15, 0, 0, 0, 1, None, None) == INVALID
    assert lib.atr_locate_pairs_batch(None, None, 150, 0, None, None, 150, 1, 2.0, 15, 0, 0, 1, 1, None, None) == UNSUPPORTED
    assert lib.atr_fastq_count_lines(None, -1, None, None, None) == INVALID
    assert lib.atr_fastq_count_lines(None, 1 << 33, None, None, None) == INVALID                 # chunk >= 4 GiB
    assert lib.atr_fastq_index(None, 10, None, None, -4, None, None, None) == INVALID
    assert lib.atr_pack_records(None, None, None, None, 1, 100, None, 0, None, None, None, None) == INVALID
    assert lib.atr_clip_batch(None, None, None, 1, -1, 0, None) == INVALID
    assert lib.atr_clip_batch(None, None, None, 1, 0, 1, None) == INVALID                         # back must be <= 0
    assert lib.atr_quality_trim_batch(None, None, None, None, 1, 0, 20, 33, 0, None) == INVALID
    assert lib.atr_nend_trim_batch(None, None, None, None, None, None, 1, None) == INVALID
    assert lib.atr_match_trim_batch(None, None, 0, None, None, None, None, 1, None) == INVALID
    assert lib.atr_read_filter_batch(None, None, None, None, None, None, None, 1, 0, -1, -1.0, 0, 0, None, None, None) == INVALID
    assert lib.atr_pair_filter_batch(None, None, 1, 3, None, None) == INVALID                     # min_affected is 1 or 2
    assert lib.atr_insert_plan_batch(None, None, None, None, None, None, None, None, None, None, None, None, None, 1, 1,
                                     1, 3, -1, 1, None, None, None, None, None, None) == INVALID  # trim_action 0..2
    assert lib.atr_fastq_emit(None, None, None, None, None, None, None, 0, 1, 0, None, None, None, None) == INVALID
    assert lib.atr_adapter_postfilter(None, 0, 10, 3, 0.1, None, 0, 0.0, 0, None) == 0
    assert lib.atr_fastq_work_bytes(-5) == 0 and lib.atr_fastq_emit_work_bytes(-5) == 0
    assert lib.atr_fastq_work_bytes(1 << 20) > 0 and lib.atr_locate_work_bytes(1000) > 8000
