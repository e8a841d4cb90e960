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
