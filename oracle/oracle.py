"""ctypes front-end of the CPU oracle (oracle/align_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg; never from ``atropos_amd``.  Parity
status: pinned against the reference (see align_oracle.c header).

The Python-level pieces restated here (reference file:line):
  rmp()            atropos/util/__init__.py:117-155 (RandomMatchProbability)
  InsertOracle     atropos/align/__init__.py:206-233 (constructor defaults)
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "align_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


class _InsertParams(C.Structure):
    _fields_ = [
        ("adapter1", C.c_char_p), ("alen1", C.c_int),
        ("adapter2", C.c_char_p), ("alen2", C.c_int),
        ("insert_max_rmp", C.c_double), ("adapter_max_rmp", C.c_double),
        ("min_insert_overlap", C.c_int), ("max_insert_mismatch_frac", C.c_double),
        ("min_adapter_overlap", C.c_int), ("max_adapter_mismatch_frac", C.c_double),
        ("adapter_check_cutoff", C.c_int),
        ("adapter_wildcards", C.c_int), ("read_wildcards", C.c_int),
        ("rmp_insert", C.POINTER(C.c_double)), ("rmp_adapter", C.POINTER(C.c_double)),
        ("rmp_ld", C.c_int),
        ("max_mismatch_by_alen", C.POINTER(C.c_int)),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        I6 = C.POINTER(C.c_int)
        L.orc_locate.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, I6]
        L.orc_locate.restype = C.c_int
        L.orc_compare_prefixes.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, I6]
        L.orc_compare_prefixes.restype = None
        L.orc_compare_suffixes.argtypes = L.orc_compare_prefixes.argtypes
        L.orc_compare_suffixes.restype = None
        L.orc_multi_locate.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int,
                                       C.c_int, C.c_int, I6]
        L.orc_multi_locate.restype = C.c_int
        L.orc_reverse_complement.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.orc_reverse_complement.restype = C.c_int
        L.orc_match_insert.argtypes = [C.POINTER(_InsertParams), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                       I6, I6, I6, I6]
        L.orc_match_insert.restype = C.c_int
        L.orc_locate_many.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                      C.c_void_p, C.c_int]
        L.orc_locate_many.restype = C.c_int
        L.orc_acgt_table.argtypes = [C.c_char_p]
        L.orc_iupac_table.argtypes = [C.c_char_p]
        _lib = L
    return _lib


def _b(s):
    return s if isinstance(s, bytes) else s.encode("latin-1")


def locate(ref, query, e, flags=15, wildcard_ref=False, wildcard_query=False,
           min_overlap=1, indel_cost=1):
    """Aligner(ref, e, flags, wildcard_ref, wildcard_query, min_overlap,
    indel_cost).locate(query) -> 6-tuple or None."""
    out = (C.c_int * 6)()
    r, q = _b(ref), _b(query)
    rc = lib().orc_locate(r, len(r), q, len(q), e, flags, int(wildcard_ref), int(wildcard_query),
                          min_overlap, indel_cost, out)
    if rc < 0:
        raise MemoryError
    return tuple(out) if rc else None


def compare_prefixes(ref, query, wildcard_ref=False, wildcard_query=False):
    out = (C.c_int * 6)()
    r, q = _b(ref), _b(query)
    lib().orc_compare_prefixes(r, len(r), q, len(q), int(wildcard_ref), int(wildcard_query), out)
    return tuple(out)


def compare_suffixes(ref, query, wildcard_ref=False, wildcard_query=False):
    out = (C.c_int * 6)()
    r, q = _b(ref), _b(query)
    lib().orc_compare_suffixes(r, len(r), q, len(q), int(wildcard_ref), int(wildcard_query), out)
    return tuple(out)


def multi_locate(ref, query, e, flags=15, min_overlap=1, max_matches=100):
    """MultiAligner(e, flags, min_overlap).locate(ref, query, max_matches)."""
    r, q = _b(ref), _b(query)
    out = (C.c_int * (6 * (max_matches + len(r) + 2)))()
    n = lib().orc_multi_locate(r, len(r), q, len(q), e, flags, min_overlap, max_matches, out)
    if n < 0:
        raise MemoryError
    if n == 0:
        return None
    return [tuple(out[6 * t:6 * t + 6]) for t in range(n)]


def reverse_complement(seq):
    s = _b(seq)
    dst = C.create_string_buffer(len(s) + 1)
    if lib().orc_reverse_complement(s, len(s), dst) != 0:
        raise KeyError("base without complement")
    return dst.raw[:len(s)].decode("latin-1")


# ---- RandomMatchProbability (util/__init__.py:117-155) -----------------------

_FACT = [1, 1]


def _fact(n):
    while len(_FACT) <= n:
        _FACT.append(_FACT[-1] * len(_FACT))
    return _FACT[n]


def rmp(matches, size, match_prob=0.25, mismatch_prob=0.75):
    """Binomial tail sum_{i=matches..size} C(size,i) p^i q^(size-i), evaluated in
    the reference's order: bigint factorials, true division (floor division on
    OverflowError), left-to-right float accumulation."""
    if matches == size:
        return match_prob ** matches
    nfac = _fact(size)
    prob = 0.0
    for i in range(matches, size + 1):
        j = size - i
        try:
            div = nfac / _fact(i) / _fact(j)
        except OverflowError:
            div = nfac // _fact(i) // _fact(j)
        prob += (mismatch_prob ** j) * (match_prob ** i) * div
    return prob


_TABLE_CACHE = {}


def rmp_table(max_size, match_prob=0.25, mismatch_prob=0.75):
    """[size][matches] table of rmp(), leading dimension max_size+1; entries with
    matches > size are 0.  Every entry is the same left-to-right float sum rmp()
    computes (term_k + term_k+1 + ...), evaluated for all k of one size at once."""
    import numpy as np
    key = (max_size, match_prob, mismatch_prob)
    if key in _TABLE_CACHE:
        return _TABLE_CACHE[key]
    ld = max_size + 1
    t = np.zeros((ld, ld), dtype=np.float64)
    for size in range(ld):
        nfac = _fact(size)
        terms = np.empty(size + 1, dtype=np.float64)
        for i in range(size + 1):
            j = size - i
            try:
                div = nfac / _fact(i) / _fact(j)
            except OverflowError:
                div = nfac // _fact(i) // _fact(j)
            terms[i] = (mismatch_prob ** j) * (match_prob ** i) * div
        acc = 0.0 + terms                      # prob = 0.0; prob += term_k
        for d in range(1, size + 1):
            acc[:size + 1 - d] += terms[d:]
        acc[size] = match_prob ** size         # matches == size shortcut
        t[size, :size + 1] = acc
    _TABLE_CACHE[key] = t
    return t


class InsertOracle(object):
    """InsertAligner(adapter1, adapter2, ...).match_insert(seq1, seq2) twin.
    Returns None or (insert_tuple, m1, m2) with m1/m2 6-tuples
    (astart, astop, rstart, rstop, matches, errors) or None."""

    def __init__(self, adapter1, adapter2, insert_max_rmp=1e-6, adapter_max_rmp=0.001,
                 min_insert_overlap=1, max_insert_mismatch_frac=0.2,
                 min_adapter_overlap=1, max_adapter_mismatch_frac=0.2,
                 adapter_check_cutoff=9, base_probs=None,
                 adapter_wildcards=True, read_wildcards=False, max_len=320):
        import numpy as np
        self._a1, self._a2 = _b(adapter1), _b(adapter2)
        bp = base_probs or dict(match_prob=0.25, mismatch_prob=0.75)
        ld = max(max_len, len(self._a1), len(self._a2)) + 1
        self._ins = np.ascontiguousarray(rmp_table(ld - 1, **bp))
        self._ada = self._ins if base_probs is None else np.ascontiguousarray(rmp_table(ld - 1))
        frac = float(max_adapter_mismatch_frac)
        self._mm = np.array([round(a * frac) for a in range(ld)], dtype=np.int32)
        p = _InsertParams()
        p.adapter1, p.alen1 = self._a1, len(self._a1)
        p.adapter2, p.alen2 = self._a2, len(self._a2)
        p.insert_max_rmp, p.adapter_max_rmp = insert_max_rmp, adapter_max_rmp
        p.min_insert_overlap = min_insert_overlap
        p.max_insert_mismatch_frac = float(max_insert_mismatch_frac)
        p.min_adapter_overlap = min_adapter_overlap
        p.max_adapter_mismatch_frac = frac
        p.adapter_check_cutoff = adapter_check_cutoff
        p.adapter_wildcards, p.read_wildcards = int(adapter_wildcards), int(read_wildcards)
        p.rmp_insert = self._ins.ctypes.data_as(C.POINTER(C.c_double))
        p.rmp_adapter = self._ada.ctypes.data_as(C.POINTER(C.c_double))
        p.rmp_ld = ld
        p.max_mismatch_by_alen = self._mm.ctypes.data_as(C.POINTER(C.c_int))
        self._p = p
        self._ld = ld

    def match_insert(self, seq1, seq2):
        s1, s2 = _b(seq1), _b(seq2)
        if min(len(s1), len(s2)) >= self._ld:
            raise ValueError("read longer than the oracle's RMP table")
        ins = (C.c_int * 6)()
        m1 = (C.c_int * 6)()
        m2 = (C.c_int * 6)()
        has = (C.c_int * 6)()
        rc = lib().orc_match_insert(C.byref(self._p), s1, len(s1), s2, len(s2), ins, m1, m2, has)
        if rc == -1:
            raise KeyError("base without complement")
        if rc < 0:
            raise MemoryError
        if rc == 0:
            return None
        return (tuple(ins), tuple(m1) if has[0] else None, tuple(m2) if has[1] else None)


def locate_many(ref, reads, lens, e, flags, wildcard_ref=False, wildcard_query=False,
                min_overlap=1, indel_cost=1, nthreads=1):
    """reads: uint8 numpy [nreads, stride] of ASCII; lens: int32 [nreads].
    Returns int32 [nreads, 6]; row[1] == -1 marks None."""
    import numpy as np
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    out = np.empty((reads.shape[0], 6), dtype=np.int32)
    r = _b(ref)
    rc = lib().orc_locate_many(r, len(r), e, flags, int(wildcard_ref), int(wildcard_query), min_overlap,
                               indel_cost, reads.ctypes.data, lens.ctypes.data, reads.shape[1],
                               reads.shape[0], out.ctypes.data, nthreads)
    if rc != 0:
        raise MemoryError
    return out
