"""ctypes front-end of the CPU oracle (oracle/align_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg; never from ``atropos_amd``.  Parity
status: pinned against the reference (see align_oracle.c header).

The Python-level pieces restated here (reference file:line):
  match_to / linked_many   atropos/adapters/__init__.py:338-400, :671-690 (in align_oracle.c)
  rmp()            atropos/util/__init__.py:117-155 (RandomMatchProbability)
  InsertOracle     atropos/align/__init__.py:206-233 (constructor defaults)
  correct_errors / insert_correct_many   atropos/commands/trim/modifiers.py:219-350, :397-404 (in align_oracle.c)
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
        L.orc_match_to.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_int, I6]
        L.orc_match_to.restype = C.c_int
        L.orc_linked_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]
        L.orc_linked_many.restype = C.c_int
        L.orc_match_insert_many.argtypes = [C.POINTER(_InsertParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        L.orc_match_insert_many.restype = C.c_int
        L.orc_correct_errors.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, I6, C.c_int,
                                         C.c_int, C.c_int, I6, I6]
        L.orc_correct_errors.restype = C.c_int
        L.orc_insert_correct_many.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                                                 C.c_void_p, C.c_int]
        L.orc_insert_correct_many.restype = C.c_int
        L.orc_quality_trim_index.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, I6, I6]
        L.orc_quality_trim_index.restype = None
        L.orc_nextseq_trim_index.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_nextseq_trim_index.restype = C.c_int
        L.orc_n_end_trim.argtypes = [C.c_char_p, C.c_int, I6, I6]
        L.orc_n_end_trim.restype = None
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
    if rc == -2:
        raise AssertionError("empty alignment (_align.pyx:490)")
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


def match_to(seq, flags, read, e, min_overlap=3, indel_cost=1, adapter_wildcards=True, read_wildcards=False):
    """Adapter(seq, flags, e, min_overlap, read_wildcards, adapter_wildcards, indel_cost=...).match_to(read)
    for an adapter with indels and no RMP filter -> (astart, astop, rstart, rstop, matches, errors) or
    None.  The constructor's overrides are applied here: ACGT-only adapters have no wildcards
    (adapters/__init__.py:268-270), min_overlap is capped at the adapter length (:285)."""
    s, r = _b(seq), _b(read)
    aw = bool(adapter_wildcards) and not set(seq) <= set("ACGT")
    out = (C.c_int * 6)()
    rc = lib().orc_match_to(s, len(s), flags, r, len(r), e, min(min_overlap, len(s)), indel_cost, int(aw),
                            int(read_wildcards), out)
    if rc < 0:
        raise MemoryError
    return tuple(out) if rc else None


class _LinkedParams(C.Structure):
    _fields_ = [("nad", C.c_int), ("fronts", C.POINTER(C.c_char_p)), ("flens", C.POINTER(C.c_int)),
                ("backs", C.POINTER(C.c_char_p)), ("blens", C.POINTER(C.c_int)), ("e", C.c_double),
                ("min_overlap", C.c_int), ("indel_cost", C.c_int), ("read_wildcards", C.c_int),
                ("front_wildcards", C.POINTER(C.c_int)), ("back_wildcards", C.POINTER(C.c_int))]


def linked_many(fronts, backs, reads, lens, e, min_overlap=3, indel_cost=1, adapter_wildcards=True,
                read_wildcards=False, nthreads=1):
    """AdapterCutter._best_match over LinkedAdapter(front, back).match_to for a batch (reads: uint8
    numpy [n, stride] ASCII, lens int32 [n]).  Returns (which int8 [n, 2] = (first matching adapter
    or -1, number of matching 5' parts), front int32 [n, 6], back int32 [n, 6]; [:, 1] == -1: None)."""
    import numpy as np
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n = reads.shape[0]
    nad = len(fronts)
    fb, bb = [_b(x) for x in fronts], [_b(x) for x in backs]
    p = _LinkedParams()
    p.nad = nad
    p.fronts = (C.c_char_p * nad)(*fb)
    p.backs = (C.c_char_p * nad)(*bb)
    p.flens = (C.c_int * nad)(*[len(x) for x in fb])
    p.blens = (C.c_int * nad)(*[len(x) for x in bb])
    p.e, p.min_overlap, p.indel_cost, p.read_wildcards = e, min_overlap, indel_cost, int(read_wildcards)
    p.front_wildcards = (C.c_int * nad)(*[int(bool(adapter_wildcards) and not set(x) <= set("ACGT")) for x in fronts])
    p.back_wildcards = (C.c_int * nad)(*[int(bool(adapter_wildcards) and not set(x) <= set("ACGT")) for x in backs])
    which = np.empty((n, 2), dtype=np.int8)
    front = np.empty((n, 6), dtype=np.int32)
    back = np.empty((n, 6), dtype=np.int32)
    rc = lib().orc_linked_many(C.addressof(p), reads.ctypes.data, lens.ctypes.data, reads.shape[1], n,
                               which.ctypes.data, front.ctypes.data, back.ctypes.data, nthreads)
    if rc != 0:
        raise MemoryError
    return which, front, back


def match_insert_many(orc, reads1, lens1, reads2, lens2, nthreads=1):
    """InsertOracle.match_insert for a batch of equal-stride uint8 matrices; int32 [n, 3, 6] with
    [:, t, 1] == -1 for None (t = 0 insert match, 1 / 2 the adapter matches)."""
    import numpy as np
    reads1 = np.ascontiguousarray(reads1, dtype=np.uint8)
    reads2 = np.ascontiguousarray(reads2, dtype=np.uint8)
    assert reads1.shape == reads2.shape
    lens1 = np.ascontiguousarray(lens1, dtype=np.int32)
    lens2 = np.ascontiguousarray(lens2, dtype=np.int32)
    out = np.empty((reads1.shape[0], 3, 6), dtype=np.int32)
    rc = lib().orc_match_insert_many(C.byref(orc._p), reads1.ctypes.data, lens1.ctypes.data, reads2.ctypes.data,
                                     lens2.ctypes.data, reads1.shape[1], reads1.shape[0], out.ctypes.data, nthreads)
    if rc != 0:
        raise KeyError("base without complement (or allocation failure)")
    return out


CORRECT_ACTIONS = {"N": 0, "conservative": 1, "liberal": 2}
_CORRECT_ERRORS = {-1: KeyError, -2: IndexError, -3: ValueError, -4: MemoryError}


def correct_errors(seq1, qual1, seq2, qual2, insert_match, mismatch_action, min_qual_difference=1,
                   truncate_seqs=False):
    """ErrorCorrectorMixin(mismatch_action, min_qual_difference).correct_errors(read1, read2, insert_match,
    truncate_seqs) on fresh reads (corrected == 0).  Returns (sequence1, qualities1, sequence2, qualities2,
    (changed1, changed2)) as the reads hold them afterwards; raises what the reference raises."""
    s1, s2 = C.create_string_buffer(_b(seq1), len(seq1) + 1), C.create_string_buffer(_b(seq2), len(seq2) + 1)
    q1 = C.create_string_buffer(_b(qual1), len(qual1) + 1) if qual1 else None
    q2 = C.create_string_buffer(_b(qual2), len(qual2) + 1) if qual2 else None
    if q1 is None or q2 is None:
        qa = qb = None
    else:
        qa, qb = q1, q2
    im = (C.c_int * 6)(*[int(x) for x in insert_match[:4]], 0, 0)
    changed, newlen = (C.c_int * 6)(), (C.c_int * 6)()
    rc = lib().orc_correct_errors(s1, qa, len(seq1), s2, qb, len(seq2), im, CORRECT_ACTIONS[mismatch_action],
                                  min_qual_difference, int(truncate_seqs), changed, newlen)
    if rc < 0:
        raise _CORRECT_ERRORS[rc]("correct_errors")
    o1 = s1.raw[:newlen[0]].decode("ascii")
    o2 = s2.raw[:newlen[1]].decode("ascii")
    # qualities are only rewritten when the reference has them (has_quals) and the read changed
    oq1 = q1.raw[:newlen[0] if (qa is not None and changed[0]) else len(qual1)].decode("ascii") if q1 is not None else qual1
    oq2 = q2.raw[:len(qual2)].decode("ascii") if q2 is not None else qual2
    return o1, oq1, o2, oq2, (changed[0], changed[1])


def insert_correct_many(records, seq1, qual1, lens1, seq2, qual2, lens2, mismatch_action, min_qual_difference=1,
                        nthreads=1):
    """The correction step of InsertAdapterCutter.__call__ over a batch, IN PLACE on the four uint8
    numpy matrices (same shape, C-contiguous): pairs whose insert match (records int32 [n, 3, 6] of
    match_insert_many) has errors > 0 are corrected with truncate_seqs=True.  Returns (changed int32
    [n, 2], newlen int32 [n, 2]); changed[:, 0] in (-1, -2, -3): KeyError / IndexError / ValueError."""
    import numpy as np
    n = seq1.shape[0]
    for a in (seq1, seq2, qual1, qual2):
        assert a is None or (a.dtype == np.uint8 and a.flags.c_contiguous and a.shape == seq1.shape)
    records = np.ascontiguousarray(records, dtype=np.int32)
    lens1 = np.ascontiguousarray(lens1, dtype=np.int32)
    lens2 = np.ascontiguousarray(lens2, dtype=np.int32)
    changed = np.zeros((n, 2), dtype=np.int32)
    newlen = np.zeros((n, 2), dtype=np.int32)
    rc = lib().orc_insert_correct_many(records.ctypes.data, seq1.ctypes.data, None if qual1 is None else qual1.ctypes.data,
                                       lens1.ctypes.data, seq2.ctypes.data, None if qual2 is None else qual2.ctypes.data,
                                       lens2.ctypes.data, seq1.shape[1], n, CORRECT_ACTIONS[mismatch_action],
                                       min_qual_difference, changed.ctypes.data, newlen.ctypes.data, nthreads)
    if rc != 0:
        raise MemoryError
    return changed, newlen


def quality_trim_index(qualities, cutoff_front, cutoff_back, base=33):
    """quality_trim_index(qualities, cutoff_front, cutoff_back, base) of _qualtrim.pyx:7-50 -> (start, stop)."""
    q = _b(qualities)
    a, b = C.c_int(), C.c_int()
    lib().orc_quality_trim_index(q, len(q), int(cutoff_front), int(cutoff_back), int(base), C.byref(a), C.byref(b))
    return a.value, b.value


def nextseq_trim_index(sequence, qualities, cutoff, base=33):
    """nextseq_trim_index(read, cutoff, base) of _qualtrim.pyx:53-84 on the read's two strings -> stop."""
    s, q = _b(sequence), _b(qualities)
    assert len(s) == len(q)
    return int(lib().orc_nextseq_trim_index(s, q, len(q), int(cutoff), int(base)))


def n_end_trim(sequence):
    """NEndTrimmer.__call__ (modifiers.py:776-784) -> (start_cut, end_cut) as passed to subseq."""
    s = _b(sequence)
    a, b = C.c_int(), C.c_int()
    lib().orc_n_end_trim(s, len(s), C.byref(a), C.byref(b))
    return a.value, b.value
