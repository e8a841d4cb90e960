# coding: utf-8
"""Alignment module: drop-in for ``atropos.align`` (reference
atropos/align/__init__.py + atropos/align/_align.pyx) whose arithmetic runs in
hand-written HIP kernels on an MI355X.

Same names, argument meaning, return tuples and error behaviour as the reference:
``Aligner``, ``MultiAligner``, ``compare_prefixes``, ``locate``,
``compare_suffixes``, ``Match``, ``MatchInfo``, ``InsertAligner`` and the flag
constants.  Every per-read call is a batch of one on the GPU (there is no CPU
implementation behind this module); the batched twins (``Aligner.locate_batch`` ...)
are the ones to use for throughput.
"""
import ctypes as _C
from collections import namedtuple

import numpy as np
import torch

from .. import _lib
from ..batch import ReadBatch, LocateResult, _as_ascii_matrix
from ..util import RandomMatchProbability, reverse_complement, rmp_table, BASE_COMPLEMENTS

# flags for global alignment (reference align/__init__.py:17-26)
START_WITHIN_SEQ1 = 1
START_WITHIN_SEQ2 = 2
STOP_WITHIN_SEQ1 = 4
STOP_WITHIN_SEQ2 = 8
SEMIGLOBAL = START_WITHIN_SEQ1 | START_WITHIN_SEQ2 | STOP_WITHIN_SEQ1 | STOP_WITHIN_SEQ2


class Aligner(object):
    """Locate one string within another by an optimal semiglobal alignment with unit
    mismatch cost (reference: cdef class Aligner, _align.pyx:121-494).

    ``Aligner(reference, max_error_rate, flags=SEMIGLOBAL, wildcard_ref=False,
    wildcard_query=False, min_overlap=1, indel_cost=1).locate(query)`` returns
    ``(refstart, refstop, querystart, querystop, matches, errors)`` or ``None``.

    Optimality criteria, in order: error rate at most ``max_error_rate``; most
    matches; fewest errors; leftmost in the query.
    """

    def __init__(self, reference, max_error_rate, flags=SEMIGLOBAL, wildcard_ref=False, wildcard_query=False,
                 min_overlap=1, indel_cost=1):
        if not isinstance(reference, str):
            raise TypeError("reference must be str")
        self._backend = _lib.get_backend()
        self._handle = None
        self.max_error_rate = float(max_error_rate)
        self.flags = int(flags)
        self.wildcard_ref = bool(wildcard_ref)
        self.wildcard_query = bool(wildcard_query)
        if min_overlap < 1:
            raise ValueError('Minimum overlap must be at least 1')            # _align.pyx:219-220
        if indel_cost < 1:
            raise ValueError('Insertion/deletion cost must be at least 1')    # _align.pyx:229-230
        self._min_overlap = int(min_overlap)
        self._indel_cost = int(indel_cost)
        self._debug, self._dpmatrix = False, None
        self._set_reference(reference)

    # -- construction / destruction -------------------------------------------------
    def _set_reference(self, reference):
        ref_bytes = reference.encode('ascii')                                 # _align.pyx:243
        self._release()
        self.str_reference = reference
        self._ref_bytes = ref_bytes
        self._long = False
        if len(ref_bytes) == 0:
            # An empty reference can never produce a non-empty alignment (the reference
            # implementation returns None for every query); no device state is needed.
            self._table_kind, self._table = _lib.TABLE_DNA15, None
            return
        be = self._backend
        self._long = len(ref_bytes) > _lib.MAX_REF_LEN
        if self._long:
            # A reference of 129 .. 320 bases has no aligner handle (the column kernels hold the rows in
            # registers): locate_batch runs the per-pair aligner with this reference on every pair
            # (PairAligner, register strips of 128 rows) -- the same Aligner.locate, batch-of-n.
            # (beyond 320 bases: the per-pair aligner's long path, 64-bit cells with the column in a workspace)
            if len(ref_bytes) > _lib.MAX_LONG_READ_LEN:
                raise _lib.AtroposHipError("Aligner: references longer than %d bases are outside the device envelope"
                                           % _lib.MAX_LONG_READ_LEN)
            self._table_kind = self._pair_aligner()._table_kinds()[1]
            self._table = be.translate_table(self._table_kind)
            return
        self._handle = be.aligner_create(ref_bytes, self.max_error_rate, self.flags, self.wildcard_ref,
                                         self.wildcard_query, self._min_overlap, self._indel_cost)
        self._table_kind, self._table = be.aligner_query_table(self._handle)

    def _pair_aligner(self):
        return PairAligner(self.max_error_rate, self.flags, self.wildcard_ref, self.wildcard_query,
                           self._min_overlap, self._indel_cost)

    LONG_CHUNK = 1 << 20                              # reads per PairAligner call of a long reference

    def _locate_long(self, reads):
        """locate_batch for a reference of more than MAX_REF_LEN bases: the reference is laid out once per
        read of a chunk and the chunk goes through the per-pair aligner."""
        be = self._backend
        pa = self._pair_aligner()
        ref_row = torch.frombuffer(bytearray(self._ref_bytes), dtype=torch.uint8).to(be.device)
        step = min(self.LONG_CHUNK, max(64, (1 << 28) // len(self._ref_bytes)))     # (the reference is laid out per read)
        if isinstance(reads, ReadBatch):
            # a packed batch (the device-resident pipelines' adapter stage): whole tiles of 64 reads per chunk
            self._check_batch(reads)
            self._need_tile64(reads, "locate_batch with a reference of more than %d bases" % _lib.MAX_REF_LEN)
            n, nch = reads.nreads, (reads.max_len + 31) // 32
            step = max(64, step // 64 * 64)
            chunks = (ReadBatch(reads.packed[(lo // 64) * nch * 1024:], None if reads.lens is None else
                                reads.lens[lo:min(n, lo + step)].contiguous(), min(n, lo + step) - lo, reads.max_len,
                                reads.table_kind, reads.table) for lo in range(0, n, step))
        elif isinstance(reads, (list, tuple)):
            n = len(reads)
            chunks = ((reads[i:i + step]) for i in range(0, n, step))
        else:
            n = reads.shape[0]
            chunks = ((reads[i:i + step]) for i in range(0, n, step))
        recs = []
        for chunk in chunks:
            count = len(chunk) if isinstance(chunk, (list, tuple, ReadBatch)) else chunk.shape[0]
            refs = ref_row[None, :].expand(count, -1).contiguous()
            recs.append(pa.locate_batch(refs, chunk, unknown_queries_ok=True).records)
        if not recs:
            rec = be.empty((0, 8), torch.int16)
            return LocateResult(rec)
        return LocateResult(recs[0] if len(recs) == 1 else torch.cat(recs, 0))

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            self._backend.aligner_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __reduce__(self):                                                     # _align.pyx:210-212
        return (Aligner, (self.str_reference, self.max_error_rate, self.flags, self.wildcard_ref,
                          self.wildcard_query, self._min_overlap, self._indel_cost))

    # -- properties (_align.pyx:214-249) --------------------------------------------
    @property
    def min_overlap(self):
        return self._min_overlap

    @min_overlap.setter
    def min_overlap(self, value):
        if value < 1:
            raise ValueError('Minimum overlap must be at least 1')
        self._min_overlap = int(value)
        if self._handle is not None:
            self._backend.aligner_set_min_overlap(self._handle, self._min_overlap)

    def _set_indel_cost(self, value):
        if value < 1:
            raise ValueError('Insertion/deletion cost must be at least 1')
        self._indel_cost = int(value)
        if self._handle is not None:
            self._backend.aligner_set_indel_cost(self._handle, self._indel_cost)

    indel_cost = property(None, _set_indel_cost, doc="Matches cost 0, mismatches cost 1; only the "
                          "insertion/deletion cost can be changed (write-only, as in the reference).")

    @property
    def reference(self):
        """The translated reference bytes (reference: property reference, :234-249)."""
        if self.wildcard_ref:
            return self._ref_bytes.translate(self._backend.translate_table(_lib.TABLE_IUPAC))
        if self.wildcard_query:
            return self._ref_bytes.translate(self._backend.translate_table(_lib.TABLE_ACGT))
        return self._ref_bytes

    @reference.setter
    def reference(self, reference):
        self._set_reference(reference)

    @property
    def dpmatrix(self):
        """The dynamic programming matrix of the last ``locate`` call as a :class:`DPMatrix`, once
        ``enable_debug()`` has been called (reference: _align.pyx:250-257)."""
        return self._dpmatrix

    def enable_debug(self):
        """Store the DP matrix while running ``locate`` and make it available as ``.dpmatrix``
        (reference: _align.pyx:259-264).  A debugging aid: the matrix of one pair comes from a
        one-lane kernel that walks the reference's own loop (``atr_locate_debug``)."""
        self._debug = True

    # -- packing ----------------------------------------------------------------------
    @property
    def table_kind(self):
        """Which translate table (``_lib.TABLE_*``) query batches must be packed with."""
        return self._table_kind

    def pack(self, reads, layout="tile64"):
        """Pack reads for this aligner.  ``reads``: a ReadBatch (returned unchanged if
        compatible), a sequence of str/bytes, or a uint8 [nreads, width] array/tensor.
        ``layout``: "tile64" (4-bit codes: what every entry point reads, whatever the batch size), "plane64" (bit
        planes of the codes: what the two-pass pre-pass of ``locate_batch`` reads) or "auto" -- what ``locate_batch``
        picks when it packs the reads itself: plane64 for a long batch when that pre-pass takes this aligner, else
        tile64.  (A plane64 batch serves ``locate_batch`` with its default path only.)"""
        if isinstance(reads, ReadBatch):
            self._check_batch(reads)
            return reads
        kind = self._table_kind
        table = self._table if self._table is not None else self._backend.translate_table(kind)
        if isinstance(reads, (list, tuple)):
            widths = set(map(len, reads))
            planes = self._wants_planes(layout, len(reads), max(widths) if widths else 0, ragged=len(widths) > 1)
            return ReadBatch.from_strings(reads, kind, table, self._backend, planes=planes)
        planes = self._wants_planes(layout, reads.shape[0], reads.shape[1])
        return ReadBatch.from_ascii(reads, None, None, kind, table, self._backend, planes=planes)

    def prepare(self, max_len, ragged=False):
        """Have the library build (or load from its code-object cache) the pre-pass kernel specialised for this
        adapter and read length (``atr_aligner_prepare``): worth it for an aligner that will see millions of reads
        -- ``locate_batch`` does it by itself once the aligner has seen 2 M reads.  Returns whether such a kernel is
        in place; the records are the same either way."""
        be = self._backend
        if self._handle is None or not hasattr(be, "aligner_prepare"):
            return False
        return be.aligner_prepare(self._handle, max_len, ragged)

    def _wants_planes(self, layout, nreads, width, ragged=False):
        if layout == "tile64" or self._handle is None or getattr(self, "_long", False) or width > _lib.MAX_READ_LEN:
            return False
        be = self._backend
        ok = hasattr(be, "locate_planes_applies") and width > 0 and be.locate_planes_applies(self._handle, width, ragged)
        if layout == "plane64":
            if not ok:
                raise _lib.AtroposUnsupported("the plane64 layout (two-pass pre-pass) does not take this aligner / read length")
            return True
        return ok and nreads >= _lib.PLANES_MIN_READS

    def _check_batch(self, batch):
        if self._handle is None:
            return
        if batch.table_kind != self._table_kind or (
                self._table_kind == _lib.TABLE_CUSTOM and batch.table != self._table):
            raise ValueError("read batch was packed with translate table %d but this aligner needs %d "
                             "(use aligner.pack(reads))" % (batch.table_kind, self._table_kind))

    @staticmethod
    def _need_tile64(batch, what):
        if batch.layout != "tile64":
            raise ValueError("%s reads the tile64 layout; this batch is %s (aligner.pack(reads, layout=\"tile64\"))"
                             % (what, batch.layout))

    # -- alignment ----------------------------------------------------------------------
    def locate_stream(self, batches, depth=2):
        """``locate_batch`` over a sequence of batches with the calls issued on ``depth`` streams in turn (a workspace
        each): the call of batch i + 1 -- its pre-pass -- runs while the exact DP of batch i is still finishing, which a
        single stream cannot do (the two phases of ONE call are both issue-bound and serial).  A generator: the result
        of batch i is yielded once batch i + depth - 1 has been issued; the caller's current stream is made to wait
        for it (``wait_event``), so device work queued afterwards sees complete records -- host reads need the usual
        synchronisation.  Same records as ``locate_batch``, batch by batch.  (Long batches: with short ones the calls
        are latency-bound and there is nothing to overlap.)"""
        import collections
        be = self._backend
        if not hasattr(be, "side_streams"):                      # (the CPU test double: one call after the other)
            for b in batches:
                yield self.locate_batch(b)
            return
        streams = be.side_streams(max(1, int(depth)))
        caller = torch.cuda.current_stream(be.device)
        pending = collections.deque()

        def hand_over():
            res, ev, keep = pending.popleft()
            caller.wait_event(ev)
            return res

        for i, b in enumerate(batches):
            st = streams[i % len(streams)]
            batch = self.pack(b, layout="auto") if not isinstance(b, ReadBatch) else b
            st.wait_stream(caller)                                   # (the batch was produced on the caller's stream)
            with torch.cuda.stream(st):
                res = self.locate_batch(batch)
                ev = torch.cuda.Event()
                ev.record(st)
            for t in (batch.packed, batch.lens):
                if t is not None:
                    t.record_stream(st)
            res.records.record_stream(caller)
            pending.append((res, ev, batch))
            if len(pending) >= len(streams):
                yield hand_over()
        while pending:
            yield hand_over()

    def locate_batch(self, reads, filtered=True, path=None):
        """Batched ``locate``: one result record per read (see LocateResult).  The library picks
        its kernels by batch size (a wavefront per read for short batches, the filtered pipeline --
        bit-parallel pre-pass + windowed DP -- for long ones); ``path`` ("full", "filtered", "wave",
        ``_lib.LOCATE_PATHS``) names one, ``filtered=False`` is "full"; the records are identical
        on every path."""
        if getattr(self, "_long", False):
            return self._locate_long(reads)
        be = self._backend
        if isinstance(reads, (list, tuple)) and len(reads) > 1 and path != "wave":
            # a few reads beyond the batch pipelines' length (the reference has no limit, _align.pyx:266-291): they go
            # through the long-read sweep as a batch of their own, so that the others are not padded to their length
            lens = np.fromiter(map(len, reads), dtype=np.int64, count=len(reads))
            if int(lens.max()) > _lib.MAX_READ_LEN and int(lens.min()) <= _lib.MAX_READ_LEN:
                long_idx = np.nonzero(lens > _lib.MAX_READ_LEN)[0]
                short_idx = np.nonzero(lens <= _lib.MAX_READ_LEN)[0]
                rec = be.empty((len(reads), 8), torch.int16)
                for idx in (short_idx, long_idx):
                    part = self.locate_batch([reads[i] for i in idx.tolist()], filtered, path).records
                    rec[torch.from_numpy(idx).to(rec.device)] = part
                return LocateResult(rec)
        if (self._handle is not None and path in (None, "auto", "wave") and filtered and isinstance(reads, (list, tuple))
                and 0 < len(reads) <= _lib.WAVE_MAX_READS and hasattr(be, "locate_ascii_batch")):
            # a short list of strings: the rows go to the device as they are and the wavefront-per-read kernel
            # translates them itself (no pack kernel in between)
            mat, lens = _as_ascii_matrix(reads)
            width = (mat.shape[1] + 3) // 4 * 4
            if width <= _lib.MAX_READ_LEN + 3 and int(lens.max()) <= _lib.MAX_READ_LEN:
                if width != mat.shape[1]:
                    mat = np.pad(mat, ((0, 0), (0, width - mat.shape[1])))
                lens_t = None if int(lens.min()) == int(lens.max()) else torch.from_numpy(lens).to(be.device)
                return LocateResult(be.locate_ascii_batch(self._handle, torch.from_numpy(mat).to(be.device), lens_t,
                                                          int(lens.max())))
        if (self._handle is not None and filtered and path in (None, "auto") and torch.is_tensor(reads) and reads.dim() == 2
                and reads.dtype == torch.uint8 and reads.is_cuda and reads.shape[0] > _lib.WAVE_MAX_READS
                and hasattr(be, "locate_ascii_planes_batch") and reads.stride(1) == 1
                and self._wants_planes("auto", reads.shape[0], reads.shape[1])):
            # a long batch of ASCII rows on the device: packed and pre-passed by ONE kernel (atr_locate_ascii_planes_batch)
            return self.locate_ascii(reads)[0]
        batch = self.pack(reads, layout="plane64" if path == "pieces" else
                          "auto" if (filtered and path in (None, "auto")) else "tile64")
        if self._handle is None:                      # empty reference: nothing ever matches
            rec = be.empty((batch.nreads, 8), torch.int16)
            rec.zero_()
            rec[:, 1] = -1
            return LocateResult(rec)
        if batch.layout == "plane64":
            if path not in (None, "auto", "pieces") or not filtered:
                self._need_tile64(batch, "locate_batch(path=%r)" % (path if filtered else "full"))
            return LocateResult(be.locate_planes_batch(self._handle, batch.packed, batch.lens, batch.nreads, batch.max_len))
        if path == "pieces":
            raise ValueError("path \"pieces\" (the two-pass pre-pass) reads the plane64 layout: aligner.pack(reads, layout=\"plane64\")")
        return LocateResult(be.locate_batch(self._handle, batch.packed, batch.lens, batch.nreads, batch.max_len,
                                            filtered, path))

    def locate_ascii(self, reads, lens=None, max_len=None):
        """``locate`` for a long batch that is still text: ``reads`` a uint8 [n, width] tensor of ASCII rows on the device,
        ``lens`` int32 [n] or None.  One kernel packs every read into bit planes and runs the two-pass pre-pass on the
        registers it packed into (``atr_locate_ascii_planes_batch``; the aligner must be one ``pack(layout="auto")`` would
        choose bit planes for).  Returns (LocateResult, the packed ``ReadBatch`` the call left behind -- e.g. for the next
        adapter of a ``times > 1`` round); the records equal ``locate_batch(pack(reads))``."""
        be = self._backend
        max_len = reads.shape[1] if max_len is None else int(max_len)
        rec, planes = be.locate_ascii_planes_batch(self._handle, reads, lens, max_len)
        return LocateResult(rec), ReadBatch(planes, lens, reads.shape[0], max_len, self.table_kind, self._table, layout="plane64")

    def compare_batch(self, reads, suffix=False):
        """``compare_prefixes(reference, read, wildcard_ref, wildcard_query)`` (``compare_suffixes`` with
        ``suffix``) for a batch packed for this aligner -- what Adapter.match_to runs instead of
        ``locate`` for anchored adapters without indels (reference: adapters/__init__.py:370-380,
        _align.pyx:501-544).  Returns the int16 [n, 8] record tensor on the device."""
        if getattr(self, "_long", False):
            raise _lib.AtroposHipError("compare_batch: references longer than %d bases go through "
                                       "align.compare_batch(ref, queries, ...)" % _lib.MAX_REF_LEN)
        batch = self.pack(reads, layout="tile64")
        self._need_tile64(batch, "compare_batch")
        be = self._backend
        if self._handle is None:                      # empty reference: the empty overlap
            rec = be.empty((batch.nreads, 8), torch.int16)
            rec.zero_()
            if suffix:
                lens = batch.lens if batch.lens is not None else torch.full((batch.nreads,), batch.max_len)
                rec[:, 2] = rec[:, 3] = lens.to(rec.device).to(torch.int16)
            return rec
        return be.compare_packed(self._handle, batch.packed, batch.lens, batch.nreads, batch.max_len, suffix)

    def locate(self, query):
        """locate(query) -> (refstart, refstop, querystart, querystop, matches, errors)

        Find the query within the reference associated with this aligner
        (reference: Aligner.locate, _align.pyx:266-491)."""
        if not isinstance(query, str):
            raise TypeError("query must be str")
        qbytes = query.encode('ascii')                 # UnicodeEncodeError like the reference (:281)
        be = self._backend
        if getattr(self, "_long", False) and not self._debug:
            result = self._pair_aligner().locate(self.str_reference, query, unknown_queries_ok=True)
        elif self._handle is not None and hasattr(be, "locate_one") and len(qbytes) <= _lib.MAX_READ_LEN:
            # the per-read API as the module swap uses it: one library call (HipBackend.locate_one)
            result = be.locate_one(self._handle, qbytes)
        else:
            result = self.locate_batch([query]).tuples()[0]
        if self._debug and self._handle is not None:
            batch = self.pack([query])
            costs, rec = self._backend.locate_debug(self._handle, batch.packed, len(self._ref_bytes), len(query))
            assert LocateResult(rec).tuples()[0] == result
            self._dpmatrix = DPMatrix(self.str_reference, query)
            for i, row in enumerate(costs.tolist()):
                for j, cost in enumerate(row):
                    if cost != _DEBUG_NOT_COMPUTED:
                        self._dpmatrix.set_entry(i, j, cost)
        return result


def locate(reference, query, max_error_rate, flags=SEMIGLOBAL, wildcard_ref=False, wildcard_query=False,
           min_overlap=1):
    """Convenience wrapper (reference: _align.pyx:496-499; the indel cost is always 1)."""
    aligner = Aligner(reference, max_error_rate, flags, wildcard_ref, wildcard_query)
    aligner.min_overlap = min_overlap
    return aligner.locate(query)


def _ascii_tensor(strings, backend):
    """list of str -> (uint8 tensor [n, width] on the backend's device, int32 lens tensor)."""
    mat, lens = _as_ascii_matrix(strings)
    return (torch.from_numpy(mat).to(backend.device), torch.from_numpy(lens).to(backend.device))


def compare_prefixes(ref, query, wildcard_ref=False, wildcard_query=False):
    """Find out whether one string is the prefix of the other one, allowing IUPAC
    wildcards in ref and/or query if the appropriate flag is set; returns a tuple
    compatible with ``Aligner.locate`` (reference: _align.pyx:501-544)."""
    return _compare(ref, query, wildcard_ref, wildcard_query, False)


def compare_suffixes(suffix_ref, suffix_query, wildcard_ref=False, wildcard_query=False):
    """Suffix twin of compare_prefixes, used for anchored 3' adapters without indels
    (reference: align/__init__.py:28-44)."""
    return _compare(suffix_ref, suffix_query, wildcard_ref, wildcard_query, True)


def _compare(ref, query, wildcard_ref, wildcard_query, suffix):
    be = _lib.get_backend()
    if hasattr(be, "compare_one") and len(ref) <= _COMPARE_REF_MAX:
        return be.compare_one(ref.encode('ascii'), query.encode('ascii'), wildcard_ref, wildcard_query, suffix)
    rec = compare_batch(ref, [query], wildcard_ref, wildcard_query, suffix).cpu().numpy()[0]
    return tuple(int(v) for v in rec[:6])


def compare_batch(ref, queries, wildcard_ref=False, wildcard_query=False, suffix=False, lens=None):
    """compare_prefixes (or compare_suffixes) of one reference against a batch of
    queries: list of str, or a uint8 [n, width] ASCII tensor (+ optional int32 ``lens``).
    Returns the int16 [n, 8] record tensor on the device."""
    be = _lib.get_backend()
    ref_b = ref.encode('ascii')
    if isinstance(queries, (list, tuple)):
        for q in queries:
            q.encode('ascii')
        q_t, lens = _ascii_tensor(queries, be)
    else:
        q_t = queries.to(be.device)
        lens = None if lens is None else lens.to(device=be.device, dtype=torch.int32)
    if q_t.shape[0] and q_t.shape[1] == 0:
        q_t = torch.zeros((q_t.shape[0], 1), dtype=torch.uint8, device=be.device)
        lens = torch.zeros((q_t.shape[0],), dtype=torch.int32, device=be.device) if lens is None else lens
    if len(ref_b) > _COMPARE_REF_MAX:
        return _compare_long(be, ref_b, q_t, lens, wildcard_ref, wildcard_query, suffix)
    return be.compare_batch(ref_b, q_t, lens, q_t.shape[1], wildcard_ref, wildcard_query, suffix)


_COMPARE_REF_MAX = 1024          # reference bytes one atr_compare_batch call takes (they travel as kernel arguments)


def _compare_long(be, ref_b, q_t, lens, wildcard_ref, wildcard_query, suffix):
    """compare_prefixes / compare_suffixes with a reference of more than 1024 characters (the reference has no limit,
    _align.pyx:501-544): the compared stretch is cut into pieces of 1024 positions -- matches add up -- and a suffix
    compare is the prefix compare of the reversed strings.  Coordinates beyond the records' int16 fields (32767) are
    refused."""
    m, n_rows, width = len(ref_b), q_t.shape[0], q_t.shape[1]
    dev = q_t.device
    if lens is None:
        lens = torch.full((n_rows,), width, dtype=torch.int32, device=dev)
    lens = lens.clamp(min=0, max=width).to(torch.int32)
    if m > 32767 or width > 32767:
        raise _lib.AtroposUnsupported("compare_prefixes: more than 32767 characters (the records' int16 fields)")
    if suffix:
        # reverse every query inside its own length, and the reference
        cols = torch.arange(width, device=dev, dtype=torch.int64)[None, :]
        src = (lens.to(torch.int64)[:, None] - 1 - cols).clamp(min=0)
        q_t = torch.where(cols < lens[:, None], torch.gather(q_t, 1, src), torch.zeros_like(q_t))
        ref_b = ref_b[::-1]
    matches = torch.zeros((n_rows,), dtype=torch.int64, device=dev)
    for c in range(0, min(m, width), _COMPARE_REF_MAX):
        piece = ref_b[c:c + _COMPARE_REF_MAX]
        sub = q_t[:, c:c + len(piece)].contiguous()
        sub_lens = (lens - c).clamp(min=0, max=sub.shape[1]).to(torch.int32)
        rec = be.compare_batch(piece, sub, sub_lens, sub.shape[1], wildcard_ref, wildcard_query, False)
        matches += rec[:, 4].to(torch.int64)
    ln = torch.minimum(lens.to(torch.int64), torch.tensor(m, dtype=torch.int64, device=dev))
    out = torch.zeros((n_rows, 8), dtype=torch.int64, device=dev)
    if suffix:
        out[:, 0] = m - ln; out[:, 1] = m; out[:, 2] = lens.to(torch.int64) - ln; out[:, 3] = lens.to(torch.int64)
    else:
        out[:, 1] = ln; out[:, 3] = ln
    out[:, 4] = matches; out[:, 5] = ln - matches
    return out.to(torch.int16)


def case_sensitive_pair_table(dna15):
    """Translate table for the LITERAL pair compare of reads with soft-masked (lower-case) bases: the
    aligner MergeOverlapping builds compares characters, so ``a`` differs from ``A`` and equals ``a``
    (reference: _align.pyx:390-391), and its reverse complement keeps the case
    (util/__init__.py:67-88).  The 4-bit codes only have to be distinct per character and closed under
    the device's complement (nibble bit-reversal): upper-case A C G T N W B D H V keep their DNA15
    codes, a / t take 3 / 12, c / g take 5 / 10, n takes 6 -- the codes of M K R Y S, which this table
    therefore cannot hold (0: no code)."""
    table = bytearray(dna15)
    for ch in b"MKRYS":
        table[ch] = 0
    for ch, code in zip(b"atcgn", (3, 12, 5, 10, 6)):
        table[ch] = code
    return bytes(table)


_DEBUG_NOT_COMPUTED = -(1 << 31)


class DPMatrix(object):
    """The dynamic programming matrix of one ``Aligner.locate`` call, for debugging: one row per
    reference position, one column per query position; an entry is None where the aligner never
    computed a value (reference: class DPMatrix, _align.pyx:88-119)."""

    def __init__(self, reference, query):
        self.reference, self.query = reference, query
        self._rows = [[None] * (len(query) + 1) for _ in range(len(reference) + 1)]

    def set_entry(self, i, j, cost):
        self._rows[i][j] = cost

    def __str__(self):
        lines = [" " * 5 + " ".join("%2s" % base for base in self.query)]
        for base, row in zip(" " + self.reference, self._rows):
            lines.append(base + " " + " ".join("  " if cost is None else "%2d" % cost for cost in row))
        return "\n".join(lines)


LONG_PAIRS_WORK_BYTES = 1 << 29                     # workspace of one call of the long-pair kernel (the DP columns)


def long_pairs_records(be, rpacked, rlens, rmax, revcomp_ref, qpacked, qlens, qmax, n, max_error_rate, flags,
                       wildcard_ref, wildcard_query, min_overlap, indel_cost):
    """atr_locate_pairs_long_batch (64-bit cells, the DP column in a workspace) over tile64-packed sides, whole tiles
    of 64 pairs at a time so that the workspace stays bounded.  int16 [n, 8] records."""
    per_pair = 9 * (rmax + 1)
    step = max(64, (LONG_PAIRS_WORK_BYTES // per_pair) // 64 * 64)
    rch, qch = (rmax + 31) // 32, (qmax + 31) // 32
    out = []
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rp = rpacked[(lo // 64) * rch * 1024:]                  # a tile of 64 pairs: nchunks x 64 x 16 bytes
        qp = qpacked[(lo // 64) * qch * 1024:]
        out.append(be.locate_pairs_long_batch(
            rp, None if rlens is None else rlens[lo:hi].contiguous(), rmax, revcomp_ref,
            qp, None if qlens is None else qlens[lo:hi].contiguous(), qmax, hi - lo,
            max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost))
    if not out:
        return be.empty((0, 8), torch.int16)
    return out[0] if len(out) == 1 else torch.cat(out, 0)


class PairAligner(object):
    """``Aligner(ref_p, max_error_rate, flags, ...).locate(query_p)`` for many independent
    (reference, query) pairs in one GPU call -- the aligner ``MergeOverlapping`` constructs per
    read pair (commands/trim/modifiers.py:889-894), where the reference changes with every
    call.  ``revcomp_ref``: the reference of a pair is the reverse complement of the given
    sequence (``reverse_complement(read2.sequence)``), formed on the device.
    Sequences of up to 255 bases; results are the reference's 6-tuples / None."""

    def __init__(self, max_error_rate, flags=SEMIGLOBAL, wildcard_ref=False, wildcard_query=False, min_overlap=1,
                 indel_cost=1, revcomp_ref=False):
        if min_overlap < 1:
            raise ValueError("Minimum overlap must be at least 1")          # _align.pyx:219-220
        if indel_cost < 1:
            raise ValueError("Insertion/deletion cost must be at leat 1")   # :229-230
        self.max_error_rate, self.flags = float(max_error_rate), int(flags)
        self.wildcard_ref, self.wildcard_query = bool(wildcard_ref), bool(wildcard_query)
        self.min_overlap, self.indel_cost, self.revcomp_ref = int(min_overlap), int(indel_cost), bool(revcomp_ref)

    def _table_kinds(self):
        """Translate tables of (reference, query): _align.pyx:243-248, :292-297."""
        if not (self.wildcard_ref or self.wildcard_query):
            return _lib.TABLE_DNA15, _lib.TABLE_DNA15
        return (_lib.TABLE_IUPAC if self.wildcard_ref else _lib.TABLE_ACGT,
                _lib.TABLE_IUPAC if self.wildcard_query else _lib.TABLE_ACGT)

    def _pack(self, seqs, kind, be, literal, case_table=None, may_retry=False, unknown_ok=False):
        if isinstance(seqs, ReadBatch):
            if seqs.table_kind != kind:
                raise ValueError("batch packed with table %d, this side needs %d" % (seqs.table_kind, kind))
            return seqs
        table = be.translate_table(kind)
        if isinstance(seqs, (list, tuple)):
            mat, lens = _as_ascii_matrix(seqs)
            ascii_t, lens_t = torch.from_numpy(mat).to(be.device), torch.from_numpy(lens).to(be.device)
            max_len = int(lens.max()) if len(seqs) else 0
        else:
            ascii_t = seqs.to(be.device)
            lens_t, max_len = None, ascii_t.shape[1]
        if max_len > _lib.MAX_LONG_READ_LEN:
            raise _lib.AtroposHipError("PairAligner: sequences longer than %d bases are outside the device envelope"
                                       % _lib.MAX_LONG_READ_LEN)
        if literal or self.revcomp_ref:
            # the literal compare works on 4-bit codes: every base needs one (and a complement)
            if case_table is not None:
                table = case_table
            packed, bad = be.pack_reads(ascii_t, lens_t, max_len, table, count_invalid=True)
            if bad and case_table is None and literal and may_retry:
                return None                               # the caller packs both sides again, case-sensitively
            if bad and (literal or kind == _lib.TABLE_DNA15) and not (unknown_ok and literal and not self.revcomp_ref):
                raise ValueError("%d sequence(s) contain characters the device pair aligner has no 4-bit code for "
                                 "(upper-case IUPAC letters, or A C G T N W B D H V in either case)" % bad)
        else:
            packed = be.pack_reads(ascii_t, lens_t, max_len, table)
        return ReadBatch(packed, lens_t, ascii_t.shape[0], max_len, kind, table)

    def locate_batch(self, references, queries, need=None, path="auto", unknown_queries_ok=False):
        """references/queries: lists of str, uint8 [n, width] ASCII tensors, or ReadBatches
        packed with the right tables.  Returns a LocateResult (int16 [n, 8] records).
        path: the kernel family (``_lib.PAIRS_PATHS``; "auto" picks by batch size: a wavefront per pair for short
        batches, the cost / threat / band pipeline or the full sweep for long ones).
        unknown_queries_ok: a query character without a 4-bit code (anything but IUPAC letters) takes code 0 and
        matches nothing instead of raising -- exact for the literal compare as long as the references hold coded
        characters only (they are still checked): what ``Aligner.locate`` does with such reads.
        need: per pair (list or int32 tensor), the number of matches below which the caller ignores the
        alignment (MergeOverlapping's ``matches >= min_overlap``, modifiers.py:896-897): such pairs may come
        back as None, which lets the library stop after its cost pass for pairs that cannot overlap that far."""
        be = _lib.get_backend()
        rk, qk = self._table_kinds()
        literal = not (self.wildcard_ref or self.wildcard_query)
        raw = not isinstance(references, ReadBatch) and not isinstance(queries, ReadBatch)
        rb = self._pack(references, rk, be, literal, may_retry=raw)
        qb = self._pack(queries, qk, be, literal, may_retry=raw) if rb is not None else None
        if rb is None or qb is None:
            # soft-masked reads in the literal compare: BOTH sides with the table that tells the cases apart (the
            # two tables share codes, so one side alone must never switch)
            both = case_sensitive_pair_table(be.translate_table(_lib.TABLE_DNA15))
            rb = self._pack(references, rk, be, literal, case_table=both)
            qb = self._pack(queries, qk, be, literal, case_table=both, unknown_ok=unknown_queries_ok)
        if rb.nreads != qb.nreads:
            raise ValueError("need as many references as queries")
        if need is not None and not torch.is_tensor(need):
            need = torch.tensor(list(need), dtype=torch.int32)
        if need is not None:
            need = need.to(device=rb.packed.device, dtype=torch.int32).contiguous()
            if need.numel() != rb.nreads:
                raise ValueError("need: one entry per pair")
        longest = max(rb.max_len, qb.max_len)
        if longest > _lib.PAIRS_MAX_LEN or (longest > 255 and not (self.flags & STOP_WITHIN_SEQ2)):
            return LocateResult(self._locate_long_pairs(be, rb, qb))
        rec = be.locate_pairs_batch(rb.packed, rb.lens, rb.max_len, self.revcomp_ref, qb.packed, qb.lens, qb.max_len,
                                    rb.nreads, self.max_error_rate, self.flags, self.wildcard_ref, self.wildcard_query,
                                    self.min_overlap, self.indel_cost, need=need, path=path)
        return LocateResult(rec)

    def _locate_long_pairs(self, be, rb, qb):
        """A side beyond PAIRS_MAX_LEN (the reference has no length limit, _align.pyx:266-291); `need` does not apply
        (every pair gets the reference's record)."""
        return long_pairs_records(be, rb.packed, rb.lens, rb.max_len, self.revcomp_ref, qb.packed, qb.lens, qb.max_len,
                                  rb.nreads, self.max_error_rate, self.flags, self.wildcard_ref, self.wildcard_query,
                                  self.min_overlap, self.indel_cost)

    def locate(self, reference, query, unknown_queries_ok=False):
        be = _lib.get_backend()
        if hasattr(be, "locate_pair_one") and len(reference) <= 319 and len(query) <= _lib.PAIRS_MAX_LEN:
            # one pair: translate here (bytes.translate, as the reference does per read), one library call
            rk, qk = self._table_kinds()
            literal = not (self.wildcard_ref or self.wildcard_query)
            rc = reference.encode('ascii').translate(be.translate_table(rk))
            qc = query.encode('ascii').translate(be.translate_table(qk))
            checked = literal or self.revcomp_ref
            clean = not checked or (b"\0" not in rc and (b"\0" not in qc or (unknown_queries_ok and literal and not self.revcomp_ref)))
            if clean and (self.flags & STOP_WITHIN_SEQ2 or max(len(rc), len(qc)) <= 255):
                return be.locate_pair_one(rc, self.revcomp_ref, qc, self.max_error_rate, self.flags, self.wildcard_ref,
                                          self.wildcard_query, self.min_overlap, self.indel_cost)
            # (soft-masked or uncoded characters: the batch path sorts those out)
        return self.locate_batch([reference], [query], unknown_queries_ok=unknown_queries_ok).tuples()[0]


class MultiAligner(object):
    """Same as Aligner, but 1) returns up to ``max_matches`` matches rather than a single
    best match, and 2) does not allow indels or wildcards (reference: cdef class
    MultiAligner, _align.pyx:548-787)."""

    def __init__(self, max_error_rate, flags=SEMIGLOBAL, min_overlap=1):
        self.max_error_rate = float(max_error_rate)
        self.flags = int(flags)
        self._min_overlap = int(min_overlap)

    def __reduce__(self):
        return (MultiAligner, (self.max_error_rate, self.flags, self._min_overlap))

    def locate_batch(self, references, queries, max_matches=100):
        """One (reference, query) pair per entry; returns a list with, per pair, None or
        the list of 6-tuples ``locate`` returns."""
        be = _lib.get_backend()
        if len(references) != len(queries):
            raise ValueError("need as many references as queries")
        if len(references) == 0:
            return []
        for s_ in references:
            s_.encode('ascii')
        for s_ in queries:
            s_.encode('ascii')
        r_t, r_l = _ascii_tensor(references, be)
        q_t, q_l = _ascii_tensor(queries, be)
        max_m = int(r_l.max().item())
        stride = max_matches + max_m + 2          # the last-column scan appends past max_matches (:750-763)
        out, counts = be.multi_locate_batch(r_t, r_l, q_t, q_l, self.max_error_rate, self.flags, self._min_overlap,
                                            max_matches, max_m, stride)
        out = out.cpu().numpy()
        counts = counts.cpu().numpy()
        res = []
        for p in range(len(references)):
            c = int(counts[p])
            res.append(None if c == 0 else [tuple(int(v) for v in out[p, t, :6]) for t in range(c)])
        return res

    def locate(self, reference, query, max_matches=100):
        """locate(reference, query) -> list of (refstart, refstop, querystart, querystop,
        matches, errors), or None."""
        be = _lib.get_backend()
        if hasattr(be, "multi_locate_one") and len(reference) <= 20000 and len(query) <= 32000:
            return be.multi_locate_one(reference.encode('ascii'), query.encode('ascii'), self.max_error_rate, self.flags,
                                       self._min_overlap, max_matches)
        return self.locate_batch([reference], [query], max_matches)[0]


# ---- the match record the boundary objects pass around (reference align/__init__.py:51-175) ----

MatchInfo = namedtuple("MatchInfo", "read_name errors rstart rstop seq_before seq_adapter seq_after adapter_name "
                                    "qual_before qual_adapter qual_after is_front asize rsize_adapter rsize_total")

_RECORD_FIELDS = ("astart", "astop", "rstart", "rstop", "matches", "errors")


class Match(object):
    """One adapter match: the six numbers of a result record (interval in the adapter, interval in
    the read, matching and erroneous positions) plus its context -- ``front`` (does the match
    remove the read's 5' end?  guessed from ``rstart == 0`` when not given), the ``adapter`` and
    the ``read`` -- and ``length``, the adapter bases covered.  ``ValueError`` for an empty match or
    one without a single matching position, as in the reference (:85-88)."""

    __slots__ = _RECORD_FIELDS + ("front", "adapter", "read", "length")

    def __init__(self, astart, astop, rstart, rstop, matches, errors, front=None, adapter=None, read=None):
        covered = astop - astart
        if covered <= 0:
            raise ValueError("Match length must be >= 0")
        if errors >= covered:
            raise ValueError("A Match requires at least one matching position.")
        for slot, value in zip(_RECORD_FIELDS, (astart, astop, rstart, rstop, matches, errors)):
            setattr(self, slot, value)
        self.length = covered
        self.front = (rstart == 0) if front is None else front
        self.adapter, self.read = adapter, read

    def record(self):
        return tuple(getattr(self, slot) for slot in _RECORD_FIELDS)

    def __repr__(self):
        return "Match(%s)" % ", ".join("%s=%d" % item for item in zip(_RECORD_FIELDS, self.record()))

    def copy(self):
        return Match(*self.record(), front=self.front, adapter=self.adapter, read=self.read)

    def wildcards(self, wildcard_char='N'):
        """The read characters that sit opposite the adapter's ``wildcard_char`` positions (not
        reliable with indels: the alignment itself is not kept)."""
        adapter, bases = self.adapter.sequence, self.read.sequence
        pairs = zip(adapter[self.astart:self.astop], bases[self.rstart:])
        return "".join(base for letter, base in pairs if letter == wildcard_char)

    def rest(self):
        """What a front adapter leaves before it, any other adapter after it."""
        bases = self.read.sequence
        return bases[:self.rstart] if self.front else bases[self.rstop:]

    def get_info_record(self):
        """The row of the info file for this match (reference :143-170)."""
        bases, quals = self.read.sequence, self.read.qualities or ""
        lo, hi = self.rstart, self.rstop
        removed = hi - lo
        if self.front:
            removed = hi if lo > 0 else removed               # everything up to the adapter's end goes
        elif hi < len(bases):
            removed = len(bases) - lo                         # everything from the adapter's start goes
        split = lambda text: (text[:lo], text[lo:hi], text[hi:])
        return MatchInfo(self.read.name, self.errors, lo, hi, *split(bases), self.adapter.name, *split(quals),
                         self.front, self.length, hi - lo, removed)


class InsertResult(object):
    """Result records of a batched ``match_insert``: int16 tensor [npairs, 3, 8] on the
    device -- per pair the insert match tuple, Match 1 and Match 2 (see
    include/atropos_hip.h, atr_insert_match_batch)."""

    def __init__(self, records):
        self.records = records

    def __len__(self):
        return self.records.shape[0]

    def numpy(self):
        return self.records.cpu().numpy()

    def found(self):
        return self.records[:, 0, 1] >= 0

    def results(self):
        """What per-pair ``match_insert`` calls return: None or (insert_match_tuple,
        Match | None, Match | None)."""
        out = []
        for rec in self.numpy():
            if rec[0, 1] < 0:
                out.append(None)
                continue
            ins = tuple(int(v) for v in rec[0, :6])
            ms = [None if rec[t, 1] < 0 else Match(*(int(v) for v in rec[t, :6])) for t in (1, 2)]
            out.append((ins, ms[0], ms[1]))
        return out


class InsertAligner(object):
    """Insert matching: align read 1 to the reverse complement of read 2; if the inserts
    overlap, look for the adapters in the overhangs (reference: InsertAligner,
    align/__init__.py:178-377).  Only works with paired-end reads with 3' adapters.

    Args (same as the reference): adapter1, adapter2, match_probability
    (``callable(matches, size, **base_probs)``), insert_max_rmp, adapter_max_rmp,
    min_insert_overlap, max_insert_mismatch_frac, min_adapter_overlap,
    max_adapter_mismatch_frac, adapter_check_cutoff, base_probs, adapter_wildcards,
    read_wildcards.

    Device envelope: adapters up to 64 bases, reads up to 256 bases, read 2 upper-case
    (every base needs a complement anyway -- ``KeyError`` as in the reference; lower-case
    complements are the one thing the 4-bit device alphabet does not carry).
    """

    def __init__(self, adapter1, adapter2, match_probability=None, insert_max_rmp=1E-6, adapter_max_rmp=0.001,
                 min_insert_overlap=1, max_insert_mismatch_frac=0.2, min_adapter_overlap=1,
                 max_adapter_mismatch_frac=0.2, adapter_check_cutoff=9, base_probs=None, adapter_wildcards=True,
                 read_wildcards=False):
        self._backend, self._handle = _lib.get_backend(), None
        self.adapter1, self.adapter2 = adapter1, adapter2
        self.adapter1_len, self.adapter2_len = len(adapter1), len(adapter2)
        self.match_probability = _default_rmp() if match_probability is None else match_probability
        self.base_probs = base_probs or dict(match_prob=0.25, mismatch_prob=0.75)
        thresholds = dict(insert_max_rmp=insert_max_rmp, adapter_max_rmp=adapter_max_rmp,
                          min_insert_overlap=min_insert_overlap, min_adapter_overlap=min_adapter_overlap,
                          max_insert_mismatch_frac=float(max_insert_mismatch_frac),
                          max_adapter_mismatch_frac=float(max_adapter_mismatch_frac),
                          adapter_check_cutoff=adapter_check_cutoff, adapter_wildcards=adapter_wildcards,
                          read_wildcards=read_wildcards)
        for key, value in thresholds.items():
            setattr(self, key, value)
        self.aligner = MultiAligner(max_insert_mismatch_frac, START_WITHIN_SEQ1 | STOP_WITHIN_SEQ2, min_insert_overlap)
        self._build()

    def _build(self):
        a1, a2 = self.adapter1.encode('ascii'), self.adapter2.encode('ascii')
        if max(len(a1), len(a2)) > _lib.INSERT_MAX_ADAPTER:
            raise _lib.AtroposHipError("InsertAligner: adapters longer than %d bases are outside the device "
                                       "envelope" % _lib.INSERT_MAX_ADAPTER)
        if not (self.adapter_wildcards or self.read_wildcards):
            valid = set("ACGTRYSWKMBDHVN")
            if not (set(self.adapter1) <= valid and set(self.adapter2) <= valid):
                raise _lib.AtroposHipError("InsertAligner without wildcard matching needs upper-case IUPAC adapters")
        ld = _lib.INSERT_MAX_READ + 1
        # host-side tables, evaluated with the reference's Python semantics
        self._rmp_insert = np.ascontiguousarray(rmp_table(self.match_probability, ld - 1, **self.base_probs))
        if self.base_probs == dict(match_prob=0.25, mismatch_prob=0.75):
            self._rmp_adapter = self._rmp_insert                      # match_probability(m, n) uses the defaults
        else:
            self._rmp_adapter = np.ascontiguousarray(rmp_table(self.match_probability, ld - 1))
        self._mm = np.array([round(a * self.max_adapter_mismatch_frac) for a in range(_lib.INSERT_MAX_ADAPTER + 1)],
                            dtype=np.int32)                          # align/__init__.py:290
        cfg = _lib.InsertConfig()
        cfg.adapter1, cfg.alen1 = a1, len(a1)
        cfg.adapter2, cfg.alen2 = a2, len(a2)
        cfg.insert_max_rmp, cfg.adapter_max_rmp = float(self.insert_max_rmp), float(self.adapter_max_rmp)
        cfg.min_insert_overlap = int(self.min_insert_overlap)
        cfg.max_insert_mismatch_frac = self.max_insert_mismatch_frac
        cfg.min_adapter_overlap = int(self.min_adapter_overlap)
        cfg.max_adapter_mismatch_frac = self.max_adapter_mismatch_frac
        cfg.adapter_check_cutoff = int(self.adapter_check_cutoff)
        cfg.adapter_wildcards, cfg.read_wildcards = int(bool(self.adapter_wildcards)), int(bool(self.read_wildcards))
        cfg.rmp_insert = self._rmp_insert.ctypes.data
        cfg.rmp_adapter = self._rmp_adapter.ctypes.data
        cfg.rmp_ld = ld
        cfg.max_mismatch_by_alen = self._mm.ctypes.data
        cfg.n_mismatch = len(self._mm)
        self._cfg = cfg
        self._handle = self._backend.insert_aligner_create(cfg)

    def __del__(self):
        try:
            if self._handle is not None:
                self._backend.insert_aligner_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def pack(self, reads, check=False, max_len=None, cased=False):
        """Pack one side of the pairs (DNA15 codes as bit planes, the "plane64" layout of
        atr_pack_planes).  With ``check`` EVERY base must be an upper-case IUPAC letter -- stricter than
        match_insert, which only complements read 2 up to the length of read 1; ``pack_pair`` applies
        exactly that rule (and handles soft-masked reads).  ``max_len``: layout width (both sides of a
        batch need the same one).  ``cased``: the case-sensitive table of soft-masked batches
        (``atr_insert_match_batch_coded``; both sides must use it)."""
        if isinstance(reads, ReadBatch):
            if reads.table_kind not in (_lib.TABLE_DNA15, _lib.TABLE_CUSTOM) or reads.layout != "plane64":
                raise ValueError("the insert aligner needs reads packed by InsertAligner.pack (DNA15 codes, plane64 layout)")
            return reads
        be = self._backend
        table = be.case_sensitive_table() if cased else be.translate_table(_lib.TABLE_DNA15)
        if isinstance(reads, (list, tuple)):
            mat, lens = _as_ascii_matrix(reads)
            if max_len is not None and max_len > mat.shape[1]:
                mat = np.pad(mat, ((0, 0), (0, max_len - mat.shape[1])))
            ascii_t, lens_t = torch.from_numpy(mat).to(be.device), torch.from_numpy(lens).to(be.device)
            if max_len is None:
                max_len = int(lens.max()) if len(reads) else 0
        else:
            ascii_t = reads.to(be.device)
            lens_t, max_len = None, ascii_t.shape[1]
        if max_len > _lib.INSERT_MAX_READ:
            raise _lib.AtroposHipError("InsertAligner: reads longer than %d bases are outside the device envelope"
                                       % _lib.INSERT_MAX_READ)
        if check:
            packed, bad = be.pack_reads(ascii_t, lens_t, max_len, table, count_invalid=True, planes=True)
            if bad:
                raise ValueError("%d read(s) contain bases without an upper-case IUPAC code; the device insert "
                                 "aligner cannot reverse-complement them" % bad)
            batch = ReadBatch(packed, lens_t, ascii_t.shape[0], max_len, _lib.TABLE_DNA15, table, layout="plane64")
            batch.uncoded_reads = 0
            return batch
        packed, bad = be.pack_reads(ascii_t, lens_t, max_len, table, count_invalid=True, planes=True)
        batch = ReadBatch(packed, lens_t, ascii_t.shape[0], max_len, _lib.TABLE_CUSTOM if cased else _lib.TABLE_DNA15, table,
                          layout="plane64")
        batch.uncoded_reads = bad              # reads with a character the table has no code for (anywhere in the read)
        return batch

    def pack_pair(self, reads1, reads2):
        """Both sides of a batch of pairs, in one layout width, validated the way match_insert does it:
        read 2 is cut to the length of read 1 before it is reverse-complemented
        (reference: align/__init__.py:259-267), so only that prefix of read 2 has to consist of
        upper-case IUPAC letters.  Sides that are already ReadBatches are taken as they are."""
        width = None
        if isinstance(reads1, (list, tuple)) and isinstance(reads2, (list, tuple)):
            # whatever the longest read of each side is
            width = max([len(r) for r in reads1] + [len(r) for r in reads2] + [0])
        prepacked = isinstance(reads2, ReadBatch)
        b1, b2 = self.pack(reads1, max_len=width), self.pack(reads2, max_len=width)
        if (getattr(b1, "uncoded_reads", 0) or getattr(b2, "uncoded_reads", 0)) and not (
                prepacked or isinstance(reads1, ReadBatch)):
            # soft-masked reads (or characters without any code): match_insert tells the cases apart in the insert
            # compare and folds them in the adapter compares -- both sides again, with the case-sensitive codes
            b1, b2 = self.pack(reads1, max_len=width, cased=True), self.pack(reads2, max_len=width, cased=True)
        if b1.nreads != b2.nreads:
            raise ValueError("need as many first reads as second reads")
        if (b1.table_kind == _lib.TABLE_CUSTOM) != (b2.table_kind == _lib.TABLE_CUSTOM):
            raise ValueError("both sides of a soft-masked batch must be packed with the case-sensitive table")
        if not prepacked and b1.max_len == b2.max_len:
            bad = self._backend.planes_count_uncoded(b2.packed, b2.lens, b1.lens, b2.nreads, b2.max_len)
            if bad:
                raise ValueError("%d second read(s) contain bases without an upper-case IUPAC code where they face "
                                 "the first read; the device insert aligner cannot reverse-complement them" % bad)
        return b1, b2

    def match_insert_batch(self, reads1, reads2):
        """Batched ``match_insert``; reads1/reads2: ReadBatch (DNA15), list of str, or
        uint8 [n, width] ASCII tensors of equal-length reads.  Lists with reads of more than 320 bases: those pairs
        go through ``match_insert`` one by one (``_match_insert_long``), the others through the kernel."""
        if (isinstance(reads1, (list, tuple)) and isinstance(reads2, (list, tuple)) and len(reads1) == len(reads2) and reads1
                and max(max(map(len, reads1)), max(map(len, reads2))) > _lib.INSERT_MAX_READ):
            long_ix = [i for i, (a, b) in enumerate(zip(reads1, reads2)) if max(len(a), len(b)) > _lib.INSERT_MAX_READ]
            short_ix = [i for i in range(len(reads1)) if max(len(reads1[i]), len(reads2[i])) <= _lib.INSERT_MAX_READ]
            rec = torch.zeros((len(reads1), 3, 8), dtype=torch.int16)
            rec[:, :, 1] = -1
            if short_ix:
                part = self.match_insert_batch([reads1[i] for i in short_ix], [reads2[i] for i in short_ix]).records
                rec[torch.tensor(short_ix)] = part.cpu()
            for i in long_ix:
                res = self.match_insert(reads1[i], reads2[i])
                if res is not None:
                    rec[i, 0, :6] = torch.tensor(res[0], dtype=torch.int16)
                    for t in (1, 2):
                        if res[t] is not None:
                            m = res[t]
                            rec[i, t, :6] = torch.tensor([m.astart, m.astop, m.rstart, m.rstop, m.matches, m.errors], dtype=torch.int16)
            return InsertResult(rec.to(self._backend.device) if hasattr(self._backend, "device") else rec)
        b1, b2 = self.pack_pair(reads1, reads2)
        max_len = max(b1.max_len, b2.max_len)
        if b1.max_len != b2.max_len:
            raise ValueError("both read batches must be packed with the same max_len (got %d and %d)"
                             % (b1.max_len, b2.max_len))
        rec = self._backend.insert_match_batch(self._handle, b1.packed, b1.lens, b2.packed, b2.lens, b1.nreads, max_len,
                                               cased=b1.table_kind == _lib.TABLE_CUSTOM)
        return InsertResult(rec)

    def match_insert_correct_batch(self, planes1, planes2, seq1, qual1, seq2, qual2, mismatch_action="liberal",
                                   min_qual_difference=1, changed=None, newlen=None):
        """``match_insert`` of every pair AND, where the insert match has errors, ``correct_errors(read1, read2,
        insert_match, truncate_seqs=True)`` in place on the ASCII matrices -- the first two steps of
        ``InsertAdapterCutter.__call__`` (commands/trim/modifiers.py:385-404) as one kernel
        (``atr_insert_match_correct_batch``): same records, bytes and counts as ``match_insert_batch`` followed by the
        backend's ``insert_correct_batch``, the reads' planes streamed once.  planes1 / planes2: plane64 ReadBatches
        (``pack``) of the matrices' rows; seq / qual: uint8 [n, width] tensors.  Returns (InsertResult, changed,
        newlen)."""
        from ..modifiers import COMP_TABLE
        action = {"N": 0, "conservative": 1, "liberal": 2}[mismatch_action]
        if planes1.table_kind == _lib.TABLE_CUSTOM or planes2.table_kind == _lib.TABLE_CUSTOM:
            raise ValueError("the fused call takes DNA15-packed reads")
        rec, changed, newlen = self._backend.insert_match_correct_batch(
            self._handle, planes1, planes2, seq1, qual1, seq2, qual2, action, min_qual_difference, COMP_TABLE, changed, newlen)
        return InsertResult(rec), changed, newlen

    def match_insert(self, seq1, seq2):
        """Use the insert overlap to find the adapters of one pair.

        Returns None if the inserts do not match, else ``(insert_match, Match1, Match2)``
        where the Matches are None when the overhang is too short for an adapter match
        (reference: align/__init__.py:250-377)."""
        n = min(len(seq1), len(seq2))
        seq1.encode('ascii')
        for base in reversed(seq2[:n]):           # reverse_complement(seq2): KeyError on unknown bases
            BASE_COMPLEMENTS[base]
        if max(len(seq1), len(seq2)) > _lib.INSERT_MAX_READ:
            return self._match_insert_long(seq1, seq2)
        be = self._backend
        if hasattr(be, "insert_match_one") and max(len(seq1), len(seq2)) <= _lib.INSERT_MAX_READ:
            table = be.translate_table(_lib.TABLE_DNA15)
            b1, b2 = seq1.encode('ascii'), seq2.encode('ascii')
            if b"\0" not in b1.translate(table) and b"\0" not in b2.translate(table):      # upper-case IUPAC letters only
                rec = be.insert_match_one(self._handle, b1, b2)
                if rec[1] < 0:
                    return None
                ms = [None if rec[8 * t + 1] < 0 else Match(*rec[8 * t:8 * t + 6]) for t in (1, 2)]
                return (tuple(rec[0:6]), ms[0], ms[1])
            # (soft-masked reads and uncoded characters: the batch path sorts those out)
        return self.match_insert_batch([seq1], [seq2]).results()[0]


    # Reads beyond the insert kernel's plane registers (more than 320 bases; the reference has no limit,
    # align/__init__.py:250-265).  The rule of match_insert on top of this package's own building blocks -- the general
    # MultiAligner (every overlap within the mismatch fraction, device kernel), compare_prefixes (the overhangs against
    # the adapters) and the match-probability tables -- one pair at a time: a fallback for what no paired-end Illumina
    # run produces, not a throughput path.  MultiAligner's own limit (736 bases) is the new one.
    def _match_insert_long(self, seq1, seq2):
        full1, full2 = len(seq1), len(seq2)
        common = min(full1, full2)
        read1, read2 = seq1[:common], seq2[:common]          # the longer read is cut before anything else (:259-265)
        overlaps = self.aligner.locate(reverse_complement(read2), read1)
        if not overlaps:
            return None
        ranked = []
        for hit in overlaps:
            offset = min(hit[0], common - hit[3])             # bases of adapter the pair shows behind the insert
            size = common - offset
            chance = self.match_probability(hit[4], size, **self.base_probs)
            if chance <= self.insert_max_rmp:
                ranked.append((chance, hit, offset, size))
        ranked.sort(key=lambda entry: entry[0])              # least likely by chance first; stable among equals (:371)
        for _, hit, offset, size in ranked:
            found = self._adapters_behind_insert(hit, offset, size, read1, read2, full1, full2)
            if found is not None:
                return found
        return None

    def _adapters_behind_insert(self, hit, offset, size, read1, read2, full1, full2):
        if offset < self.min_adapter_overlap:                # overlap, but no overhang worth an adapter match (:272-279)
            return (hit, None, None)
        sides = []
        for read, adapter in ((read1, self.adapter1), (read2, self.adapter2)):
            # (the overhang is the REFERENCE of the compare: the read side takes the adapter_wildcards table, :285-288)
            cmp_rec = compare_prefixes(read[size:], adapter, wildcard_ref=self.adapter_wildcards, wildcard_query=self.read_wildcards)
            alen = min(offset, len(adapter))
            sides.append((cmp_rec, alen, round(alen * self.max_adapter_mismatch_frac)))
        (c1, alen1, cap1), (c2, alen2, cap2) = sides
        if c1[5] > cap1 and c2[5] > cap2:
            return None
        if min(alen1, alen2) > self.adapter_check_cutoff:
            if self.match_probability(c1[4], alen1) * self.match_probability(c2[4], alen2) > self.adapter_max_rmp:
                return None
        errors = min(c1[5], c2[5])                           # both Matches carry the better side's count (:308-314)

        def as_match(alen, read_len):
            alen = min(alen, read_len - size)
            bad = min(alen, errors)
            return Match(0, alen, size, read_len, alen - bad, bad)
        return (hit, as_match(alen1, full1), as_match(alen2, full2))


_DEFAULT_RMP = None


def _default_rmp():
    """The shared default ``RandomMatchProbability()`` (the reference evaluates its default
    argument once at import, align/__init__.py:208)."""
    global _DEFAULT_RMP
    if _DEFAULT_RMP is None:
        _DEFAULT_RMP = RandomMatchProbability()
    return _DEFAULT_RMP
