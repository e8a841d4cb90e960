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
from collections import namedtuple

from .. import _lib
from ..batch import ReadBatch, LocateResult
from ..util import RandomMatchProbability, reverse_complement

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
        self._set_reference(reference)

    # -- construction / destruction -------------------------------------------------
    def _set_reference(self, reference):
        ref_bytes = reference.encode('ascii')                                 # _align.pyx:243
        self._release()
        self.str_reference = reference
        self._ref_bytes = ref_bytes
        if len(ref_bytes) == 0:
            # An empty reference can never produce a non-empty alignment (the reference
            # implementation returns None for every query); no device state is needed.
            self._table_kind, self._table = _lib.TABLE_DNA15, None
            return
        be = self._backend
        self._handle = be.aligner_create(ref_bytes, self.max_error_rate, self.flags, self.wildcard_ref,
                                         self.wildcard_query, self._min_overlap, self._indel_cost)
        self._table_kind, self._table = be.aligner_query_table(self._handle)

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
        """Always None: the DP matrix is never materialised on the device."""
        return None

    def enable_debug(self):
        raise NotImplementedError(
            "the DP-matrix debug dump (reference _align.pyx:259-264) is a CPU-only debugging aid "
            "and is not provided by the device implementation")

    # -- packing ----------------------------------------------------------------------
    @property
    def table_kind(self):
        """Which translate table (``_lib.TABLE_*``) query batches must be packed with."""
        return self._table_kind

    def pack(self, reads):
        """Pack reads for this aligner.  ``reads``: a ReadBatch (returned unchanged if
        compatible), a sequence of str/bytes, or a uint8 [nreads, width] array/tensor."""
        if isinstance(reads, ReadBatch):
            self._check_batch(reads)
            return reads
        kind = self._table_kind
        table = self._table if self._table is not None else self._backend.translate_table(kind)
        if isinstance(reads, (list, tuple)):
            return ReadBatch.from_strings(reads, kind, table, self._backend)
        return ReadBatch.from_ascii(reads, None, None, kind, table, self._backend)

    def _check_batch(self, batch):
        if self._handle is None:
            return
        if batch.table_kind != self._table_kind or (
                self._table_kind == _lib.TABLE_CUSTOM and batch.table != self._table):
            raise ValueError("read batch was packed with translate table %d but this aligner needs %d "
                             "(use aligner.pack(reads))" % (batch.table_kind, self._table_kind))

    # -- alignment ----------------------------------------------------------------------
    def locate_batch(self, reads):
        """Batched ``locate``: one result record per read (see LocateResult)."""
        batch = self.pack(reads)
        be = self._backend
        if self._handle is None:                      # empty reference: nothing ever matches
            import torch
            rec = be.empty((batch.nreads, 8), torch.int16)
            rec.zero_()
            rec[:, 1] = -1
            return LocateResult(rec)
        return LocateResult(be.locate_batch(self._handle, batch.packed, batch.lens, batch.nreads, batch.max_len))

    def locate(self, query):
        """locate(query) -> (refstart, refstop, querystart, querystop, matches, errors)

        Find the query within the reference associated with this aligner
        (reference: Aligner.locate, _align.pyx:266-491)."""
        if not isinstance(query, str):
            raise TypeError("query must be str")
        query.encode('ascii')                          # UnicodeEncodeError like the reference (:281)
        return self.locate_batch([query]).tuples()[0]


def locate(reference, query, max_error_rate, flags=SEMIGLOBAL, wildcard_ref=False, wildcard_query=False,
           min_overlap=1):
    """Convenience wrapper (reference: _align.pyx:496-499; the indel cost is always 1)."""
    aligner = Aligner(reference, max_error_rate, flags, wildcard_ref, wildcard_query)
    aligner.min_overlap = min_overlap
    return aligner.locate(query)


# Common match-result object returned by aligners (reference align/__init__.py:51-175)

class Match(object):
    """An alignment match.

    Args:
        astart, astop: match interval within the adapter.
        rstart, rstop: match interval within the read.
        matches: number of matching bases.
        errors: number of mismatching bases (and indels).
        front: whether the match is to the front of the read (guessed from
            ``rstart == 0`` when None).
        adapter, read: the Adapter and the read object.
    """
    __slots__ = ['astart', 'astop', 'rstart', 'rstop', 'matches', 'errors', 'front', 'adapter', 'read', 'length']

    def __init__(self, astart, astop, rstart, rstop, matches, errors, front=None, adapter=None, read=None):
        self.astart = astart
        self.astop = astop
        self.rstart = rstart
        self.rstop = rstop
        self.matches = matches
        self.errors = errors
        self.front = self._guess_is_front() if front is None else front
        self.adapter = adapter
        self.read = read
        # Number of aligned characters in the adapter; with indels this may differ from
        # the number of characters in the read.
        self.length = self.astop - self.astart
        if self.length <= 0:
            raise ValueError('Match length must be >= 0')
        if self.length - self.errors <= 0:
            raise ValueError('A Match requires at least one matching position.')

    def __repr__(self):
        return 'Match(astart={0}, astop={1}, rstart={2}, rstop={3}, matches={4}, errors={5})'.format(
            self.astart, self.astop, self.rstart, self.rstop, self.matches, self.errors)

    def copy(self):
        return Match(self.astart, self.astop, self.rstart, self.rstop, self.matches, self.errors, self.front,
                     self.adapter, self.read)

    def _guess_is_front(self):
        return self.rstart == 0

    def wildcards(self, wildcard_char='N'):
        """The read characters that the adapter's wildcard characters matched (not
        reliable with indels, as the alignment itself is not kept)."""
        wildcards = [
            self.read.sequence[self.rstart + i]
            for i in range(self.length)
            if (self.adapter.sequence[self.astart + i] == wildcard_char and
                self.rstart + i < len(self.read.sequence))]
        return ''.join(wildcards)

    def rest(self):
        """The part of the read before the match for a front adapter, after it otherwise."""
        if self.front:
            return self.read.sequence[:self.rstart]
        return self.read.sequence[self.rstop:]

    def get_info_record(self):
        seq = self.read.sequence
        qualities = self.read.qualities
        if qualities is None:
            qualities = ''
        rsize = rsize_total = self.rstop - self.rstart
        if self.front and self.rstart > 0:
            rsize_total = self.rstop
        elif not self.front and self.rstop < len(seq):
            rsize_total = len(seq) - self.rstart
        return MatchInfo(
            self.read.name, self.errors, self.rstart, self.rstop, seq[0:self.rstart],
            seq[self.rstart:self.rstop], seq[self.rstop:], self.adapter.name, qualities[0:self.rstart],
            qualities[self.rstart:self.rstop], qualities[self.rstop:], self.front, self.astop - self.astart,
            rsize, rsize_total)


MatchInfo = namedtuple("MatchInfo", (
    "read_name", "errors", "rstart", "rstop", "seq_before", "seq_adapter", "seq_after", "adapter_name",
    "qual_before", "qual_adapter", "qual_after", "is_front", "asize", "rsize_adapter", "rsize_total"))
