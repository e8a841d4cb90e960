"""Device-resident read batches and result arrays.

The reference hands one Python ``str`` per call to ``Aligner.locate``
(atropos/align/_align.pyx:266); the batched twins here take a ``ReadBatch``:
ASCII reads translated through a 256-entry table and packed to 4 bits per base in
the tile64 layout (include/atropos_hip.h), resident in GPU memory.
"""
from collections.abc import Sequence

import numpy as np
import torch

from . import _lib


def _as_ascii_matrix(reads, readonly_ok=False):
    """list of str/bytes -> (uint8 ndarray [n, max_len], int32 lens).  readonly_ok: equal-length reads may come back as a
    read-only view of the joined text (the caller copies it into a staging buffer anyway)."""
    n = len(reads)
    flat = None
    if n:
        try:
            flat = "".join(reads).encode("ascii")         # one pass in C (UnicodeEncodeError as the per-read encode);
        except TypeError:                                 # TypeError: not all str -- no type check of its own (0.7 ms per 65 k)
            flat = None
    if flat is not None:
        distinct = set(map(len, reads))                   # (C-level pass; the int32 array below costs twice as much)
        if len(distinct) == 1:
            lens = np.full(n, distinct.pop(), dtype=np.int32)
        else:
            lens = np.fromiter(map(len, reads), dtype=np.int32, count=n)
    else:
        rows = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in reads]
        flat = b"".join(rows)
        lens = np.fromiter(map(len, rows), dtype=np.int32, count=n)
    max_len = int(lens.max()) if n else 0
    width = max(max_len, 1)
    data = np.frombuffer(flat, dtype=np.uint8)
    if n and int(lens.min()) == max_len and max_len > 0:  # equal lengths: the text IS the matrix (a read-only view of it)
        return (data.reshape(n, max_len) if readonly_ok else data.reshape(n, max_len).copy()), lens
    mat = np.zeros((n, width), dtype=np.uint8)
    if data.size:
        mat[np.arange(width, dtype=np.int32)[None, :] < lens[:, None]] = data      # row-major fill of the ragged rows
    return mat, lens


class ReadBatch(object):
    """4-bit packed reads on the device.

    Attributes:
        packed: uint8 tensor, tile64 layout.
        lens: int32 tensor [nreads] or None (all reads ``max_len`` long).
        nreads, max_len: batch shape.
        table_kind, table: which translate table the reads were packed with.
    """

    def __init__(self, packed, lens, nreads, max_len, table_kind, table, layout="tile64"):
        self.packed, self.lens = packed, lens
        self.nreads, self.max_len = int(nreads), int(max_len)
        self.table_kind, self.table = table_kind, bytes(table)
        self.layout = layout            # "tile64" (4-bit codes) or "plane64" (bit planes; insert aligner)

    def __len__(self):
        return self.nreads

    @classmethod
    def from_ascii(cls, ascii_2d, lens=None, max_len=None, table_kind=_lib.TABLE_DNA15, table=None,
                   backend=None, starts=None, planes=False):
        """ascii_2d: uint8 tensor/ndarray [nreads, width] of ASCII codes; lens: per-read
        lengths (None = every read is ``width`` long); starts: per-read first base (int32
        tensor): the batch then holds the slices ``read[start:]``; planes: the plane64 layout
        (bit planes of the codes) instead of tile64."""
        be = backend or _lib.get_backend()
        if isinstance(ascii_2d, np.ndarray):
            ascii_2d = torch.from_numpy(np.ascontiguousarray(ascii_2d, dtype=np.uint8))
        if ascii_2d.dtype != torch.uint8 or ascii_2d.dim() != 2:
            raise ValueError("reads must be a uint8 [nreads, width] array")
        ascii_2d = ascii_2d.to(be.device)
        if ascii_2d.stride(1) != 1:
            ascii_2d = ascii_2d.contiguous()
        nreads, width = ascii_2d.shape
        if lens is not None:
            if isinstance(lens, np.ndarray):
                lens = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32))
            lens = lens.to(device=be.device, dtype=torch.int32).contiguous()
            if lens.numel() != nreads:
                raise ValueError("lens must have one entry per read")
        if max_len is None:
            max_len = width if lens is None else (int(lens.max().item()) if nreads else 0)
        if max_len > width and nreads:
            raise ValueError("max_len exceeds the row width")
        limit = _lib.MAX_READ_LEN if (planes or starts is not None) else _lib.MAX_LONG_READ_LEN
        if max_len > limit:
            raise ValueError("reads longer than %d bases are outside the device kernels' envelope" % limit)
        if table is None:
            if table_kind == _lib.TABLE_CUSTOM:
                raise ValueError("a custom table must be given explicitly")
            table = be.translate_table(table_kind)
        if starts is not None:
            starts = starts.to(device=be.device, dtype=torch.int32).contiguous()
            full = lens if lens is not None else torch.full((nreads,), width, dtype=torch.int32, device=be.device)
            packed = be.pack_reads(ascii_2d, full, max_len, bytes(table), starts=starts)
            return cls(packed, (full - starts).clamp_(min=0), nreads, max_len, table_kind, table)
        if planes:
            packed = be.pack_reads(ascii_2d, lens, max_len, bytes(table), planes=True)
            return cls(packed, lens, nreads, max_len, table_kind, table, layout="plane64")
        packed = be.pack_reads(ascii_2d, lens, max_len, bytes(table))
        return cls(packed, lens, nreads, max_len, table_kind, table)

    @classmethod
    def from_strings(cls, reads, table_kind=_lib.TABLE_DNA15, table=None, backend=None, planes=False):
        """reads: sequence of ``str`` (ASCII) or ``bytes``."""
        mat, lens = _as_ascii_matrix(reads, readonly_ok=True)
        if len(reads) == 0:
            be = backend or _lib.get_backend()
            if table is None:
                table = be.translate_table(table_kind)
            return cls(be.empty((16,), torch.uint8), None, 0, 0, table_kind, table)
        if int(lens.min()) == mat.shape[1]:               # equal lengths: no length array (the kernels' uniform path)
            lens = None
        be = backend or _lib.get_backend()
        stage = getattr(be, "stage_host_bytes", None)
        if stage is not None:                             # page-locked staging: one memcpy + one DMA instead of a pageable copy
            mat = stage(mat)
        elif not mat.flags.writeable:
            mat = mat.copy()
        return cls.from_ascii(mat, lens, None, table_kind, table, be, planes=planes)


class RecordTuples(Sequence):
    """Read-only sequence of result tuples over an int16 [n, 6] array: item i is ``None`` (refstop == -1) or the
    tuple of its six ints.  Compares equal to a list (or another RecordTuples) with the same items."""

    __slots__ = ("_arr",)
    _CHUNK = 8192

    def __init__(self, arr):
        self._arr = arr

    def __len__(self):
        return self._arr.shape[0]

    @staticmethod
    def _make(arr):
        out = list(zip(*arr.T.tolist()))                  # the tuples are built in C: six column lists, one zip
        for i in np.flatnonzero(arr[:, 1] < 0).tolist():
            out[i] = None
        return out

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._make(self._arr[i])
        row = self._arr[i]
        return None if row[1] < 0 else tuple(row.tolist())

    def __iter__(self):
        for lo in range(0, self._arr.shape[0], self._CHUNK):
            for t in self._make(self._arr[lo:lo + self._CHUNK]):
                yield t

    def tolist(self):
        return self._make(self._arr)

    def __eq__(self, other):
        if isinstance(other, RecordTuples):
            return self._arr.shape == other._arr.shape and self.tolist() == other.tolist()
        if isinstance(other, (list, tuple)):
            return len(other) == len(self) and self.tolist() == list(other)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "RecordTuples(%d records)" % len(self)


class LocateResult(object):
    """Result records of a batched ``locate``: int16 tensor [nreads, 8] on the device,
    columns (refstart, refstop, querystart, querystop, matches, errors, 0, 0);
    ``refstop == -1`` marks "no match" (the reference returns ``None``)."""

    def __init__(self, records):
        self.records = records

    def __len__(self):
        return self.records.shape[0]

    def numpy(self):
        return self.records.cpu().numpy()

    def found(self):
        return self.records[:, 1] >= 0

    def tuples(self):
        """The 6-tuples / None that per-read ``locate`` calls return, as a read-only sequence (``RecordTuples``): the
        records come to the host once, the Python tuples are made when they are looked at -- 65 536 of them cost more
        than packing, aligning and copying the batch together."""
        return RecordTuples(self.numpy()[:, :6])
