"""Device-resident read batches and result arrays.

The reference hands one Python ``str`` per call to ``Aligner.locate``
(atropos/align/_align.pyx:266); the batched twins here take a ``ReadBatch``:
ASCII reads translated through a 256-entry table and packed to 4 bits per base in
the tile64 layout (include/atropos_hip.h), resident in GPU memory.
"""
import numpy as np
import torch

from . import _lib


def _as_ascii_matrix(reads):
    """list of str/bytes -> (uint8 ndarray [n, max_len], int32 lens)."""
    n = len(reads)
    if n and set(map(type, reads)) == {str}:              # (one pass in C; a generator of isinstance calls costs 20 ms per 65 k)
        flat = "".join(reads).encode("ascii")             # one pass in C (UnicodeEncodeError as the per-read encode)
        lens = np.fromiter(map(len, reads), dtype=np.int32, count=n)
    else:
        rows = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in reads]
        flat = b"".join(rows)
        lens = np.fromiter(map(len, rows), dtype=np.int32, count=n)
    max_len = int(lens.max()) if n else 0
    width = max(max_len, 1)
    data = np.frombuffer(flat, dtype=np.uint8)
    if n and int(lens.min()) == max_len and max_len > 0:  # equal lengths: the text IS the matrix
        return data.reshape(n, max_len).copy(), lens
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
        mat, lens = _as_ascii_matrix(reads)
        if len(reads) == 0:
            be = backend or _lib.get_backend()
            if table is None:
                table = be.translate_table(table_kind)
            return cls(be.empty((16,), torch.uint8), None, 0, 0, table_kind, table)
        if int(lens.min()) == mat.shape[1]:               # equal lengths: no length array (the kernels' uniform path)
            lens = None
        return cls.from_ascii(mat, lens, None, table_kind, table, backend, planes=planes)


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
        """List of 6-tuples / None, exactly what per-read ``locate`` calls return."""
        arr = self.numpy()[:, :6]
        out = list(zip(*arr.T.tolist()))                  # the tuples are built in C: six column lists, one zip
        for i in np.flatnonzero(arr[:, 1] < 0).tolist():
            out[i] = None
        return out
