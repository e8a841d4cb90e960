"""Device-resident FASTQ batches: the MI355X replacement for the reference's per-record
``FastqReader`` / ``Sequence`` / ``FastqFormat`` path (atropos/io/_seqio.pyx:163-245,
atropos/io/seqio.py:686-700).

A chunk of the file is uploaded as raw bytes; the library finds and validates the records
(``atr_fastq_index``), every trimming step is an update of a kept interval per read, and the
surviving records are formatted on the device (``atr_fastq_emit``).  No per-read Python
object exists anywhere on this path.
"""
import os

import torch

from . import _lib


class FormatError(Exception):
    """Malformed input file (atropos/io/seqio.py FormatError)."""


def _universal_newlines(raw):
    """What Python's text mode (the reference's xopen(..., 'r')) makes of a line's bytes."""
    return raw.replace(b"\r\n", b"\n").replace(b"\r", b"\n")


def _text(b):
    return bytes(b).decode("utf-8", "replace")


class FastqBatch(object):
    """Whole FASTQ records in device memory.

    Attributes:
        data: uint8 tensor with the chunk's bytes (padded to a multiple of 16).
        nbytes: bytes of text in ``data``.
        records: int32 [n, 8] descriptors (``atr_fastq_record``): name_off, name_len,
            seq_off, seq_len, qual_off, qual_len, flags, reserved.
        line_ends: uint32 positions of the line terminators (kept for ``head``).

    Line ends follow Python's universal newlines ("\\n", "\\r\\n", lone "\\r"), which is how
    the reference reads its input.
    """

    def __init__(self, data, nbytes, records, backend, line_ends=None):
        self.data, self.nbytes, self.records, self.backend = data, nbytes, records, backend
        self.line_ends = line_ends

    def head(self, nrec):
        """The first ``nrec`` records as a batch of their own, and the number of bytes of text
        they occupy (paired files are consumed in lock step: the shorter chunk decides)."""
        nrec = min(int(nrec), len(self))
        consumed = int(self.line_ends[4 * nrec - 1].item()) + 1 if nrec else 0
        return FastqBatch(self.data, self.nbytes, self.records[:nrec], self.backend, self.line_ends), consumed

    def __len__(self):
        return self.records.shape[0]

    @property
    def seq_lens(self):
        return self.records[:, 3].contiguous()

    @classmethod
    def from_matrix(cls, ascii_2d, lens=None, backend=None):
        """A batch over a uint8 [n, width] matrix of sequences already on the device (no names,
        no qualities): lets the record-based packer and the compacting adapter stages run on
        reads that did not come from a FASTQ file."""
        be = backend or _lib.get_backend()
        n, width = ascii_2d.shape
        nbytes = n * width
        if nbytes >= (1 << 32) - 16:
            raise ValueError("the matrix must be smaller than 4 GiB")
        data = be.empty(((nbytes + 15) // 16 * 16 + 16,), torch.uint8)
        data[:nbytes].view(n, width).copy_(ascii_2d)
        data[nbytes:].zero_()
        records = torch.zeros((n, 8), dtype=torch.int32, device=data.device)
        off = torch.arange(n, device=data.device, dtype=torch.int64) * width
        records[:, 2] = off.to(torch.int32) if nbytes < (1 << 31) else (off - (off >= (1 << 31)) * (1 << 32)).to(torch.int32)
        records[:, 3] = width if lens is None else lens.to(torch.int32)
        return cls(data, nbytes, records, be)

    @classmethod
    def from_bytes(cls, buf, final=True, backend=None):
        """Index ``buf`` (bytes-like FASTQ text that starts at a record boundary).

        final=True: ``buf`` is the rest of the file -- a missing last newline is tolerated
        and leftover lines raise "FASTQ file ended prematurely" (_seqio.pyx:244-245).
        final=False: only whole records are taken; the number of bytes consumed is returned
        so that the caller can prepend the remainder to its next chunk.
        Returns (batch, consumed_bytes)."""
        be = backend or _lib.get_backend()
        buf = bytes(buf) if not isinstance(buf, (bytes, bytearray)) else buf
        unterminated = bool(final and len(buf) and not buf.endswith((b"\n", b"\r")))
        if unterminated:
            buf = bytes(buf) + b"\n"
        nbytes = len(buf)
        if nbytes >= (1 << 32) - 16:
            raise ValueError("a FASTQ batch must be smaller than 4 GiB; read the file in chunks")
        padded = (nbytes + 15) // 16 * 16 + 16
        host = torch.zeros((padded,), dtype=torch.uint8)
        if nbytes:
            host[:nbytes] = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        data = be.empty((padded,), torch.uint8)
        data.copy_(host)
        return cls.from_device(data, nbytes, final, be, host_text=buf, unterminated=unterminated)

    @classmethod
    def from_device(cls, data, nbytes, final=True, backend=None, host_text=None, unterminated=False):
        """Index FASTQ text that already sits in device memory: ``data`` is a uint8 tensor,
        16-byte aligned, readable up to the next multiple of 16 beyond ``nbytes`` plus one
        byte, whose text ends in a line end.  Returns (batch, consumed_bytes)."""
        be = backend or _lib.get_backend()
        records, line_ends, nlines, err = be.fastq_index(data, nbytes)
        nrec = nlines // 4
        if err != _lib.INT64_MAX:
            if host_text is None:
                host_text = bytes(data[:nbytes].cpu().numpy().tobytes())
            cls._raise_format_error(host_text, line_ends, err)
        if final and nlines % 4 != 0:
            # the reference validates the lines of the incomplete last record as it reads them
            # (_seqio.pyx:208-238) before it runs out of input (:244-245)
            tail = [int(v) for v in line_ends[max(4 * nrec - 1, 0):nlines].cpu().tolist()]
            if nrec == 0:
                tail = [-1] + tail
            if host_text is None:
                host_text = bytes(data[:nbytes].cpu().numpy().tobytes())
            lines = [_universal_newlines(bytes(host_text[tail[i] + 1:tail[i + 1] + 1])) for i in range(len(tail) - 1)]
            if unterminated:
                # the file's last line had no line end: the reference still drops its last
                # character (line[:strip]); give the line a "\n" in place of that character
                lines[-1] = lines[-1][:-2] + b"\n" if len(lines[-1]) > 1 else lines[-1]
            if not lines[0].startswith(b"@"):
                raise FormatError("Line {0} in FASTQ file is expected to start with '@', but found {1!r}".format(
                    1, _text(lines[0])[:10]))
            if len(lines) >= 3 and lines[2] != b"+\n":
                plus, name = lines[2][:-1], lines[0][1:-1]
                if not plus.startswith(b"+"):
                    raise FormatError("Line {0} in FASTQ file is expected to start with '+', but found {1!r}".format(
                        3, _text(plus)[:10]))
                if len(plus) > 1 and plus[1:] != name:
                    raise FormatError(
                        "At line {0}: Sequence descriptions in the FASTQ file don't match "
                        "({1!r} != {2!r}).\n"
                        "The second sequence description must be either empty "
                        "or equal to the first description.".format(3, _text(name), _text(plus[1:])))
            raise FormatError("FASTQ file ended prematurely")
        consumed = nbytes
        if not final:
            consumed = int(line_ends[4 * nrec - 1].item()) + 1 if nrec else 0
        return cls(data, nbytes, records[:nrec], be, line_ends), consumed

    @staticmethod
    def _raise_format_error(buf, line_ends, err):
        """Re-create the reference's message for the first invalid record."""
        r, code = err // 8, err % 8
        ends = [int(v) for v in line_ends[max(4 * r - 1, 0):4 * r + 4].cpu().tolist()]
        if r == 0:
            ends = [-1] + ends
        # the lines as the reference sees them: newline-translated, each ending in "\n"
        lines = [_universal_newlines(bytes(buf[ends[i] + 1:ends[i + 1] + 1])) for i in range(4)]
        name = _text(lines[0][1:-1])
        if code == _lib.FASTQ_ERR_AT:                                      # _seqio.pyx:209-211 / :199-201
            raise FormatError("Line {0} in FASTQ file is expected to start with '@', but found {1!r}".format(
                1, _text(lines[0])[:10]))
        if code == _lib.FASTQ_ERR_PLUS:                                    # :224-227
            raise FormatError("Line {0} in FASTQ file is expected to start with '+', but found {1!r}".format(
                3, _text(lines[2][:-1])[:10]))
        if code == _lib.FASTQ_ERR_NAME2:                                   # :229-235
            raise FormatError(
                "At line {0}: Sequence descriptions in the FASTQ file don't match "
                "({1!r} != {2!r}).\n"
                "The second sequence description must be either empty "
                "or equal to the first description.".format(3, name, _text(lines[2][:-1])[1:]))
        seq, qual = lines[1][:-1], lines[3][:-1]
        rname = name if len(name) <= 100 else name[:97] + "..."            # util.truncate_string
        cause = FormatError(
            "In read named {0!r}: length of quality sequence ({1}) and "
            "length  of read ({2}) do not match".format(rname, len(qual), len(seq)))
        raise FormatError("Error creating sequence record at line {}".format(4)) from cause

    # ------------------------------------------------------------------ views for tests / small batches
    def to_records(self, begin=None, end=None):
        """Host copies [(name, sequence, qualities, name2), ...] -- test helper, not a product path."""
        raw = bytes(self.data[:self.nbytes].cpu().numpy().tobytes())
        out = []
        rec = self.records.cpu().tolist()
        b0 = None if begin is None else begin.cpu().tolist()
        e0 = None if end is None else end.cpu().tolist()
        for i, (no, nl, so, sl, qo, ql, fl, _) in enumerate(rec):
            a = 0 if b0 is None else b0[i]
            b = sl if e0 is None else max(a, e0[i])
            name = raw[no:no + nl].decode("ascii", "replace")
            out.append((name, raw[so + a:so + b].decode("ascii", "replace"),
                        raw[qo + a:qo + b].decode("ascii", "replace"), name if fl & 1 else ""))
        return out


class RecordSource(object):
    """Read source of the device-resident adapter matchers (``Adapter.match_source``) over the
    kept intervals of a FastqBatch: packs ``sequence[begin:end]`` once per translate table."""

    def __init__(self, batch, begin, end):
        self.fq, self.begin, self.end = batch, begin, end
        self.n = len(batch)
        self._batches = {}

    def max_len(self):
        if getattr(self, "_max_len", None) is None:
            self._max_len = int((self.end - self.begin).clamp_(min=0).max().item()) if self.n else 0
        return self._max_len

    def split_long(self):
        """(source of the reads with the over-long ones emptied, their indices, an AsciiSource of the over-long
        reads): one record beyond the batch pipelines' length does not abort the chunk -- the adapter matcher sends
        those through the long-read sweep (the reference has no length limit, _align.pyx:266-291)."""
        from .adapters import AsciiSource
        lens = (self.end - self.begin).clamp_(min=0)
        long_idx = torch.nonzero(lens > _lib.MAX_READ_LEN).squeeze(1)
        long_lens = lens.index_select(0, long_idx).to(torch.int32)
        width = (int(long_lens.max().item()) + 3) // 4 * 4
        if width > _lib.MAX_LONG_READ_LEN:
            raise ValueError("reads longer than %d bases are outside the device kernels' envelope" % _lib.MAX_LONG_READ_LEN)
        rec = self.fq.records.index_select(0, long_idx)
        off = (rec[:, 2].to(torch.int64) & 0xFFFFFFFF) + self.begin.index_select(0, long_idx).to(torch.int64)
        pos = off[:, None] + torch.arange(width, device=off.device, dtype=torch.int64)[None, :]
        mat = self.fq.data[pos.clamp_(max=self.fq.data.numel() - 1)]
        mat = torch.where((mat >= 97) & (mat <= 122), mat - 32, mat)          # match_to upper-cases the read (:349)
        mat = torch.where(torch.arange(width, device=off.device)[None, :] < long_lens[:, None], mat, torch.zeros_like(mat))
        short = RecordSource(self.fq, self.begin, torch.where(lens > _lib.MAX_READ_LEN, self.begin, self.end))
        return short, long_idx, AsciiSource(mat.contiguous(), long_lens.contiguous())

    def batch_for(self, aligner):
        """The batch ``aligner.locate_batch`` is fastest on: bit planes when the two-pass pre-pass takes the
        aligner on a (ragged) batch of this size, else 4-bit codes."""
        planes = aligner._wants_planes("auto", self.n, self.max_len(), ragged=True)
        return self.batch(aligner.table_kind, aligner._table, planes=planes)

    def batch(self, table_kind, table, planes=False):
        from .batch import ReadBatch
        key = (table_kind, bytes(table), bool(planes))
        if key not in self._batches:
            be = self.fq.backend
            max_len = self.max_len()
            if max_len > _lib.MAX_READ_LEN:
                raise ValueError("reads longer than %d bases are outside the device kernels' envelope"
                                 % _lib.MAX_READ_LEN)
            # Adapter.match_to upper-cases the read first (adapters/__init__.py:349): fold that
            # into the translate table instead of touching the bytes
            folded = bytearray(table)
            for c in range(ord("a"), ord("z") + 1):
                folded[c] = table[c - 32]
            packed, lens = be.pack_records(self.fq.data, self.fq.records, self.begin, self.end, max_len, bytes(folded),
                                           planes=bool(planes))
            self._batches[key] = ReadBatch(packed, lens, self.n, max_len, table_kind, table,
                                           layout="plane64" if planes else "tile64")
        return self._batches[key]

    def sliced(self, starts):
        """The source of the reads ``read[start:]`` (start relative to the kept interval)."""
        begin = self.begin + starts.to(self.begin.dtype)
        return RecordSource(self.fq, begin, torch.maximum(self.end, begin))

    def planes(self, max_len, table_kind, table, check=False, count=False):
        """plane64 pack (insert aligner) of the kept intervals with a given max_len; no case
        folding (InsertAligner compares the reads as they are).  check: every base must have a code;
        count: leave the number of reads with an uncoded character in ``.uncoded_reads``."""
        from .batch import ReadBatch
        be = self.fq.backend
        if count:
            packed, lens, bad = be.pack_records(self.fq.data, self.fq.records, self.begin, self.end, max_len, bytes(table),
                                                count_invalid=True, planes=True)
            batch = ReadBatch(packed, lens, self.n, max_len, table_kind, table, layout="plane64")
            batch.uncoded_reads = bad
            return batch
        if check:
            packed, lens, bad = be.pack_records(self.fq.data, self.fq.records, self.begin, self.end, max_len, bytes(table),
                                                count_invalid=True, planes=True)
            if bad:
                raise ValueError("%d read(s) contain bases without an upper-case IUPAC code; the device insert "
                                 "aligner cannot reverse-complement them" % bad)
        else:
            packed, lens = be.pack_records(self.fq.data, self.fq.records, self.begin, self.end, max_len, bytes(table),
                                           planes=True)
        return ReadBatch(packed, lens, self.n, max_len, table_kind, table, layout="plane64")


# ---------------------------------------------------------------------------------------------
# file streaming: page-locked staging, threaded reads, read-ahead and write-behind
_STAGING_POOL = []           # page-locked buffers are expensive to create: released ones are kept for reuse


def _staging(nbytes, pinned):
    for i, t in enumerate(_STAGING_POOL):
        if t.numel() >= nbytes and t.is_pinned() == bool(pinned):
            return _STAGING_POOL.pop(i)
    t = torch.empty((nbytes,), dtype=torch.uint8)
    return t.pin_memory() if pinned else t


STAGING_POOL_BYTES = 4 << 30  # page-locked memory the pool keeps between runs (more starves other processes of a shared node);
if os.environ.get("ATR_STAGING_POOL_GB"):     # ATR_STAGING_POOL_GB: a host that runs file after file with many part files keeps more
    STAGING_POOL_BYTES = int(float(os.environ["ATR_STAGING_POOL_GB"]) * (1 << 30))


def _release(buffers):
    _STAGING_POOL.extend(buffers)
    del _STAGING_POOL[:-64]                                   # at most 64 (a paired run with merging into eight parts holds 56)
    total = 0
    for i in range(len(_STAGING_POOL) - 1, -1, -1):           # ... and at most STAGING_POOL_BYTES, the newest first
        total += _STAGING_POOL[i].numel()
        if total > STAGING_POOL_BYTES:
            del _STAGING_POOL[:i + 1]
            break


IO_THREADS = 8              # pread slices per chunk (a page-cached file scales to ~6 GB/s per thread); ATR_IO_THREADS
if os.environ.get("ATR_IO_THREADS"):
    IO_THREADS = max(1, int(os.environ["ATR_IO_THREADS"]))


class StageClock(object):
    """Seconds the streaming loop spent WAITING per stage (what bounds file -> file)."""

    def __init__(self):
        self.seconds = {}

    def add(self, stage, t0):
        import time
        self.seconds[stage] = self.seconds.get(stage, 0.0) + (time.perf_counter() - t0)


class ChunkedFastqReader(object):
    """Feeds a FASTQ file to the GPU in chunks of whole records.  The file is read straight into
    page-locked staging buffers (``IO_THREADS`` ``pread`` slices per chunk, no intermediate bytes objects).

    Plain files: ``READ_AHEAD`` chunks are being READ while the GPU works on the current one.  What a chunk
    carries over from the one before it (its unfinished last record; in a paired run the surplus records of the
    file with the smaller records) is known only once that chunk has been indexed on the device, but the file
    READ does not depend on it: the new bytes land behind a reserve at the front of the staging buffer, the
    carried-over tail is copied in front of them when it is known, and the host -> device copy (its own
    stream) covers both.  Rounds 2-5 read chunk i + 1 only after chunk i was indexed -- read + upload in one
    chain, the upload engine idle during every read (the "upload_and_index" wait of ``trim_file``): 16 M x 150 bp
    into eight part files 64 -> 107 M reads/s.  (A producer thread that also INDEXES each chunk on the upload stream, so
    that the next upload starts a record count after the last, was measured next: 96 M -- its index kernels and its
    Python run against the caller's -- and is not kept.)

    ``.gz`` / ``.bz2`` / ``.xz`` input (what the reference's xopen opens by extension) is decompressed by one
    read-ahead thread straight into the staging buffer: the decompressor then sets the pace."""

    READ_AHEAD = 3
    RESERVE = 64 << 20                                       # room for the carried-over tail of the previous chunk

    def __init__(self, path, chunk_bytes, backend=None, clock=None):
        import collections
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.be = backend or _lib.get_backend()
        self.clock = clock or StageClock()
        self.chunk_bytes = int(chunk_bytes)
        cap = self.chunk_bytes + self.RESERVE
        pinned = getattr(self.be, "name", "") == "hip"
        self.file = open(path, "rb")
        self.fd = self.file.fileno()
        self.size = os.path.getsize(path)
        self.pos = 0
        self.stream = None
        name = str(path)
        if name.endswith(".gz"):
            import gzip
            self.stream = gzip.open(self.file, "rb")
        elif name.endswith(".bz2"):
            import bz2
            self.stream = bz2.open(self.file, "rb")
        elif name.endswith(".xz"):
            import lzma
            self.stream = lzma.open(self.file, "rb")
        self.buf = [_staging(cap + 32, pinned) for _ in range(2 if self.stream is not None else self.READ_AHEAD + 1)]
        self.readers = ThreadPoolExecutor(IO_THREADS)
        self.ahead = ThreadPoolExecutor(1)
        self.k = 0
        # the host -> device copy of a chunk runs on its own stream: it overlaps the GPU work on the chunk before
        self.upload_stream = torch.cuda.Stream(device=self.be.device) if pinned else None
        self.host = None
        self.nbytes = 0
        self.final = False
        if self.stream is not None:
            self.pending = self.ahead.submit(self._fill, 0, b"")
        else:
            self.reads = collections.deque()                 # (slot, pread jobs, bytes asked for, last chunk of the file)
            self.next_slot = 0
            self.read_done = False
            self.carry_est = 0
            self.uploaded = [None] * len(self.buf)           # per staging buffer: the event behind its last upload
            for _ in range(self.READ_AHEAD):
                self._issue_read()
            self.pending = self._assemble(b"")

    # ---- plain files: reads ahead of the carry
    def _issue_read(self):
        import os
        if self.read_done:
            return
        slot = self.next_slot
        self.next_slot = (slot + 1) % len(self.buf)
        if self.uploaded[slot] is not None:
            self.uploaded[slot].synchronize()                # (three chunks ago: long gone)
            self.uploaded[slot] = None
        # a chunk is carry + new bytes ~ chunk_bytes in all: in a paired run the file with the smaller records
        # carries its surplus records over every time, and reading a full chunk on top of it would let that
        # surplus grow without bound (1 % of a chunk per step for records 1 % apart).  The carry of the chunk
        # this read will follow is not known yet; the last one seen stands in for it (it drifts slowly).
        want = max(self.chunk_bytes - self.carry_est, self.chunk_bytes // 4)
        want = min(want, self.size - self.pos)
        view = memoryview(self.buf[slot].numpy())
        step = ((want + IO_THREADS - 1) // IO_THREADS + 4095) & ~4095
        jobs = []
        for t in range(IO_THREADS):
            lo, hi = t * step, min(want, (t + 1) * step)
            if hi > lo:
                jobs.append(self.readers.submit(os.preadv, self.fd, [view[self.RESERVE + lo:self.RESERVE + hi]], self.pos + lo))
        self.pos += want
        self.read_done = self.pos >= self.size
        self.reads.append((slot, jobs, want, self.read_done))

    def _assemble(self, carry):
        """The next chunk: its carried-over head in front of the bytes read for it, and its upload."""
        import time
        t0 = time.perf_counter()
        if self.reads:
            slot, jobs, want, final = self.reads.popleft()
            got = sum(j.result() for j in jobs)
            if got != want:
                raise IOError("short read: the input file changed while it was being read")
        else:                                                 # the file is read; records carried over are left
            slot, got, final = self.next_slot, 0, True
            self.next_slot = (slot + 1) % len(self.buf)
            if self.uploaded[slot] is not None:
                self.uploaded[slot].synchronize()
                self.uploaded[slot] = None
        self.clock.add("wait_file_read", t0)
        n0 = len(carry)
        if n0 > self.RESERVE:
            raise ValueError("%d bytes carried over from one chunk to the next (a FASTQ record, or the surplus records of "
                             "one file of a pair, of more than %d bytes)" % (n0, self.RESERVE))
        start = self.RESERVE - n0
        host = self.buf[slot][start:]
        if n0:
            memoryview(host.numpy())[:n0] = carry
        self.carry_est = n0
        nbytes = n0 + got
        unterminated = bool(final and nbytes and int(host[nbytes - 1]) not in (10, 13))
        if unterminated:
            host[nbytes] = 10                                 # tolerate a missing last newline (_seqio.pyx:240-243)
            nbytes += 1
        data = ready = None
        if self.upload_stream is not None:
            with torch.cuda.device(self.be.device), torch.cuda.stream(self.upload_stream):
                data = torch.empty(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=self.be.device)
                data[:nbytes].copy_(host[:nbytes], non_blocking=True)
                data[nbytes:].zero_()
                ready = torch.cuda.Event()
                ready.record()
            self.uploaded[slot] = ready
        self._issue_read()                                    # the buffer of the chunk before this one is free again
        return nbytes, final, unterminated, data, ready, host

    # ---- compressed input: one sequential decompressor, one chunk ahead
    def _fill(self, k, carry):
        view = memoryview(self.buf[k].numpy())
        n0 = len(carry)
        if n0:
            view[:n0] = carry
        room = self.buf[k].numel() - 32 - n0
        want = max(self.chunk_bytes - n0, min(self.chunk_bytes // 4, room))
        if want <= 0:
            raise ValueError("FASTQ record of more than %d bytes" % self.buf[k].numel())
        want, got = min(want, room), 0
        while got < want:
            n = self.stream.readinto(view[n0 + got:n0 + want])
            if not n:
                break
            got += n
        nbytes, final = n0 + got, got < want
        host = self.buf[k]
        unterminated = bool(final and nbytes and int(host[nbytes - 1]) not in (10, 13))
        if unterminated:
            host[nbytes] = 10                                 # tolerate a missing last newline (_seqio.pyx:240-243)
            nbytes += 1
        data = ready = None
        if self.upload_stream is not None:
            with torch.cuda.device(self.be.device), torch.cuda.stream(self.upload_stream):
                data = torch.empty(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=self.be.device)
                data[:nbytes].copy_(host[:nbytes], non_blocking=True)
                data[nbytes:].zero_()
                ready = torch.cuda.Event()
                ready.record()
        return nbytes, final, unterminated, data, ready, host

    def next_batch(self):
        """Upload and index the next chunk; returns the FastqBatch of its whole records."""
        import time
        t0 = time.perf_counter()
        res = self.pending.result() if hasattr(self.pending, "result") else self.pending
        self.nbytes, self.final, unterminated, data, ready, self.host = res
        if self.stream is not None:
            self.clock.add("wait_file_read", t0)
        t0 = time.perf_counter()
        host = self.host
        nbytes = self.nbytes
        if data is None:
            data = self.be.empty(((nbytes + 15) // 16 * 16 + 16,), torch.uint8)
            data[:nbytes].copy_(host[:nbytes], non_blocking=True)
            data[nbytes:].zero_()
        else:
            with torch.cuda.device(self.be.device):
                torch.cuda.current_stream().wait_event(ready)
                data.record_stream(torch.cuda.current_stream())
        batch, self.consumed = FastqBatch.from_device(data, nbytes, self.final, self.be, unterminated=unterminated)
        self.clock.add("upload_and_index", t0)
        return batch

    def advance(self, consumed=None):
        """The caller took ``consumed`` bytes of the current chunk (default: all whole records);
        the rest is carried over and the next chunk is put together (plain files: its bytes are in the staging
        buffer already, its upload starts here).  Returns True when the file is exhausted and nothing is
        carried over."""
        consumed = self.consumed if consumed is None else consumed
        carry = bytes(self.host[consumed:self.nbytes].numpy().tobytes())
        if self.final and not carry:
            return True
        if self.stream is not None:
            self.k = 1 - self.k
            self.pending = self.ahead.submit(self._fill, self.k, carry)
        else:
            self.pending = self._assemble(carry)
        return False

    def close(self):
        if self.stream is None:
            for _, jobs, _, _ in self.reads:                  # reads still in flight own their buffers
                for j in jobs:
                    j.result()
            for ev in self.uploaded:
                if ev is not None:
                    ev.synchronize()
        self.readers.shutdown()
        self.ahead.shutdown()
        self.file.close()
        _release(self.buf)
        self.buf = []


class FastqSink(object):
    """Writes device text to a file through three page-locked buffers.  ``write`` only queues: the
    device -> host copy runs on its own stream behind an event (the caller's stream goes on with the
    next chunk) and a writer thread puts the buffer into the file once the copy has landed -- in
    order, one ``pwrite`` at a time (writes to one file are serialised by the kernel anyway).

    Every ``pwrite`` covers whole 4 KiB blocks of a block-aligned staging buffer at a block-aligned
    offset; the odd bytes at the end of a chunk are carried to the front of the next buffer (the next
    device -> host copy lands right behind them) and the last few go out in ``close``.  That makes
    ``direct=True`` (O_DIRECT, page-cache bypass) possible where the file system allows it -- off by
    default: a lone 1 GB ``pwrite`` into a new file runs at 14 GB/s that way against 6-7 GB/s through the
    page cache on the measured host, but inside the pipeline the synchronous direct writes were no faster
    than the buffered ones (110 vs 100 ms per 4 M reads) and slower than overwriting in place.
    ``keep``: do not truncate an existing file first (overwriting cached pages is about twice as fast as
    allocating new ones); the file is cut to the written length at the end."""

    BLOCK = 4096

    WRITERS = 1                                               # threads a buffer's pwrite is split over (set_writers)

    @classmethod
    def set_writers(cls, n):
        """Split every buffer's ``pwrite`` into n block-aligned ranges written by n threads (one file, disjoint
        ranges: pays where the file system does not serialise writers of one file -- tools/micro/write_parts.py)."""
        cls.WRITERS = max(1, int(n))

    def __init__(self, path, capacity, backend=None, clock=None, keep=False, direct=False, nbuf=3):
        import os
        from concurrent.futures import ThreadPoolExecutor
        be = backend or _lib.get_backend()
        self.clock = clock or StageClock()
        self.gpu = getattr(be, "name", "") == "hip"
        self.path = path
        self.nbuf = max(2, int(nbuf))
        self.buf = [_staging(capacity + self.BLOCK, self.gpu) for _ in range(self.nbuf)]
        flags = os.O_WRONLY | os.O_CREAT | (0 if keep else os.O_TRUNC)
        self.direct = False
        self.fd = -1
        if direct and hasattr(os, "O_DIRECT") and all(t.data_ptr() % self.BLOCK == 0 for t in self.buf):
            try:
                self.fd = os.open(path, flags | os.O_DIRECT, 0o644)
                self.direct = True
            except OSError:
                self.fd = -1
        if self.fd < 0:
            self.fd = os.open(path, flags, 0o644)
        self.offset = 0                                       # bytes handed to pwrite so far (a multiple of BLOCK)
        self.rem = 0                                          # carried bytes at the front of the next buffer
        self.pool = ThreadPoolExecutor(1)
        self.writers = int(self.WRITERS)                      # (fixed for this sink's life: set_writers() applies to later sinks)
        self.wpool = ThreadPoolExecutor(self.writers - 1) if self.writers > 1 else None
        self.pending = [None] * self.nbuf
        self.k = 0
        self.copy_stream = torch.cuda.Stream(device=be.device) if self.gpu else None

    def _put(self, k, rem, n, offset, done, text):
        import os
        if done is not None:
            done.synchronize()                                # the device -> host copy has landed
        del text
        host = self.buf[k]
        total = rem + n
        whole = total - total % self.BLOCK
        view = memoryview(host.numpy())
        at = 0
        if self.wpool is not None and not self.direct and whole >= (16 << 20):
            # disjoint block-aligned ranges of the same buffer and file, one thread each
            per = (whole // self.writers) // self.BLOCK * self.BLOCK

            def piece(lo, hi):
                while lo < hi:
                    lo += os.pwrite(self.fd, view[lo:hi], offset + lo)
            jobs = [self.wpool.submit(piece, i * per, (i + 1) * per) for i in range(1, self.writers - 1)]
            jobs.append(self.wpool.submit(piece, (self.writers - 1) * per, whole))
            piece(0, per)
            for job in jobs:
                job.result()
            at = whole
        while at < whole:
            try:
                at += os.pwrite(self.fd, view[at:whole], offset + at)
            except OSError:
                if not self.direct:
                    raise
                os.close(self.fd)                             # the file system refused the direct write: go on buffered
                self.fd = os.open(self.path, os.O_WRONLY)
                self.direct = False
        if total > whole:                                     # the odd bytes travel to the front of the next buffer
            self.buf[(k + 1) % self.nbuf][:total - whole].copy_(host[whole:total])

    def write(self, text):
        import time
        n = int(text.numel())
        room = self.buf[self.k].numel() - self.BLOCK           # (a staging buffer also holds up to BLOCK - 1 carried bytes)
        if n > room:
            # text that outgrew its chunk (read-name prefixes / suffixes / length tags on short records): in pieces
            for lo in range(0, n, room):
                self.write(text[lo:lo + room])
            return
        t0 = time.perf_counter()
        if self.pending[self.k] is not None:
            self.pending[self.k].result()                     # this buffer is free again after its write
        self.clock.add("wait_file_write", t0)
        rem = self.rem
        if rem + n > self.buf[self.k].numel():
            raise ValueError("FastqSink: a chunk of %d bytes does not fit the staging buffers" % n)
        host = self.buf[self.k]
        done = None
        if self.gpu and n:
            ready = torch.cuda.Event()
            ready.record()                                    # the text is complete on the caller's stream here
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                host[rem:rem + n].copy_(text, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
            text.record_stream(self.copy_stream)
        elif n:
            host[rem:rem + n].copy_(text)
        self.pending[self.k] = self.pool.submit(self._put, self.k, rem, n, self.offset, done, text)
        total = rem + n
        self.offset += total - total % self.BLOCK
        self.rem = total % self.BLOCK
        self.k = (self.k + 1) % self.nbuf

    def close(self):
        import os
        import time
        t0 = time.perf_counter()
        try:
            for job in self.pending:
                if job is not None:
                    job.result()
            if self.rem:                                      # the last odd bytes: an ordinary write
                tail = bytes(self.buf[self.k][:self.rem].numpy().tobytes())
                fd = os.open(self.path, os.O_WRONLY) if self.direct else self.fd
                try:
                    os.pwrite(fd, tail, self.offset)
                finally:
                    if self.direct:
                        os.close(fd)
        finally:
            self.clock.add("wait_file_write", t0)
            self.pool.shutdown()
            if self.wpool is not None:
                self.wpool.shutdown()
            os.ftruncate(self.fd, self.offset + self.rem)
            os.close(self.fd)
            _release(self.buf)
            self.buf = []


class PartSink(object):
    """The output as N part files ``<path>.part0 .. <path>.part<N-1>``, one ``FastqSink`` -- writer thread, staging
    buffers, file -- each: chunk k of the stream goes to part k mod N, records keep their order inside a part.
    Ordering contract: the parts concatenated are NOT the input order; part i holds the chunks i, i + N, i + 2N, ...
    of the stream in that order (an empty chunk is written as nothing and still takes its turn).
    For hosts whose file system serialises the writers of ONE file (measured on the MI355X box: 11.8 GB/s into a
    fresh file from one thread or from eight, 75 GB/s into eight files; putting the parts together again by
    copy_file_range costs as much as writing them, tools/micro/write_parts.py).  What the reference's
    ``--no-writer-process`` does with its worker processes: every worker writes a file of its own."""

    def __init__(self, path, parts, capacity, backend=None, clock=None, keep=False):
        # parts of an earlier run with more of them must not survive next to this run's (`out.part*` would mix them in)
        import glob
        import re
        for old in glob.glob(glob.escape(str(path)) + ".part*"):
            tail = old[len(str(path)) + 5:]
            if re.fullmatch(r"[0-9]+", tail) and int(tail) >= int(parts):
                os.unlink(old)
        self.paths = ["%s.part%d" % (path, i) for i in range(int(parts))]
        self.sinks = [FastqSink(p, capacity, backend, clock, keep=keep, nbuf=2) for p in self.paths]
        self.k = 0

    def write(self, text):
        self.sinks[self.k % len(self.sinks)].write(text)
        self.k += 1

    def close(self):
        first = None
        for s in self.sinks:
            try:
                s.close()
            except Exception as exc:                          # close every part, report the first failure
                first = first or exc
        if first is not None:
            raise first


class CompressedSink(object):
    """The output through a host compressor (``.gz`` / ``.bz2`` / ``.xz`` by extension, as the reference's xopen):
    a writer thread takes the chunks in order, so that the GPU goes on with the next chunk while one is compressed.
    One host thread compresses -- it sets the pace of the whole run (pigz-style parallel blocks: the caller's pipe)."""

    def __init__(self, path, clock=None):
        from concurrent.futures import ThreadPoolExecutor
        name = str(path)
        if name.endswith(".gz"):
            import gzip
            self.fh = gzip.open(name, "wb", compresslevel=6)
        elif name.endswith(".bz2"):
            import bz2
            self.fh = bz2.open(name, "wb")
        else:
            import lzma
            self.fh = lzma.open(name, "wb")
        self.clock = clock or StageClock()
        self.writer = ThreadPoolExecutor(1)
        self.pending = []

    def write(self, text):
        import time
        t0 = time.perf_counter()
        while len(self.pending) > 2:                          # (bounded: at most three chunks in flight)
            self.pending.pop(0).result()
        self.clock.add("wait_file_write", t0)
        host = text.cpu() if torch.is_tensor(text) else text  # the device -> host copy, then the compressor's turn
        self.pending.append(self.writer.submit(lambda h: self.fh.write(bytes(h.numpy().tobytes()) if torch.is_tensor(h) else h), host))

    def close(self):
        try:
            for job in self.pending:
                job.result()
        finally:
            self.writer.shutdown()
            self.fh.close()


def open_by_extension(path):
    """A binary file object for writing; ``.gz`` / ``.bz2`` / ``.xz`` compress (the reference's xopen does so for EVERY
    output, the side files -- too-short, untrimmed, info, rest, wildcard -- included)."""
    name = str(path)
    if name.endswith(".gz"):
        import gzip
        return gzip.open(name, "wb", compresslevel=6)
    if name.endswith(".bz2"):
        import bz2
        return bz2.open(name, "wb")
    if name.endswith(".xz"):
        import lzma
        return lzma.open(name, "wb")
    return open(name, "wb")


def make_sink(path, parts, capacity, backend=None, clock=None, keep=False):
    """One file, or ``parts`` > 1 part files (PartSink); a compressed file by its extension (CompressedSink)."""
    if str(path).endswith((".gz", ".bz2", ".xz")):
        if parts and int(parts) > 1:
            raise ValueError("part files of a compressed output are not provided")
        return CompressedSink(path, clock)
    if parts and int(parts) > 1:
        return PartSink(path, parts, capacity, backend, clock, keep)
    return FastqSink(path, capacity, backend, clock, keep=keep)
