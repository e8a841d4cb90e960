"""Multi-GPU sharding of read batches.

Reads (or pairs) are independent -- the reference's own parallelism is process fan-out
over batches (atropos/commands/multicore.py:297-401) -- so the node-level scheme is: one
process per GPU, contiguous shards in input order, aligner parameters replicated, NO
data-path collective.  Each rank's result records stay on its GPU; when the caller wants
them in one place they are copied to host memory and gathered over a host-side (gloo)
group in shard order, which keeps output order == input order (the reference's
``--preserve-order``).
"""
import torch


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of ``total`` items owned by ``rank``: ceil(total/world) each,
    the tail ranks may be short or empty."""
    per = -(-total // world) if world > 0 else total
    lo = min(total, per * rank)
    return lo, min(total, lo + per)


def local_shard(tensor, rank, world):
    lo, hi = shard_range(tensor.shape[0], rank, world)
    return tensor[lo:hi]


def gather_records_on_host(local_records, total, host_group=None, dst=0):
    """Copy this rank's result records to the host and gather all shards at ``dst`` over a
    host-side process group (gloo).  Returns the concatenated CPU tensor [total, ...] on
    ``dst`` and None elsewhere.  Without an initialised process group: just the CPU copy."""
    import torch.distributed as dist
    rec = local_records.detach().to("cpu").contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(host_group) == 1:
        return rec
    world = dist.get_world_size(host_group)
    rank = dist.get_rank(host_group)
    per = -(-total // world)
    # equal-sized slots (the last shards are padded), then trimmed in shard order
    slot = torch.zeros((per,) + tuple(rec.shape[1:]), dtype=rec.dtype)
    slot[:rec.shape[0]] = rec
    wire = slot.view(torch.uint8).reshape(-1)                     # gloo has no int16: ship raw bytes
    bucket = [torch.zeros_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bucket, dst=dst, group=host_group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(bucket[r].view(rec.dtype).reshape(slot.shape)[:hi - lo])
    return torch.cat(parts, dim=0)


def sharded_locate(aligner, reads, host_group=None):
    """Run ``aligner.locate_batch`` on this rank's contiguous shard of ``reads`` (a uint8
    [n, width] ASCII tensor every rank can index, or a per-rank generator result) and
    gather the records on the host at rank 0.  Returns (local LocateResult, gathered CPU
    tensor or None)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(host_group), dist.get_world_size(host_group)
    else:
        rank, world = 0, 1
    total = reads.shape[0]
    local = aligner.locate_batch(local_shard(reads, rank, world))
    return local, gather_records_on_host(local.records, total, host_group)


def sharded_locate_stream(aligner, reads, sub_batch=10_000_000, host_group=None, depth=2):
    """``sharded_locate`` for shards of many millions of reads: the rank's shard goes through
    ``Aligner.locate_stream`` in sub-batches of ``sub_batch`` reads -- packed one ahead, issued on ``depth``
    streams in turn, so that the exact DP of a sub-batch finishes under the pre-pass of the next -- and the
    records are concatenated in input order.  Returns (records int16 [shard, 8] on the device, gathered CPU
    tensor at rank 0 or None)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(host_group), dist.get_world_size(host_group)
    else:
        rank, world = 0, 1
    total = reads.shape[0]
    mine = local_shard(reads, rank, world)
    parts = (mine[lo:lo + sub_batch] for lo in range(0, mine.shape[0], sub_batch))
    recs = [r.records for r in aligner.locate_stream(parts, depth=depth)]
    local = torch.cat(recs, dim=0) if recs else torch.empty((0, 8), dtype=torch.int16, device=mine.device)
    return local, gather_records_on_host(local, total, host_group)


def device_backends(devices=None):
    """One HipBackend per GPU of this process (all visible ones by default)."""
    from . import _lib
    if devices is None:
        devices = range(torch.cuda.device_count())
    return [_lib.HipBackend(d) for d in devices]


def sharded_run_threads(backends, total, work, out_shape, out_dtype=torch.int16):
    """Single-process multi-device driver: one host thread and one stream per
    device, contiguous shards in input order, NO collective -- every device's records are copied
    into ONE host buffer (page-locked when a GPU is present) at its shard's offset.
    ``work(lo, hi)`` runs in the device's thread, inside ``_lib.thread_backend(backends[d])``, and
    returns that shard's record tensor.  Returns (host tensor [total, *out_shape], seconds per device)."""
    import contextlib
    import threading
    import time
    from . import _lib
    world = len(backends)
    out = torch.empty((total,) + tuple(out_shape), dtype=out_dtype)
    if torch.cuda.is_available():
        out = out.pin_memory()
    seconds, errors = [0.0] * world, []

    def run(d):
        try:
            be = backends[d]
            lo, hi = shard_range(total, d, world)
            t0 = time.perf_counter()
            ctx = be.worker_context() if hasattr(be, "worker_context") else contextlib.nullcontext()
            with _lib.thread_backend(be), ctx:
                if hi > lo:
                    out[lo:hi].copy_(work(lo, hi), non_blocking=True)
            seconds[d] = time.perf_counter() - t0
        except BaseException as err:                      # noqa: BLE001 -- re-raised in the caller's thread
            errors.append((d, err))

    threads = [threading.Thread(target=run, args=(d,), name="atropos-dev%d" % d) for d in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    return out, seconds


def sharded_locate_threads(make_aligner, reads, backends):
    """``Aligner.locate_batch`` over all GPUs of this process: ``make_aligner()`` builds the aligner
    (called once per device, in that device's thread), ``reads`` is a host uint8 [n, width] ASCII
    matrix (or anything ``locate_batch`` takes that can be sliced).  Returns (records int16 [n, 8] on
    the host, in input order; seconds per device)."""
    def work(lo, hi):
        return make_aligner().locate_batch(reads[lo:hi]).records
    return sharded_run_threads(backends, reads.shape[0], work, (8,))


def sharded_linked_threads(make_linked_set, reads, lens, backends):
    """``LinkedSet.match_source`` over all GPUs of this process; ``reads``: host uint8 [n, width]
    upper-case ASCII, ``lens`` int32 [n] or None.  Returns (int16 [n, 3, 8] host tensor: row 0 =
    (which, count, 0, ...), row 1 the 5' record, row 2 the 3' record; seconds per device)."""
    from .adapters import AsciiSource

    def work(lo, hi):
        from . import _lib
        dev = _lib.get_backend().device
        which, count, front, back = make_linked_set().match_source(
            AsciiSource(reads[lo:hi].to(dev), None if lens is None else lens[lo:hi].to(dev)))
        head = torch.zeros_like(front)
        head[:, 0], head[:, 1] = which.to(torch.int16), count.to(torch.int16)
        return torch.stack([head, front, back], dim=1)
    return sharded_run_threads(backends, reads.shape[0], work, (3, 8))


# ---------------------------------------------------------------------------------------------
# sharding a FASTQ file: every rank trims its own byte range of the file, no exchange step
def fastq_record_start(path, offset, probe=1 << 20):
    """The first record boundary at or after byte ``offset`` of a (single-line) FASTQ file: a line
    that starts with '@', whose line after next starts with '+' and whose sequence and quality
    lines have the same length -- a quality line may itself start with '@', the record shape
    does not."""
    import os
    size = os.path.getsize(path)
    if offset <= 0:
        return 0
    if offset >= size:
        return size
    with open(path, "rb") as fh:
        fh.seek(offset - 1)
        buf = fh.read(probe + 1)
    # line starts inside buf (a line starts after every line end; offset itself only if the byte before is one)
    starts = [i + 1 for i in range(len(buf) - 1) if buf[i:i + 1] == b"\n" or
              (buf[i:i + 1] == b"\r" and buf[i + 1:i + 2] != b"\n")]
    lines = []
    for a, b in zip(starts, starts[1:]):
        lines.append((a, buf[a:b].rstrip(b"\r\n")))
    for k in range(len(lines) - 3):
        a, l0 = lines[k]
        if l0.startswith(b"@") and lines[k + 2][1].startswith(b"+") and len(lines[k + 1][1]) == len(lines[k + 3][1]):
            # (a quality line that starts with '@' fails here: two lines on comes a sequence line, not '+')
            return offset - 1 + a
    return size


def fastq_shard_ranges(path, world):
    """[lo, hi) byte ranges of ``world`` contiguous shards of whole records."""
    import os
    size = os.path.getsize(path)
    cuts = [fastq_record_start(path, size * r // world) for r in range(world)] + [size]
    # A probe that finds no record boundary (very long or multi-line records) answers `size`: such a
    # shard is empty and its bytes belong to the shard before it.  Cuts must never run backwards, or
    # records would be emitted twice.
    for r in range(world - 1, -1, -1):
        cuts[r] = min(cuts[r], cuts[r + 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def sharded_trim_file(pipeline, path_in, path_out, rank=None, world=None, chunk_bytes=256 << 20):
    """Every rank runs ``pipeline`` (atropos_amd.trim.TrimPipeline) over its own shard of
    ``path_in`` and writes ``path_out + ".part%d" % rank``; the parts concatenated in rank order are
    the output of a single-process run.  Returns this rank's destination counts."""
    import torch.distributed as dist
    from . import _lib
    from .fastq import FastqBatch
    if rank is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = fastq_shard_ranges(path_in, world)[rank]
    totals = None
    with open(path_in, "rb") as fin, open("%s.part%d" % (path_out, rank), "wb") as fout:
        fin.seek(lo)
        left, carry = hi - lo, b""
        while True:
            block = fin.read(min(chunk_bytes, left))
            left -= len(block)
            final = left == 0
            batch, consumed = FastqBatch.from_bytes(carry + block, final=final)
            carry = b"" if final else (carry + block)[consumed:]
            res = pipeline.run(batch)
            fout.write(res.text(_lib.DEST_KEEP))
            counts = res.counts()
            totals = counts if totals is None else {k: totals[k] + v for k, v in counts.items()}
            if final:
                break
    return totals

