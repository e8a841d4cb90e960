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
