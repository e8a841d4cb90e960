"""The N > 1 path on CPU: world_size-2 gloo process group, one emulated 'GPU' per rank,
contiguous shards, host-side gather -- result must equal the single-process run."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from atropos_amd import _lib, shard, synth
    from atropos_amd.align import Aligner
    from tests.emu.backend import EmuBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.set_backend(EmuBackend())
    w = synth.workload("C2", 0, total)
    al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    local, gathered = shard.sharded_locate(al, w["reads"])
    lo, hi = shard.shard_range(total, rank, world)
    assert len(local) == hi - lo
    if rank == 0:
        torch.save(gathered, out_path)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 1000), (2, 1001), (3, 130)])
def test_sharded_locate_matches_single_process(tmp_path, emu_backend, world, total):
    from atropos_amd import synth
    from atropos_amd.align import Aligner
    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(world, _free_port(), total, out_path), nprocs=world, join=True)
    gathered = torch.load(out_path)
    w = synth.workload("C2", 0, total)
    al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    single = al.locate_batch(w["reads"]).records
    assert gathered.shape == single.shape
    assert torch.equal(gathered, single)


def test_shard_range_partition():
    from atropos_amd.shard import shard_range
    for total in (0, 1, 7, 64, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo <= hi for lo, hi in spans)
