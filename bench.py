#!/usr/bin/env python3
"""Headline benchmark: reads/s of 150 bp single-end adapter alignment
(BASELINE.json configs[1], "C2": 10 M synthetic 150 bp reads, one TruSeq 3' adapter,
e = 0.1) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (atr_locate_batch) over one batch of 10 M packed
reads resident in HBM; with N GPUs every rank owns its own 10 M-read shard (weak
scaling, no data-path collective: reads are independent, results stay per GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ = 75 + 16      # ceil(150/2) packed-nibble bytes in + 16-byte result out
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E spec (MI355X_MICROARCH.md)


def measured_traffic(reads, filtered):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, collected in their
    own rocprofv3 passes by tools/profile_r.sh and reduced into profiles/hbm_traffic.json with
    the gfx950 correction of MI355X_MICROARCH.md), scaled to this launch's read count."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not filtered or not os.path.exists(path):
        return None
    with open(path) as fh:
        t = json.load(fh)
    return t["hbm_bytes_per_read"] * reads


VALU_PEAK_T = 39.3                 # T lane-ops/s at 4 issue cycles per wave64 op: 1024 SIMDs x 64 lanes x 2.4 GHz / 4
                                   # (tools/micro, profiles/r13_valu_issue_rates.txt: the rate of mixed integer streams)


def measured_valu(reads, filtered, kernel_ms):
    """The issue-bound view: VALU wave-instructions of one call (SQ_INSTS_VALU pass of tools/profile_r.sh,
    reduced into profiles/hbm_traffic.json) over the live kernel time of THIS run."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not filtered or not os.path.exists(path):
        return None
    with open(path) as fh:
        t = json.load(fh)
    if "valu_wave_insts_per_launch" not in t:
        return None
    insts = t["valu_wave_insts_per_launch"] * reads / t["reads_per_launch"]
    ach = insts * 64 / (kernel_ms * 1e-3) / 1e12
    return {"wave_insts_per_launch": insts, "achieved": ach, "peak": VALU_PEAK_T, "unit": "T lane-ops/s",
            "frac": ach / VALU_PEAK_T}


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity capped by the
    cgroup CPU quota (the GPU boxes expose 256 logical CPUs but grant a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(workload, sample_reads):
    """The oracle ("port" of the reference's Cython loop: same one-column DP with the
    Ukkonen cut-off, -O2) timed on this box's host cores, all of them, on a bounded
    sample of the same workload: the sample is aligned repeatedly until about 8 s of
    wall time (so thread start-up does not dominate on a many-core host)."""
    from oracle import oracle as O
    cores = usable_cores()
    lens = np.full(len(sample_reads), sample_reads.shape[1], np.int32)
    args = (workload["max_error_rate"], 14, False, False, workload["min_overlap"], workload["indel_cost"], cores)
    O.locate_many(workload["adapter"], sample_reads[:4096], lens[:4096], *args)
    t0 = time.perf_counter()
    O.locate_many(workload["adapter"], sample_reads, lens, *args)
    one = time.perf_counter() - t0
    reps = int(max(1, min(200, 8.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.locate_many(workload["adapter"], sample_reads, lens, *args)
    dt = time.perf_counter() - t0
    return {"value": reps * len(sample_reads) / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the same C2 batch x %d passes, oracle/align_oracle.c on %d threads, "
                      "%.1f s wall" % (len(sample_reads), reps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (C2: 10 M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-sweep", action="store_true",
                    help="time the unfiltered full-column DP kernel instead of the filtered pipeline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from atropos_amd import _lib, synth
    from atropos_amd.align import Aligner
    _lib.set_backend(_lib.HipBackend(local_rank))

    # this rank's shard of the synthetic read set: reads [rank*R, (rank+1)*R)
    w = synth.workload("C2", rank * args.reads, args.reads, device="cuda:%d" % local_rank)
    al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    batch = al.pack(w["reads"])
    sample = w["reads"][:min(args.reads, 2_000_000)].cpu().numpy() if rank == 0 else None
    del w["reads"]
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    filtered = not args.full_sweep
    for _ in range(args.warmup):
        res = al.locate_batch(batch, filtered)
    # timed region: EXACTLY `steps` passes, bracketed by barrier + synchronize; the HIP events
    # sit on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        res = al.locate_batch(batch, filtered)
        ev[s][1].record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    n_found = int(res.found().sum().item())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        total_reads = args.reads * world * args.steps
        achieved = ALGO_BYTES_PER_READ * args.reads / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "reads/s (whole node) 150 bp SE adapter-align",
            "value": total_reads / dt_max, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "C2: %d x 150 bp SE reads per GPU, TruSeq 34-mer 3' adapter, e=0.1, O=3, "
                                   "indel cost 1, 4-bit packed reads resident in HBM" % args.reads,
                       "reads_per_gpu": args.reads, "read_len": 150, "adapter_len": len(w["adapter"]),
                       "parallelism": "shard%d" % world, "matched_fraction": n_found / args.reads},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.reads, filtered),
                         "kernel": ("filter_kernel + scan + scatter + band_kernel + window_kernel<36,eq,indel> (one "
                                    "atr_locate_batch call)" if filtered else "locate_kernel<36,eq,indel>"),
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ,
                         "valu": measured_valu(args.reads, filtered, kernel_ms),
                         "note": "integer-VALU bound, not HBM bound: %.2f G full-matrix cell-equivalents/s"
                                 % (args.reads * 150 * 34 / (kernel_ms * 1e-3) / 1e9)},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, sample)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
