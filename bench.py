#!/usr/bin/env python3
"""Headline benchmark: reads/s of adapter alignment on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3                 # C2, the headline (BASELINE.json configs[1])
    python bench.py --config C3|C4|C5 ...                          # the other BASELINE configs, one GPU shard each
    python bench.py --gpus N ...                                   # launches its own N ranks (one process per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # the same ranks under torchrun
    python bench.py --gpus N --single-process                      # the same shards from one process (threads + streams)

The ranks of a multi-GPU run meet only at a barrier and for the max-over-ranks of the elapsed time; both go over a
gloo (host) process group -- the path has no exchange step, so there is no RCCL anywhere.

A step is one pass of the hot path over one batch of packed reads resident in HBM; with N GPUs
every rank owns its own shard (weak scaling, no data-path collective: reads are independent,
results stay per GPU).  Prints ONE JSON line on rank 0.

  C2  10 M x 150 bp SE, one TruSeq 3' adapter, e = 0.1       atr_locate_batch (filtered pipeline)
  C3  10 M pairs 2 x 150 bp, insert aligner                  atr_insert_match_batch
  C4  12.5 M x 150 bp SE (100 M / 8 GPUs), 4 linked adapters atr_linked_match_batch
  C5  2 x 250 bp pairs (100 M / 8 GPUs, streamed as sub-batches), insert aligner with read wildcards
      + error correction of the overlap (liberal)            atr_insert_match_batch + atr_insert_correct_batch
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

HBM_PEAK_GBS = 8000.0              # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0          # float4 device copy measured on MI355X (same guide)
VALU_NOMINAL_T = 78.6              # T lane-ops/s: 1024 SIMD-32s x 32 lanes x 2.4 GHz (wave64 op = 2 issue cycles)
VALU_MIXED_MODEL_T = 39.3          # every op of a mixed integer stream at 4 issue cycles (profiles/r13_valu_issue_rates.txt): a MODEL
PCIE_GBS = 63.0


def profile_counters(config):
    """PMC figures of one launch from the committed rocprofv3 passes (tools/profile_r.sh ->
    tools/reduce_traffic.py -> profiles/traffic_<config>.json).  They are NOT measured by this run."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % config)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


STREAMING_KERNELS = ("filter_kernel", "insert_kernel")   # stream the packed batch with 16 bytes per lane (see live_counters)
# kernels that stream the packed batch (16 bytes per lane, counted at half) AND fetch scattered bytes (counted in full):
# raw FETCH_SIZE + the uncounted half of the stream, whose size is known -- bytes per unit.
#   insert_correct_kernel   C5's fused kernel: 2 reads x 8 chunks x 16 B per pair, half of it
#   atr_piece_spec / piece_filter_kernel   C2's pre-pass: pass A streams 5 chunks x 16 B per read (half of it: 40), pass B
#       gathers three 16-byte pieces per flagged read.  Calibrated (profiles/round5_c2_fetch_calibration.txt): with pass
#       B's gathers compiled out FETCH_SIZE reads 438 MB for the 800 MB stream (+ lists), with them 650 MB -- the 3.3 M
#       extra requests are tallied at 64 B each; 2 x FETCH_SIZE (rounds 4 / 5 until this fix) counted them twice
# The uncounted half is derived from the config's own layout: a read of nchunks 32-base chunks streams nchunks x 16 B, a
# pair 2 x nchunks x 16 B (the round-5 constants 40 / 128 held for 150- and 250-base reads only).
#   linked_filter_kernel   C4's fused kernel (round 6: the same rule as C2 / C5 -- rounds 4 / 5 doubled its whole FETCH_SIZE,
#       gathers of the 5' DP tasks and of the compacted 3' lanes included)
MIXED_KERNELS = {"insert_correct_kernel": 2, "atr_piece_spec": 1, "piece_filter_kernel": 1, "linked_filter_kernel": 1}


def half_stream_bytes(reads_per_unit, read_len):
    """Bytes per unit of a 16-byte-per-lane plane / nibble stream that gfx950's FETCH_SIZE leaves out (half of it)."""
    return reads_per_unit * ((read_len + 31) // 32) * 16 / 2.0


def live_counters(config, reads, want_valu, read_len=150, stream_units=None):
    """HBM traffic (and VALU wave-instructions) of one step of `config`, MEASURED by this invocation: bench.py runs
    itself under `rocprofv3 --pmc <counter>` -- one pass per counter, no trace domains, as MI355X_MICROARCH.md's
    HBM section prescribes -- for three short calls and reduces the per-kernel CSV:  bytes = 2 x FETCH_SIZE +
    WRITE_SIZE (KiB) for the kernels that stream the packed batch with 16-byte loads per lane (gfx950 FETCH_SIZE
    counts half of such a stream), raw FETCH_SIZE + WRITE_SIZE for the others; the pack kernels of the set-up are
    left out.  Returns None when rocprofv3 is not there or a pass fails (the line then falls back to the committed
    profile and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    calls = 3
    totals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE") + (("SQ_INSTS_VALU",) if want_valu else ()):
            out = tempfile.mkdtemp(prefix="atr_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--config", config, "--reads", str(reads), "--steps", str(calls - 1), "--warmup", "1",
                   "--no-cpu-baseline", "--no-secondary", "--no-live-counters"]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            per = {}
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        name = row["Kernel_Name"]
                        if (("atr::" in name or "atr_piece" in name) and "pack" not in name and "front_ascii" not in name
                                and row["Counter_Name"] == counter):
                            per[name] = per.get(name, 0.0) + float(row["Counter_Value"])
            shutil.rmtree(out, ignore_errors=True)
            if not per:
                return None
            totals[counter] = per
    except Exception:                                             # noqa: BLE001 -- rocprofv3 missing / failed: fall back
        return None
    hbm = 0.0
    for name, v in totals["FETCH_SIZE"].items():
        per_unit = next((h for key, h in MIXED_KERNELS.items() if key in name), None)
        if per_unit is not None:
            hbm += v * 1024.0 + half_stream_bytes(per_unit, read_len) * (stream_units or reads) * calls
        else:
            hbm += (2.0 if any(k in name for k in STREAMING_KERNELS) else 1.0) * v * 1024.0
    hbm += sum(totals["WRITE_SIZE"].values()) * 1024.0
    res = {"hbm_bytes_per_launch": hbm / calls, "units_per_launch": reads,
           "source": "this invocation: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %d calls each (gfx950: 2 x FETCH_SIZE for "
                     "the 16-byte-per-lane streaming kernels; FETCH_SIZE + the uncounted half of the known stream for "
                     "kernels that stream and gather)" % calls}
    if want_valu:
        res["valu_wave_insts_per_launch"] = sum(totals["SQ_INSTS_VALU"].values()) / calls
    return res


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


def timed_cpu(fn, units, what, cores, target_s=8.0):
    """Run fn() repeatedly for about target_s of wall time (after one untimed call)."""
    fn()
    t0 = time.perf_counter()
    fn()
    one = time.perf_counter() - t0
    reps = int(max(1, min(200, target_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = time.perf_counter() - t0
    return {"value": reps * units / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%s x %d passes on %d threads, %.1f s wall (oracle/align_oracle.c, the C restatement of the "
                      "reference's Cython loop)" % (what, reps, cores, dt)}


def _c2_two_streams(self, steps=40):
    """Consecutive batches through the product's streaming form of the call, ``Aligner.locate_stream`` (two streams, a
    workspace each: the exact DP of batch i finishes under the pre-pass of batch i + 1), against the same calls on one
    stream.  What a double-buffered caller gets (atropos_amd.shard.sharded_locate_stream); NOT the bench's `value`,
    which is the plain call issued back to back on one stream with every step's records complete inside the timed region."""
    def run(depth):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for res in self.al.locate_stream((self.batch for _ in range(steps)), depth=depth):
            last = res
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, last

    res = {}
    for n in (1, 2):
        run(n)
        best = min(run(n)[0] for _ in range(3))
        res["streams_%d_reads_per_s" % n] = self.n / best
    ref = self.al.locate_batch(self.batch, self.filtered).records
    res["records_equal_one_call"] = bool(torch.equal(run(2)[1].records, ref))
    res["note"] = "Aligner.locate_stream(depth=2): steps alternate between two streams with a workspace each; not the line's value"
    return res


class C2(object):
    name = "C2"
    default_units = 10_000_000
    unit_reads = 1
    algo_bytes = 75 + 16               # ceil(150/2) packed-nibble bytes in + 16-byte result out
    dtype = "int32"
    metric = "reads/s (whole node) 150 bp SE adapter-align"

    def __init__(self, args, rank, dev):
        from atropos_amd import synth
        from atropos_amd.align import Aligner
        self.args, self.n = args, args.reads
        self.w = w = synth.workload("C2", rank * self.n, self.n, device=dev)
        self.al_args = (w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
        self.al = Aligner(*self.al_args)
        self.ascii = w["reads"]
        self.batch = self.al.pack(self.ascii, layout="tile64" if args.full_sweep else "auto")
        # the timed steps ROTATE over resident batches (round-5 verdict, weak 10): no step finds the batch, the records or
        # the workspace lists of the step before it in the 256 MB Infinity Cache.  Batch b = reads [b n, (b + 1) n) of the
        # same generator, shifted by world x n per extra batch so that ranks never share reads.
        self.batches = [self.batch]
        for b in range(1, max(1, args.rotate)):
            wb = synth.workload("C2", (rank + b * max(1, args.gpus)) * self.n, self.n, device=dev)
            self.batches.append(self.al.pack(wb["reads"], layout=self.batch.layout))
            del wb
        self.sample = self.ascii[:min(self.n, 2_000_000)].cpu().numpy() if rank == 0 else None
        self.filtered = not args.full_sweep
        if not args.secondary:
            self.ascii = None
            del w["reads"]

    def step(self, s):
        self.res = self.al.locate_batch(self.batches[s % len(self.batches)], self.filtered)

    def jit(self):
        """Is the pre-pass the timed steps ran the kernel compiled at run time for this aligner (jit.hpp)?"""
        return bool(self.filtered and self.batch.layout == "plane64" and self.al.prepare(150))

    def describe(self):
        w = self.w
        return {"workload": "C2: %d x 150 bp SE reads per GPU, TruSeq 34-mer 3' adapter, e=0.1, O=3, indel cost 1, "
                            "4-bit packed reads (%s layout) resident in HBM, steps rotating over %d resident batches"
                            % (self.n, self.batch.layout, len(self.batches)),
                "reads_per_gpu": self.n, "read_len": 150, "adapter_len": len(w["adapter"]), "resident_batches": len(self.batches),
                "matched_fraction": int(self.res.found().sum().item()) / self.n}

    def kernel(self):
        if not self.filtered:
            return "locate_kernel<36,eq,indel>"
        if self.batch.layout == "plane64":
            return ("%s (pass A: exact pieces on bit planes; pass B: windowed bit-vector sweep of the flagged "
                    "reads) + piece_scatter + band_kernel + window_kernel<36,eq,indel,planes> (one atr_locate_planes_batch call)"
                    % ("atr_piece_spec (hiprtc build of piece_filter.hpp for this aligner)" if self.jit() else
                       "piece_filter_kernel<5> (generic: no specialised kernel on this box)"))
        return "filter_kernel + scan + scatter + band_kernel + window_kernel<36,eq,indel> (one atr_locate_batch call)"

    def note(self, kernel_ms):
        return "integer-VALU bound, not HBM bound: %.2f G full-matrix cell-equivalents/s" % (
            self.n * 150 * 34 / (kernel_ms * 1e-3) / 1e9)

    def cpu_baseline(self):
        from oracle import oracle as O
        cores, w, sample = usable_cores(), self.w, self.sample
        lens = np.full(len(sample), sample.shape[1], np.int32)
        fn = lambda: O.locate_many(w["adapter"], sample, lens, w["max_error_rate"], 14, False, False, w["min_overlap"],
                                   w["indel_cost"], cores)
        return timed_cpu(fn, len(sample), "first %d reads of the same C2 batch" % len(sample), cores)

    two_streams = _c2_two_streams

    def hard_batches(self, run):
        """The same aligner on batches where the pre-pass cannot finish most reads (synth.hard_batch): reads/s of the same
        one-call form, the share of the reads left to the exact DP kernels (atr_locate_work_unresolved), and -- on the
        first 1 M reads -- that the records equal the full sweep's.  The friendly C2 batch beside them for scale."""
        from atropos_amd import _lib, synth
        be = _lib.get_backend()
        dev = self.batch.packed.device
        out = {"note": "10 M x 150 bp each, TruSeq 34-mer, e=0.1; unresolved = reads the pre-pass hands to band_kernel / "
                       "window_kernel; the floor is the full sweep of every column (locate_kernel)"}
        kinds = (("C2", None),) + tuple((k, k) for k in synth.HARD_KINDS)
        for name, kind in kinds:
            reads = self.ascii if kind is None else synth.hard_batch(kind, 0, self.n, device=dev)
            if reads is None:
                continue
            batch = self.al.pack(reads, layout=self.batch.layout)
            ms = run(lambda: self.al.locate_batch(batch, self.filtered))
            rec = self.al.locate_batch(batch, self.filtered).records
            left = be.last_unresolved(self.n) if hasattr(be, "last_unresolved") else None
            k = min(self.n, 1_000_000)
            full = self.al.locate_batch(self.al.pack(reads[:k], layout="tile64"), filtered=False).records
            out[name] = {"reads_per_s": self.n / (ms * 1e-3), "ms": ms,
                         "unresolved_fraction": None if left is None else left / self.n,
                         "matched_fraction": float((rec[:, 1] >= 0).float().mean().item()),
                         "records_equal_full_sweep_first_%d" % k: bool(torch.equal(rec[:k], full))}
            del batch, rec, full
            if kind is not None:
                del reads
        if self.ascii is not None:
            tiles = self.al.pack(self.ascii[:min(self.n, 2_000_000)], layout="tile64")
            ms = run(lambda: self.al.locate_batch(tiles, filtered=False), reps=2)
            out["full_sweep_floor_reads_per_s"] = min(self.n, 2_000_000) / (ms * 1e-3)
        return out

    def secondary(self, kernel_ms):
        """The other figures SURVEY 8(d) lists, measured outside the timed region."""
        out = {"cell_updates_per_s_full_matrix": self.n * 150 * 35 / (kernel_ms * 1e-3)}

        def run(fn, reps=5):
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        if self.ascii is not None:
            ms = run(lambda: self.al.locate_batch(self.al.pack(self.ascii, layout=self.batch.layout), self.filtered))
            out["pack_then_locate_reads_per_s"] = self.n / (ms * 1e-3)      # ASCII in HBM -> records, two calls (pack, locate)
            out["pack_inclusive_reads_per_s"] = out["pack_then_locate_reads_per_s"]
            if self.filtered and self.batch.layout == "plane64" and hasattr(self.al._backend, "locate_ascii_planes_batch"):
                # the fused ASCII entry: rows -> bit planes in registers -> pass A in one kernel (atr_locate_ascii_planes_batch)
                planes_buf = [None]

                def fused():
                    res, left = self.al.locate_ascii(self.ascii) if planes_buf[0] is None else (
                        self.al._backend.locate_ascii_planes_batch(self.al._handle, self.ascii, None, 150, planes_buf[0]))
                    planes_buf[0] = left.packed if hasattr(left, "packed") else left
                    return res
                ms = run(fused)
                out["fused_ascii_reads_per_s"] = self.n / (ms * 1e-3)
                out["pack_inclusive_reads_per_s"] = max(out["pack_inclusive_reads_per_s"], out["fused_ascii_reads_per_s"])
                rec = fused()
                rec = rec.records if hasattr(rec, "records") else rec
                out["fused_ascii_records_equal"] = bool(torch.equal(rec, self.al.locate_batch(self.batch, self.filtered).records))
            # ragged batch: the same reads cut to lengths 100..150 (what quality-trimmed data looks like)
            from atropos_amd.batch import ReadBatch
            g = torch.Generator(device=self.ascii.device).manual_seed(5)
            lens = torch.randint(100, 151, (self.n,), generator=g, device=self.ascii.device, dtype=torch.int32)
            planes = self.filtered and self.al._wants_planes("auto", self.n, 150, ragged=True)     # (the two-pass pre-pass)
            rb = ReadBatch.from_ascii(self.ascii, lens, 150, self.al.table_kind, self.al._table, planes=planes)
            ms = run(lambda: self.al.locate_batch(rb, self.filtered))
            out["ragged_batch_reads_per_s"] = self.n / (ms * 1e-3)
            out["ragged_batch_layout"] = rb.layout
        out["pcie_inclusive_bound_reads_per_s"] = PCIE_GBS * 1e9 / (150 + 16)    # ASCII in + record out over PCIe Gen5 x16
        try:
            out["two_streams"] = self.two_streams()
        except Exception as exc:                                              # noqa: BLE001 -- a side figure
            out["two_streams"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
        try:
            out["hard_batches"] = self.hard_batches(run)
        except Exception as exc:                                              # noqa: BLE001 -- a side figure
            out["hard_batches"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        # the small-batch regime of the drop-in path (tools/bench_small.py): the unchanged trim command hands over
        # <= 1000 reads per call (/root/reference/atropos/commands/base.py:179), the per-read API a batch of one
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_small
            out["small_batches"] = bench_small.measure()
        except Exception as exc:                                              # noqa: BLE001 -- a side figure
            out["small_batches"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        # file -> file: a page-cached FASTQ file through the device-resident trim pipeline into part files
        try:
            out["file_to_file"] = self.file_to_file()
        except Exception as exc:                                              # noqa: BLE001 -- a side figure
            out["file_to_file"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        return out

    def file_to_file(self, nreads=12_000_000, parts=8, chunk_mb=128, runs=2):
        """`atropos trim -a ADAPTER -q 20 --trim-n -m 20` on a FASTQ file of ``nreads`` C2 reads in the page cache
        (/tmp), trimmed text into ``parts`` part files (TrimPipeline.trim_file: threaded reads READ_AHEAD chunks ahead
        into page-locked buffers, upload of chunk i + 1 under the kernels of chunk i, write-behind per part)."""
        import tempfile
        import bench_fastq                                                   # tools/ (on sys.path above)
        from atropos_amd.trim import pipeline_from_args
        data, nbytes = bench_fastq.device_fastq(nreads)
        tmp = tempfile.mkdtemp(prefix="atr_f2f_")
        src, dst = os.path.join(tmp, "in.fastq"), os.path.join(tmp, "out.fastq")
        names = ["%s.part%d" % (dst, i) for i in range(parts)]
        try:
            with open(src, "wb") as fh:
                fh.write(data[:nbytes].cpu().numpy().tobytes())
            del data
            torch.cuda.empty_cache()
            pipe = pipeline_from_args("-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -q 20 --trim-n -m 20")
            pipe.trim_file(src, dst, chunk_bytes=chunk_mb << 20, output_parts=parts)
            total, waits = 0.0, {}
            for _ in range(runs):
                for name in names:
                    os.remove(name)
                t0 = time.perf_counter()
                counts = pipe.trim_file(src, dst, chunk_bytes=chunk_mb << 20, output_parts=parts)
                total += time.perf_counter() - t0
                for k, v in pipe.stage_seconds.items():
                    waits[k] = waits.get(k, 0.0) + v
            out_bytes = sum(os.path.getsize(name) for name in names)
        finally:
            for name in [src] + names:
                if os.path.exists(name):
                    os.remove(name)
            os.rmdir(tmp)
        sec = total / runs
        return {"reads_per_s": nreads / sec, "nreads": nreads, "ms": sec * 1e3, "output_parts": parts, "chunk_mb": chunk_mb,
                "input_bytes": nbytes, "output_bytes": out_bytes, "kept": counts.get("keep"),
                "host_GBps": (nbytes + out_bytes) / sec / 1e9,
                "pcie_floor_reads_per_s": nreads / (nbytes / (PCIE_GBS * 1e9)),
                "wait_ms": {k: round(v / runs * 1e3, 2) for k, v in waits.items()},
                "note": "FASTQ text both ways over PCIe; the upload of %d B per read bounds it at pcie_floor_reads_per_s; "
                        "one output file instead of part files is bound by the kernel's one-writer-per-file rate "
                        "(tools/bench_file_to_file.py)" % (nbytes // nreads)}


class C3(object):
    name = "C3"
    default_units = 10_000_000
    unit_reads = 2
    read_len = 150
    algo_bytes = 2 * 75 + 48           # two packed reads in + three 16-byte records out per pair
    dtype = "u32 bit planes"
    metric = "reads/s (whole node) 2x150 bp PE insert-align"
    kw = {}
    corrects = False

    def __init__(self, args, rank, dev):
        from atropos_amd import synth
        from atropos_amd.align import InsertAligner
        self.args, self.n = args, args.reads
        self.ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, **self.kw)
        self.nbatches = 1
        p1, p2 = [], []
        self.sample = None
        chunk = 2_000_000
        for lo in range(0, self.n, chunk):
            w = synth.workload(self.name, rank * self.n + lo, min(chunk, self.n - lo), device=dev)
            if self.sample is None and rank == 0:
                k = 200_000
                self.sample = (w["reads1"][:k].cpu().numpy(), w["reads2"][:k].cpu().numpy())
            p1.append(w["reads1"])
            p2.append(w["reads2"])
        self.b1 = self.ia.pack(torch.cat(p1))
        self.b2 = self.ia.pack(torch.cat(p2), check=True)

    def step(self, s):
        self.res = self.ia.match_insert_batch(self.b1, self.b2)

    def describe(self):
        return {"workload": "%s: %d pairs 2 x %d bp per GPU, insert aligner (overlap + adapter match)%s, plane64-packed "
                            "reads resident in HBM" % (self.name, self.n, self.read_len,
                                                       ", read wildcards" if self.kw else ""),
                "pairs_per_gpu": self.n, "read_len": self.read_len,
                "matched_fraction": float(self.res.found().float().mean().item())}

    def kernel(self):
        return "insert_kernel (one atr_insert_match_batch call)"

    def note(self, kernel_ms):
        L = self.read_len
        return "integer-VALU bound: %.2f T diagonal cell-equivalents/s" % (self.n * L * (L + 1) / 2 / (kernel_ms * 1e-3) / 1e12)

    def cpu_baseline(self):
        from oracle import oracle as O
        cores = usable_cores()
        orc = O.InsertOracle(self.ia.adapter1, self.ia.adapter2, **self.kw)
        r1, r2 = self.sample
        lens = np.full(len(r1), r1.shape[1], np.int32)
        fn = lambda: O.match_insert_many(orc, r1, lens, r2, lens, cores)
        out = timed_cpu(fn, 2 * len(r1), "first %d pairs of the same %s batch (match_insert only)" % (len(r1), self.name), cores)
        return out

    def secondary(self, kernel_ms):
        return {"pairs_per_s": self.n / (kernel_ms * 1e-3),
                "pcie_inclusive_bound_reads_per_s": 2 * PCIE_GBS * 1e9 / (2 * self.read_len + 48)}


class C5(C3):
    """Insert aligner with read wildcards + liberal error correction of the overlap.  Correction is
    in place and happens once per read pair, so every step gets its OWN sub-batch of the shard
    (warmup + steps sub-batches of --reads pairs are generated and stay resident)."""
    name = "C5"
    default_units = 2_000_000
    read_len = 250
    algo_bytes = 2 * 125 + 48
    metric = "reads/s (whole node) 2x250 bp PE insert-align + overlap error correction"
    kw = dict(read_wildcards=True)
    corrects = True

    def __init__(self, args, rank, dev):
        from atropos_amd import _lib, synth
        from atropos_amd.align import InsertAligner
        from atropos_amd.modifiers import COMP_TABLE
        self.args, self.n = args, args.reads
        self.ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, **self.kw)
        self.be = _lib.get_backend()
        self.comp = COMP_TABLE
        self.fused = os.environ.get("ATR_BENCH_C5_TWO_CALLS", "0") != "1"      # (A/B switch: the two-kernel form of round 4)
        self.nbatches = args.warmup + args.steps
        self.batches = []
        self.sample = None
        for b in range(self.nbatches):
            w = synth.workload("C5", (rank * self.nbatches + b) * self.n, self.n, device=dev)
            if self.sample is None and rank == 0:
                k = 100_000
                self.sample = tuple(w[key][:k].cpu().numpy() for key in ("reads1", "reads2", "quals1", "quals2"))
            self.batches.append(dict(b1=self.ia.pack(w["reads1"]), b2=self.ia.pack(w["reads2"], check=True),
                                     s1=w["reads1"], s2=w["reads2"], q1=w["quals1"], q2=w["quals2"]))
        self.changed = self.be.empty((self.n, 2), torch.int32)
        self.newlen = self.be.empty((self.n, 2), torch.int32)
        self.k = 0

    def step(self, s):
        d = self.batches[self.k]
        self.k += 1
        if self.fused:
            self.res, _, _ = self.ia.match_insert_correct_batch(d["b1"], d["b2"], d["s1"], d["q1"], d["s2"], d["q2"], "liberal", 1,
                                                                self.changed, self.newlen)
        else:
            self.res = self.ia.match_insert_batch(d["b1"], d["b2"])
            self.be.insert_correct_batch(self.res.records, d["s1"], d["q1"], None, d["s2"], d["q2"], None, 2, 1, self.comp,
                                         self.changed, self.newlen, planes1=d["b1"], planes2=d["b2"])

    def describe(self):
        d = C3.describe(self)
        d["workload"] = ("C5: %d pairs 2 x 250 bp per step and GPU (every step its own sub-batch of the 12.5 M-pair "
                         "shard), insert aligner with read wildcards + liberal error correction of the overlap in "
                         "place (ASCII bases + qualities resident in HBM)" % self.n)
        d["corrected_pair_fraction"] = float(((self.changed[:, 0] > 0) | (self.changed[:, 1] > 0)).float().mean().item())
        return d

    def kernel(self):
        if self.fused:
            return "insert_correct_kernel (one atr_insert_match_correct_batch call: match + correction, the planes streamed once)"
        return "insert_kernel + correct_planes_kernel (atr_insert_match_batch + atr_insert_correct_batch)"

    def cpu_baseline(self):
        """match_insert AND the liberal correction of the overlap, as the timed GPU step does."""
        from oracle import oracle as O
        cores = usable_cores()
        orc = O.InsertOracle(self.ia.adapter1, self.ia.adapter2, **self.kw)
        r1, r2, q1, q2 = self.sample
        lens = np.full(len(r1), r1.shape[1], np.int32)

        def fn():
            a1, a2, b1, b2 = r1.copy(), r2.copy(), q1.copy(), q2.copy()      # the correction is in place
            rec = O.match_insert_many(orc, a1, lens, a2, lens, cores)
            O.insert_correct_many(rec, a1, b1, lens, a2, b2, lens, "liberal", 1, cores)
        return timed_cpu(fn, 2 * len(r1), "first %d pairs of the same C5 batch (match_insert + correct_errors, incl. "
                                          "a copy of the four matrices per pass)" % len(r1), cores)


class C4(object):
    name = "C4"
    default_units = 12_500_000
    unit_reads = 1
    dtype = "int32"
    metric = "reads/s (whole node) 150 bp SE, 4 linked adapters"

    def __init__(self, args, rank, dev):
        from atropos_amd import synth
        from atropos_amd.adapters import AsciiSource, LinkedAdapter, LinkedSet, upper_ascii
        self.args, self.n = args, args.reads
        chunks = []
        for lo in range(0, self.n, 2_500_000):
            w = synth.workload("C4", rank * self.n + lo, min(2_500_000, self.n - lo), device=dev)
            chunks.append(w["reads"])
        self.w = w
        reads = upper_ascii(torch.cat(chunks))
        self.sample = reads[:min(self.n, 1_000_000)].cpu().numpy() if rank == 0 else None
        linked = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                                min_overlap=w["min_overlap"], indel_cost=w["indel_cost"])
                  for f, b in zip(w["fronts"], w["backs"])]
        self.lset = LinkedSet(linked)
        if not self.lset.fused:
            raise SystemExit("C4: the linked set is outside the fused pipeline's envelope")
        # Round 6: the resident form is what atr_linked_group_pack leaves -- the anchored 5' parts decided where the ASCII
        # row is first touched, read[front.rstop:] as bit planes in a sub-batch per adapter -- and a step is
        # atr_linked_group_match: the two-pass pre-pass compiled for each 3' aligner on its group, the acceptance test, the
        # 3' records back in batch order.  ATR_BENCH_C4_FUSED=1: round 5's form (whole reads as tile64, 5' and 3' parts in
        # the timed call).  The pack-inclusive figures of both forms are in `secondary`.
        self.grouped = os.environ.get("ATR_BENCH_C4_FUSED", "0") != "1" and self.lset.group_applies(150)
        self.algo_bytes = (75 + 16) if self.grouped else (75 + 16 + 16)   # packed read (<= 75 B) + 3' record (+ 5' record: fused form)
        self.ascii = reads if (args.secondary or self.grouped) else None
        if self.grouped:
            self.groups = self.lset.pack_groups(reads)
            self.stream_units = sum(self.groups.group_reads())            # reads whose planes a step streams
            self.batch = None
            if not args.secondary:
                self.ascii = None
        else:
            self.batch = AsciiSource(reads).batch(self.lset.table_kind, self.lset.table)
        del reads, chunks

    def step(self, s):
        if self.grouped:
            self.res = self.lset.match_groups(self.groups)
            return
        be = self.lset._backend
        wc, front, back = be.linked_match_batch(self.lset._handle, self.batch.packed, self.batch.lens, self.batch.nreads,
                                                self.batch.max_len)
        self.res = (wc[:, 0], wc[:, 1], front, back)

    def describe(self):
        which, count, front, back = self.res
        form = ("the 5' parts decided at pack time (atr_linked_group_pack), read[front.rstop:] as bit planes in one plane64 "
                "sub-batch per adapter resident in HBM; a step = the 3' parts + acceptance test + records in batch order"
                if self.grouped else "4-bit packed whole reads (tile64) resident in HBM; a step = 5' and 3' parts")
        return {"workload": "C4: %d x 150 bp SE reads per GPU (100 M / 8), four linked adapters (anchored 20-mer 5' part "
                            "+ 33/34-mer 3' part), e=0.12, O=3, indel cost 1; %s" % (self.n, form),
                "reads_per_gpu": self.n, "read_len": 150, "resident_form": "grouped plane64" if self.grouped else "tile64",
                "front_matched_fraction": float((which >= 0).float().mean().item()),
                "back_matched_fraction": float((back[:, 1] >= 0).float().mean().item()),
                "reads_with_two_fronts": int((count > 1).sum().item())}

    def kernel(self):
        if self.grouped:
            return ("4 x [atr_piece_spec (ragged, one build per 3' aligner) + piece_scatter + band_kernel || window_kernel] on "
                    "four streams + linked_group_finish_kernel (one atr_linked_group_match call)")
        return ("linked_filter_kernel + linked_scatter + linked_band_kernel || window_kernel<36> (one launch each for the "
                "four adapters; one atr_linked_match_batch call)")

    def note(self, kernel_ms):
        if self.grouped:
            return ("integer-VALU bound; algorithmic bytes = packed read[rstop:] (<= 75 B) + the 3' record: the 5' record is "
                    "written by the pack pass, whose time is in secondary.pack_inclusive_reads_per_s")
        return "integer-VALU bound: 4 anchored 5' sweeps + one 3' pre-pass per read in one kernel"

    def cpu_baseline(self):
        from oracle import oracle as O
        cores, w, sample = usable_cores(), self.w, self.sample
        lens = np.full(len(sample), sample.shape[1], np.int32)
        fn = lambda: O.linked_many(w["fronts"], w["backs"], sample, lens, w["max_error_rate"], w["min_overlap"],
                                   w["indel_cost"], True, False, cores)
        return timed_cpu(fn, len(sample), "first %d reads of the same C4 batch" % len(sample), cores)

    def secondary(self, kernel_ms):
        from atropos_amd.adapters import AsciiSource
        out = {"pcie_inclusive_bound_reads_per_s": PCIE_GBS * 1e9 / (150 + 34)}
        if self.ascii is None:
            return out

        def run(fn, reps=5):
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        be, lset = self.lset._backend, self.lset
        if self.grouped:
            ms_pack = run(lambda: lset.pack_groups(self.ascii))
            ms = run(lambda: lset.match_groups(lset.pack_groups(self.ascii)))
            out["group_pack_ms"] = ms_pack                                  # 5' parts + grouped planes (ASCII in HBM; one host sync)
            out["pack_inclusive_reads_per_s"] = self.n / (ms * 1e-3)         # ASCII in HBM -> both records of every read
            slab_ms = run(lambda: lset.match_groups(self.groups, ordered=False))
            out["slot_order_reads_per_s"] = self.n / (slab_ms * 1e-3)        # records left in slot order (slab + permutation)
            out["group_reads"] = self.groups.group_reads()
        # round 5's form on the same reads: whole reads as tile64, 5' and 3' parts in one call
        fused_batch = AsciiSource(self.ascii).batch(lset.table_kind, lset.table)
        fms = run(lambda: be.linked_match_batch(lset._handle, fused_batch.packed, fused_batch.lens, fused_batch.nreads, fused_batch.max_len))
        out["fused_tile64_reads_per_s"] = self.n / (fms * 1e-3)
        pms = run(lambda: AsciiSource(self.ascii).batch(lset.table_kind, lset.table))
        out["fused_tile64_pack_inclusive_reads_per_s"] = self.n / ((fms + pms) * 1e-3)
        if self.grouped:
            wc, front, back = be.linked_match_batch(lset._handle, fused_batch.packed, fused_batch.lens, fused_batch.nreads, fused_batch.max_len)
            which, count, f2, b2 = self.res
            out["records_equal_fused_form"] = bool(torch.equal(which, wc[:, 0].to(torch.int32)) and torch.equal(f2[:, :6], front[:, :6])
                                                   and torch.equal(b2[:, :6], back[:, :6]))
        return out


CONFIGS = {"C2": C2, "C3": C3, "C4": C4, "C5": C5}

EMU = os.environ.get("ATROPOS_BENCH_BACKEND") == "emu"      # tests/test_distributed.py: the launcher on CPU, test double


class HostEvent(object):
    """Stand-in for torch.cuda.Event when the launcher is exercised on CPU with the test double."""
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def new_event():
    return HostEvent() if EMU else torch.cuda.Event(enable_timing=True)


def device_sync():
    if not EMU:
        torch.cuda.synchronize()


def device_name(ordinal):
    return "cpu" if EMU else "cuda:%d" % ordinal


def measure(cfg_cls, args, rank, world, local_rank, dist, headline, defer_secondary=False):
    """Build one config's workload, run warmup + EXACTLY args.steps timed steps (barrier + synchronize on both
    sides, HIP events on the launch stream around every step, max over ranks) and return its result dict
    (rank 0; None elsewhere).  headline: the config whose figures are the top-level fields of the line."""
    cargs = argparse.Namespace(**vars(args))
    cargs.reads = args.reads if (headline and args.reads is not None) else cfg_cls.default_units
    if EMU:                                                      # the CPU test double sweeps ~100 k reads/s
        cargs.reads = min(cargs.reads, int(os.environ.get("ATROPOS_BENCH_EMU_UNITS", "2000")))
    cfg = cfg_cls(cargs, rank, device_name(local_rank))
    device_sync()

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    for s in range(args.warmup):
        cfg.step(s)
    ev = [(new_event(), new_event()) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        cfg.step(args.warmup + s)
        ev[s][1].record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    tall = torch.zeros((2, world), dtype=torch.float64)          # host tensor: the group is gloo
    tall[0, rank], tall[1, rank] = dt, kernel_ms
    if dist is not None:
        dist.all_reduce(tall, op=dist.ReduceOp.SUM)
    per_rank = [float(x) for x in tall[0].tolist()]
    per_rank_kernel_ms = [float(x) for x in tall[1].tolist()]
    dt_max = max(per_rank)
    if rank != 0:
        return None

    units = cargs.reads * cfg.unit_reads                         # reads per step and GPU
    wall_ms = dt_max / args.steps * 1e3
    # roofline.achieved / frac from the WALL time of a step (what `value` is made of); the HIP-event time of the call on
    # the launch stream is kept beside it (kernel_ms, frac_event: 1 - 2 % shorter -- host gaps between calls)
    achieved = cfg.algo_bytes * cargs.reads / (wall_ms * 1e-3) / 1e9
    achieved_event = cfg.algo_bytes * cargs.reads / (kernel_ms * 1e-3) / 1e9
    prof = profile_counters(cfg.name) if not args.full_sweep else None
    desc = cfg.describe()
    live = None
    if args.live_counters and world == 1 and not args.full_sweep and not EMU:
        live = live_counters(cfg.name, cargs.reads, headline, getattr(cfg, "read_len", 150), getattr(cfg, "stream_units", None))        # (child processes on the same GPU)
    desc["parallelism"] = "shard%d" % world
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "peak_measured": HBM_MEASURED_GBS,
            "frac_of_measured": achieved / HBM_MEASURED_GBS,
            "traffic": None, "kernel": cfg.kernel(), "kernel_ms": kernel_ms, "frac_event": achieved_event / HBM_PEAK_GBS,
            "time_base": "wall ms_per_step (frac_event: HIP events around the call on the launch stream)",
            "algorithmic_bytes_per_unit": cfg.algo_bytes, "note": cfg.note(kernel_ms)}
    if hasattr(cfg, "jit"):
        roof["jit"] = cfg.jit()
    if live:
        roof["traffic"] = live["hbm_bytes_per_launch"]
        roof["traffic_source"] = live["source"]
        roof["traffic_over_algorithmic"] = live["hbm_bytes_per_launch"] / (cfg.algo_bytes * cargs.reads)
        if prof is None:
            prof = {}
        prof = dict(prof, units_per_launch=cargs.reads)
        if live.get("valu_wave_insts_per_launch"):
            prof["valu_wave_insts_per_launch"] = live["valu_wave_insts_per_launch"]
        else:
            prof.pop("valu_wave_insts_per_launch", None)
    if prof:
        scale = cargs.reads / prof["units_per_launch"]
        if not live:
            roof["traffic"] = prof["hbm_bytes_per_launch"] * scale
            roof["traffic_source"] = "profiles/traffic_%s.json (rocprofv3 PMC passes of an earlier run, not this run)" % cfg.name
        if prof.get("valu_wave_insts_per_launch"):
            lane_ops = prof["valu_wave_insts_per_launch"] * scale * 64 / (kernel_ms * 1e-3) / 1e12
            roof["valu"] = {"wave_insts_per_launch": prof["valu_wave_insts_per_launch"] * scale,
                            "source": roof["traffic_source"], "achieved": lane_ops, "unit": "T lane-ops/s",
                            "peak": VALU_NOMINAL_T, "frac": lane_ops / VALU_NOMINAL_T,
                            "mixed_stream_model": VALU_MIXED_MODEL_T,
                            "frac_of_mixed_stream_model": lane_ops / VALU_MIXED_MODEL_T,
                            "model_note": "39.3 T = every VALU op of a mixed integer stream at 4 issue cycles; a "
                                          "model of this instruction mix, not a hardware ceiling"}
            # (flat copies: a reader that keeps only the scalar fields of `roofline` still sees the issue-rate figures)
            roof["valu_wave_insts_per_launch"] = roof["valu"]["wave_insts_per_launch"]
            roof["valu_frac_of_nominal"] = roof["valu"]["frac"]
            roof["valu_frac_of_4cycle_model"] = roof["valu"]["frac_of_mixed_stream_model"]
    out = {
        "metric": cfg.metric, "value": units * world * args.steps / dt_max, "unit": "reads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg.dtype, "data": "synthetic",
        "config": desc, "roofline": roof, "launcher": args.launcher,
        "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
    }
    if world > 1:
        # the roofline object is rank 0's GPU (every rank runs the same shard size); the other ranks' event times:
        roof["per_rank_kernel_ms"] = per_rank_kernel_ms
        roof["note"] = "per GPU (rank 0); " + roof["note"]
    if args.oversubscribe:
        out["oversubscribed"] = "%d ranks on %d device(s): a launcher check, NOT a scaling figure" % (world, args.ndev)
    # side measurements and the CPU leg belong to the single-GPU line (rank 0 at N = 1): the other ranks of a
    # multi-GPU run would only wait at the next barrier for them
    if args.secondary and world == 1:
        if headline and defer_secondary:
            # the headline's side measurements (short batches, per-call paths, two streams ...) run after the other
            # configs have been timed: they leave state behind -- C4 measured 5 % lower right after them than alone
            out["_secondary_later"] = (cfg, kernel_ms)
        else:
            out["secondary"] = cfg.secondary(kernel_ms)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cfg.cpu_baseline()
    return out


def measure_threads(cfg_cls, args, devices):
    """--single-process: ONE process, one host thread + HipBackend + stream per GPU (atropos_amd.shard's
    single-process driver: no torchrun, no process group, no NCCL -- the path shards without any exchange, so a
    thread barrier is all the synchronisation the contract needs).  Every thread builds its own shard of the
    workload on its device, runs the warmup, meets the others at a barrier, times EXACTLY args.steps steps and
    synchronises its stream; the job's time is first start to last finish."""
    import threading
    from atropos_amd import _lib
    world = len(devices)
    cargs = argparse.Namespace(**vars(args))
    cargs.reads = args.reads if args.reads is not None else cfg_cls.default_units
    start = threading.Barrier(world)
    t_begin, t_end, kernel_ms, errors, cfgs = [0.0] * world, [0.0] * world, [0.0] * world, [], [None] * world

    def run(rank):
        try:
            dev = devices[rank]
            be = _lib.HipBackend(dev)
            with _lib.thread_backend(be), be.worker_context() as stream:
                cfg = cfgs[rank] = cfg_cls(cargs, rank, "cuda:%d" % dev)
                for s in range(args.warmup):
                    cfg.step(s)
                stream.synchronize()
                ev = [(new_event(), new_event()) for _ in range(args.steps)]
                start.wait()
                t_begin[rank] = time.perf_counter()
                for s in range(args.steps):
                    ev[s][0].record()
                    cfg.step(args.warmup + s)
                    ev[s][1].record()
                stream.synchronize()
                t_end[rank] = time.perf_counter()
                kernel_ms[rank] = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        except BaseException as err:                      # noqa: BLE001 -- re-raised below
            errors.append(err)
            start.abort()

    threads = [threading.Thread(target=run, args=(r,), name="bench-dev%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    dt = max(t_end) - min(t_begin)
    cfg = cfgs[0]
    units = cargs.reads * cfg.unit_reads
    kms = float(np.mean(kernel_ms))
    achieved = cfg.algo_bytes * cargs.reads / (kms * 1e-3) / 1e9
    with _lib.thread_backend(_lib.HipBackend(devices[0])):
        desc = cfg.describe()
    desc["parallelism"] = "shard%d" % world
    return {
        "metric": cfg.metric, "value": units * world * args.steps / dt, "unit": "reads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg.dtype, "data": "synthetic",
        "config": desc, "launcher": "single process, one host thread + stream per GPU (no process group)",
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": cfg.kernel(), "kernel_ms": kms, "algorithmic_bytes_per_unit": cfg.algo_bytes,
                     "note": "per GPU; " + cfg.note(kms)},
        "per_rank_ms_per_step": [(e - b) / args.steps * 1e3 for b, e in zip(t_begin, t_end)],
    }


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), pass
    rank 0's JSON line through, return the first non-zero exit code.  The CPU analogue of the fan-out is
    /root/reference/atropos/commands/multicore.py:297-401 (worker processes, results merged by the parent)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ndev = 1 if EMU else torch.cuda.device_count()
    if not EMU and ndev < args.gpus and not args.oversubscribe:
        raise SystemExit("--gpus %d but only %d GPU(s) visible on this node (--oversubscribe shares them: launcher "
                         "check only)" % (args.gpus, ndev))
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--launcher", "bench.py self-launch, one process per GPU"]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if rank == 0 else subprocess.DEVNULL))
    rc = 0
    failed = False
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and not failed:
                rc, failed = code, True
                for q in procs:                       # a dead rank would leave the others waiting at the barrier
                    q.terminate()
        time.sleep(0.05)
    return rc


def scaling_curve(args):
    """`python bench.py --scaling`: the N = 1, 2, 4, 8 lines of the headline config from one invocation, each N as a
    self-launched group of N ranks (one process per GPU, gloo barrier, no collective on the data path).  N beyond the
    node's GPU count is left out and said so on stderr -- a curve is only claimed for what ran."""
    import subprocess
    ndev = 1 if EMU else torch.cuda.device_count()
    rc = 0
    base = [a for a in sys.argv[1:] if a != "--scaling"]
    skip = False
    argv = []
    for a in base:                                        # (drop a --gpus N the caller may have given)
        if skip:
            skip = False
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        argv.append(a)
    for n in (1, 2, 4, 8):
        if n > ndev and not args.oversubscribe:
            sys.stderr.write("bench.py --scaling: N = %d left out, %d GPU(s) visible\n" % (n, ndev))
            continue
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--gpus", str(n), "--no-cpu-baseline", "--no-secondary",
                                                                   "--no-live-counters", "--no-other-configs"]
        code = subprocess.call(cmd)
        rc = rc or code
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="time this BASELINE config only; default: C2 is the headline (top-level fields) and C3, C4, "
                         "C5 follow in the same process under \"configs\"")
    ap.add_argument("--rotate", type=int, default=3, help="C2: resident batches the timed steps rotate over (default 3)")
    ap.add_argument("--reads", type=int, default=None, help="reads (C2, C4) or pairs (C3, C5) per GPU and step of the "
                                                            "headline config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the pack-inclusive / ragged-batch / small-batch side measurements")
    ap.add_argument("--no-live-counters", dest="live_counters", action="store_false",
                    help="take roofline.traffic from the committed profile instead of measuring it with rocprofv3 PMC passes "
                         "of this invocation")
    ap.add_argument("--no-other-configs", dest="others", action="store_false",
                    help="default run: C2 only, without the C3 / C4 / C5 entries")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: a host thread, backend and stream per GPU (no torchrun / NCCL); times "
                         "the headline config only")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ordinals (default 0 .. N-1)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="--gpus N on a box with fewer GPUs: ranks share devices (rank modulo count).  Exercises the launcher; "
                         "the line is marked and is not a scaling figure")
    ap.add_argument("--scaling", action="store_true",
                    help="ONE invocation for the whole curve: the N = 1, 2, 4, 8 lines back to back (those the node's GPU count "
                         "allows; headline config only, no side measurements), each with per_rank_ms_per_step and every "
                         "rank's kernel time in roofline.per_rank_kernel_ms.  The CPU analogue of the fan-out: "
                         "/root/reference/atropos/commands/multicore.py:297-401")
    ap.add_argument("--launcher", default=None, help=argparse.SUPPRESS)       # set by self_launch() for its ranks
    ap.add_argument("--full-sweep", action="store_true",
                    help="C2: time the unfiltered full-column DP kernel instead of the filtered pipeline")
    args = ap.parse_args()

    if args.scaling:
        raise SystemExit(scaling_curve(args))
    if args.single_process:
        devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit("--devices must name --gpus devices")
        print(json.dumps(measure_threads(CONFIGS[args.config or "C2"], args, devices)), flush=True)
        return
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks of this same script
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    args.ndev = 1 if EMU else torch.cuda.device_count()
    if not EMU and local_rank >= args.ndev:
        if not args.oversubscribe:
            raise SystemExit("rank %d of %d but only %d GPU(s) visible (--oversubscribe shares them: launcher check only)"
                             % (local_rank, world, args.ndev))
        local_rank %= args.ndev
    args.oversubscribe = args.oversubscribe and world > args.ndev
    if args.launcher is None:
        args.launcher = ("torch.distributed.run, one process per GPU" if "TORCHELASTIC_RUN_ID" in os.environ
                         else "one process" if world == 1 else "environment (RANK / WORLD_SIZE), one process per GPU")
    if world > 1:
        args.launcher += "; barrier + max-over-ranks over a gloo host group, no RCCL"
    if not EMU:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on the C-level stdout; the contract is ONE JSON line there
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
            dist.barrier()
        finally:
            os.dup2(keep, 1)
            os.close(keep)

    from atropos_amd import _lib
    if EMU:
        from tests.emu.backend import EmuBackend
        _lib.set_backend(EmuBackend(), _test_double=True)
    else:
        _lib.set_backend(_lib.HipBackend(local_rank))
    head = args.config or "C2"
    with_others = args.config is None and args.others and not args.full_sweep
    line = measure(CONFIGS[head], args, rank, world, local_rank, dist, True, defer_secondary=with_others)
    if args.config is None and args.others and not args.full_sweep:
        # the other BASELINE configs on the same clock: same steps / warmup, one after the other, each with its
        # own roofline and (N = 1) cpu_baseline.  A failure there must not cost the headline its line.
        others = {}
        for name in ("C3", "C4", "C5"):
            if not EMU:
                torch.cuda.empty_cache()
            try:
                res = measure(CONFIGS[name], args, rank, world, local_rank, dist, False)
            except Exception as exc:                              # noqa: BLE001 -- reported in the line
                if dist is not None:
                    raise                                         # the ranks would lose step with each other
                res = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if rank == 0:
                others[name] = res
        if rank == 0:
            line["configs"] = others
    if rank == 0 and line is not None and "_secondary_later" in line:
        cfg, kernel_ms = line.pop("_secondary_later")
        line["secondary"] = cfg.secondary(kernel_ms)
    if rank == 0:
        # LAST key of the line (the driver keeps the line's tail): every config of this invocation in one compact
        # object -- [reads/s, ms_per_step, roofline.frac, counted traffic / algorithmic bytes (null: not counted)]
        def _brief(res):
            if not isinstance(res, dict) or "value" not in res:
                return None
            roof = res.get("roofline", {})
            return [float("%.4g" % res["value"]), float("%.4g" % res["ms_per_step"]), float("%.4g" % roof.get("frac", 0.0)),
                    None if roof.get("traffic_over_algorithmic") is None else float("%.3g" % roof["traffic_over_algorithmic"])]
        summary = {head: _brief(line)}
        for name, res in line.get("configs", {}).items():
            summary[name] = _brief(res)
        line["configs_summary"] = summary
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
