#!/usr/bin/env python3
"""A/B of the C2 call (10 M x 150 bp, plane64) under different pre-pass builds: the generic kernel (ATR_JIT=0), the
run-time specialised one (ATR_JIT=1) and experiment switches ($ATR_SPEC_FLAGS, e.g. "-DATR_SPEC_X=1").  One child
process per variant (the environment is read by the library); prints ms per call (HIP events, median of rounds).

    python tools/jit/ab.py [--reads N] [--ragged] [variant ...]      variant = name:ENV=V,ENV=V  (default: generic, spec)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from atropos_amd import _lib, synth
    from atropos_amd.align import Aligner
    _lib.set_backend(None)
    w = synth.workload(args.config, 0, args.reads, device="cuda:0")
    al = Aligner(w["adapter"], args.e if args.e else w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    reads = w["reads"]
    if args.ragged:
        rng = np.random.default_rng(3)
        lens = torch.from_numpy(rng.integers(100, 151, size=args.reads).astype(np.int32)).cuda()
        col = torch.arange(150, device="cuda:0")[None, :]
        reads = torch.where(col < lens[:, None], reads, torch.zeros_like(reads))
        from atropos_amd.batch import ReadBatch
        batch = ReadBatch.from_ascii(reads, lens, None, al.table_kind, None, _lib.get_backend(), planes=True)
    elif os.environ.get("AB_LAYOUT") == "tile64":
        batch = al.pack(reads, layout="tile64")
    else:
        batch = al.pack(reads, layout="plane64")
    t_prep = None
    if os.environ.get("ATR_JIT") == "1":
        import time
        t0 = time.perf_counter()
        ok = al.prepare(150, ragged=args.ragged)
        t_prep = (time.perf_counter() - t0, ok)
    for _ in range(3):
        rec = al.locate_batch(batch)
    torch.cuda.synchronize()
    times = []
    for _ in range(args.rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            rec = al.locate_batch(batch)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / args.steps)
    r = rec.records if hasattr(rec, "records") else rec
    digest = int(r.to(torch.int64).sum().item())
    print(json.dumps({"ms": sorted(times)[len(times) // 2], "min": min(times), "digest": digest, "prepare": t_prep}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--e", type=float, default=0.0)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    variants = args.variants or ["generic:ATR_JIT=0", "spec:ATR_JIT=1"]
    for v in variants:
        name, _, envs = v.partition(":")
        env = dict(os.environ)
        for kv in envs.split(","):
            if kv:
                k, _, val = kv.partition("=")
                env[k] = val.replace("+", " ")
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--reads", str(args.reads), "--steps", str(args.steps),
               "--rounds", str(args.rounds), "--config", args.config, "--e", str(args.e)] + (["--ragged"] if args.ragged else [])
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr[-600:]
        print("%-28s %s" % (name, line), flush=True)
        for extra in [l for l in out.stdout.splitlines() if l.startswith("wave ")][:8]:      # (-DATR_X_TIMING=1: device printf)
            print("    " + extra, flush=True)


if __name__ == "__main__":
    main()
