#!/usr/bin/env python3
"""What the host side of the file -> file path can do at best on this box: multi-threaded pread of a
page-cached file into a page-locked buffer, pwrite of a page-locked buffer into a file (fresh and
overwritten), H2D / D2H copies of the same buffers.  usage: tools/micro/host_io.py [MiB]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
path = "/tmp/atr_host_io.bin"
buf = torch.empty((n,), dtype=torch.uint8)
if torch.cuda.is_available():
    buf = buf.pin_memory()
buf.random_(0, 255)
view = memoryview(buf.numpy())
with open(path, "wb") as fh:
    fh.write(view)
res = {"MiB": mib, "cpus": os.cpu_count()}
fd = os.open(path, os.O_RDWR)


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return n / ((time.perf_counter() - t0) / reps) / 1e9


for threads in (1, 2, 4, 8, 16):
    pool = ThreadPoolExecutor(threads)
    step = (n + threads - 1) // threads

    def rd():
        jobs = [pool.submit(os.preadv, fd, [view[t * step:min(n, (t + 1) * step)]], t * step) for t in range(threads)]
        assert sum(j.result() for j in jobs) == n

    def wr():
        jobs = [pool.submit(os.pwrite, fd, view[t * step:min(n, (t + 1) * step)], t * step) for t in range(threads)]
        assert sum(j.result() for j in jobs) == n

    def wr_fresh():
        p2 = path + ".new"
        f2 = os.open(p2, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        os.posix_fallocate(f2, 0, n)
        jobs = [pool.submit(os.pwrite, f2, view[t * step:min(n, (t + 1) * step)], t * step) for t in range(threads)]
        assert sum(j.result() for j in jobs) == n
        os.close(f2)
        os.remove(p2)

    def wr_mmap():
        # fresh file, grown with ftruncate, filled through a shared mapping by `threads` copiers
        import mmap
        import numpy as np
        p2 = path + ".map"
        f2 = os.open(p2, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        os.ftruncate(f2, n)
        mm = mmap.mmap(f2, n, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        dst = torch.from_numpy(np.frombuffer(mm, dtype=np.uint8))
        jobs = [pool.submit(dst[t * step:min(n, (t + 1) * step)].copy_, buf[t * step:min(n, (t + 1) * step)]) for t in range(threads)]
        for j in jobs:
            j.result()
        del jobs, dst                                  # the views keep the mapping alive; it goes with them
        mm = None
        os.close(f2)
        os.remove(p2)

    res["mmap_write_fresh_GBps_%d" % threads] = round(timed(wr_mmap, 2), 2)
    res["pread_GBps_%d" % threads] = round(timed(rd), 2)
    res["pwrite_overwrite_GBps_%d" % threads] = round(timed(wr), 2)
    res["pwrite_fresh_GBps_%d" % threads] = round(timed(wr_fresh, 2), 2)
    pool.shutdown()
# O_DIRECT: page-cache bypass (the staging buffers are page aligned); unsupported on some file systems
for threads in (1, 4):
    try:
        p2 = path + ".direct"
        f2 = os.open(p2, os.O_RDWR | os.O_CREAT | os.O_TRUNC | os.O_DIRECT)
        pool = ThreadPoolExecutor(threads)
        step = ((n + threads - 1) // threads + 4095) & ~4095

        def wr_direct():
            jobs = [pool.submit(os.pwrite, f2, view[t * step:min(n, (t + 1) * step)], t * step) for t in range(threads)]
            assert sum(j.result() for j in jobs) == n
        res["pwrite_O_DIRECT_GBps_%d" % threads] = round(timed(wr_direct, 2), 2)
        pool.shutdown()
        os.close(f2)
        os.remove(p2)
    except OSError as err:
        res["pwrite_O_DIRECT_%d" % threads] = "unsupported: %s" % err
if torch.cuda.is_available():
    dev = torch.empty((n,), dtype=torch.uint8, device="cuda")

    def h2d():
        dev.copy_(buf, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        buf.copy_(dev, non_blocking=True)
        torch.cuda.synchronize()
    res["h2d_GBps"] = round(timed(h2d), 2)
    res["d2h_GBps"] = round(timed(d2h), 2)
os.close(fd)
os.remove(path)
print(json.dumps(res))
