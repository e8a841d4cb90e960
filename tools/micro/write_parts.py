#!/usr/bin/env python3
"""How fast can this host write a fresh file: one writer, or N writers each to its own part, then a concatenation by
copy_file_range / by rename of part 0 + appends.  usage: tools/micro/write_parts.py [GiB] [dir]"""
import os
import sys
import threading
import time

import numpy as np

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
root = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
total = int(gib * (1 << 30))
block = 64 << 20
buf = np.random.default_rng(1).integers(33, 120, block, dtype=np.uint8)
mv = memoryview(buf)


def write_file(path, nbytes, off0=0, preopen=None):
    fd = preopen if preopen is not None else os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    done = 0
    while done < nbytes:
        n = min(block, nbytes - done)
        os.pwrite(fd, mv[:n], off0 + done)
        done += n
    if preopen is None:
        os.close(fd)


for writers in (1, 2, 4, 8):
    paths = [os.path.join(root, "wp_%d_%d.bin" % (writers, i)) for i in range(writers)]
    per = total // writers
    t0 = time.perf_counter()
    th = [threading.Thread(target=write_file, args=(p, per)) for p in paths]
    [t.start() for t in th]
    [t.join() for t in th]
    t1 = time.perf_counter()
    # concatenate parts 1.. onto part 0 with copy_file_range (in-kernel copy; reflink where the fs can)
    if writers > 1:
        dst = os.open(paths[0], os.O_WRONLY)
        at = per
        for p in paths[1:]:
            src = os.open(p, os.O_RDONLY)
            left, so = per, 0
            while left:
                n = os.copy_file_range(src, dst, left, so, at)
                if n == 0:
                    break
                left -= n; so += n; at += n
            os.close(src)
        os.close(dst)
    t2 = time.perf_counter()
    print("writers %d: write %.2f GB/s, concat %.2f s, all %.2f GB/s" % (writers, total / (t1 - t0) / 1e9, t2 - t1, total / (t2 - t0) / 1e9), flush=True)
    for p in paths:
        os.unlink(p)
# one file, N threads writing disjoint ranges of it
for writers in (2, 4, 8):
    path = os.path.join(root, "wp_shared.bin")
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    per = total // writers
    t0 = time.perf_counter()
    th = [threading.Thread(target=write_file, args=(path, per, i * per, fd)) for i in range(writers)]
    [t.start() for t in th]
    [t.join() for t in th]
    os.close(fd)
    t1 = time.perf_counter()
    print("one file, %d range writers: %.2f GB/s" % (writers, total / (t1 - t0) / 1e9), flush=True)
    os.unlink(path)
