#!/usr/bin/env python3
"""What the two-pass pre-pass leaves to the DP on workload C2, by kind (CPU twin of the kernels, tests/emu): the window
words of a sample's reads against their final records.  usage: tools/c2_unresolved.py [nreads]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.emu.backend import EmuBackend          # noqa: E402
from atropos_amd import _lib, synth               # noqa: E402

_lib.set_backend(EmuBackend(), _test_double=True)
from atropos_amd.align import Aligner             # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
w = synth.workload("C2", 0, n)
al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
m = len(w["adapter"])
rec = al.locate_batch(al.pack(w["reads"], layout="plane64")).numpy()[:, :6].astype(np.int64)
lib = _lib.get_backend().lib
win = np.zeros(n, np.uint32)
lib.emu_piece_last_windows.restype = C.c_longlong
assert lib.emu_piece_last_windows(win.ctypes.data_as(C.c_void_p), C.c_longlong(n)) == n
open_ = (win >> 31) != 0
band, scan = ((win >> 28) & 1) != 0, ((win >> 20) & 1) != 0
found = rec[:, 1] >= 0
rl, ql = rec[:, 1] - rec[:, 0], rec[:, 3] - rec[:, 2]
mism = rl + ql - 2 * rec[:, 4] - rec[:, 5]
indel = rec[:, 5] - mism
full = found & (rl == m)
print("reads %d, unresolved %d = %.2f %%" % (n, open_.sum(), 100.0 * open_.mean()))
def show(name, sel):
    print("  %-58s %7d  %5.2f %% of reads  %5.1f %% of unresolved" % (name, sel.sum(), 100.0 * sel.mean(), 100.0 * sel.sum() / max(1, open_.sum())))
show("row-m band reads", open_ & band & ~scan)
show("last-column band reads", open_ & band & scan)
show("window reads (no band)", open_ & ~band)
show("no match in the end", open_ & ~found)
show("whole adapter, 0 errors", open_ & full & (rec[:, 5] == 0))
show("whole adapter, 1 error, substitution", open_ & full & (rec[:, 5] == 1) & (indel == 0))
show("whole adapter, 1 error, indel", open_ & full & (rec[:, 5] == 1) & (indel == 1))
show("whole adapter, 2 errors, substitutions only", open_ & full & (rec[:, 5] == 2) & (indel == 0))
show("whole adapter, 2 errors with an indel", open_ & full & (rec[:, 5] == 2) & (indel > 0))
show("whole adapter, 3 errors", open_ & full & (rec[:, 5] == 3))
show("partial adapter at the read end, 0 errors", open_ & found & ~full & (rec[:, 5] == 0))
show("partial adapter at the read end, >= 1 error", open_ & found & ~full & (rec[:, 5] >= 1))
for k in range(0, m + 1, 4):
    sel = open_ & found & ~full & (rl >= k) & (rl < k + 4)
    if sel.sum():
        show("   partial, %2d .. %2d adapter bases" % (k, k + 3), sel)
# the window reads: why no band took them (window word: [9:0] j_lo, [19:10] j_hi, [20] last-column candidates, [27:21] rows)
wsel = open_ & ~band
j_lo, j_hi, rows = (win & 1023).astype(np.int64), ((win >> 10) & 1023).astype(np.int64), ((win >> 21) & 127).astype(np.int64)
print("window reads: %d; with last-column candidates %d, row-m only %d" % (wsel.sum(), (wsel & scan).sum(), (wsel & ~scan).sum()))
for name, sel in (("row-m only", wsel & ~scan), ("with last-column candidates", wsel & scan)):
    if sel.sum():
        wd = (j_hi - j_lo)[sel]
        print("  %-28s window columns: mean %.1f  p50 %d  p90 %d  max %d; rows mean %.1f; query start of the match: mean %.1f" % (
            name, wd.mean(), np.percentile(wd, 50), np.percentile(wd, 90), wd.max(), rows[sel].mean(), rec[sel, 2].mean()))
