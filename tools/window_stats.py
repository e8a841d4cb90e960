#!/usr/bin/env python3
"""Diagnostics of the filtered locate pipeline on workload C2 (GPU): how many reads the
pre-pass resolves, and the window lengths of the rest."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import _lib, synth          # noqa: E402
from atropos_amd.align import Aligner        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
w = synth.workload("C2", 0, n, device="cuda")
al = Aligner(w["adapter"], 0.1, 14, False, False, 3, 1)
b = al.pack(w["reads"])
res = al.locate_batch(b)
torch.cuda.synchronize()
be = _lib.get_backend()
work = be._work.view(torch.int32)
win = work[:n].to(torch.int64) & 0xFFFFFFFF
valid = (win >> 31) & 1
lo, hi, scan = win & 0x3FF, (win >> 10) & 0x3FF, (win >> 20) & 1
v = valid.bool()
print("reads", n, "unresolved", int(v.sum()), "fraction %.4f" % (v.float().mean().item()))
ln = (hi - lo)[v].float()
print("window length mean %.1f median %.1f max %d" % (ln.mean().item(), ln.median().item(), int(ln.max().item())))
print("with last-column scan: %.3f" % scan[v].float().mean().item())
found = res.found()
print("found fraction %.4f; found among unresolved %.4f" % (found.float().mean().item(), found[v].float().mean().item()))
order_total = int(work[2 * n + 8192 * 256 + 256].item())     # FastWork layout: win, order, counts, binbase, total
print("total word", order_total)
# per-wave union window (as the window kernel sees it): waves of 64 consecutive slots of `order`
order = work[n:n + order_total].to(torch.int64)
wl, wh = lo[order], hi[order]
nw = order_total // 64
u = (wh[:nw * 64].view(nw, 64).max(dim=1).values - wl[:nw * 64].view(nw, 64).min(dim=1).values).float()
print("per-wave union sweep mean %.1f columns (%d waves)" % (u.mean().item(), nw))
rows = (win >> 21) & 0x7F
for name, sel in (("scan (last-column) reads", v & (scan == 1)), ("row-m only reads", v & (scan == 0))):
    if int(sel.sum()) == 0:
        continue
    print("%s: n=%d window mean %.1f rows mean %.1f  rows==34: %.3f  j_lo mean %.1f" % (
        name, int(sel.sum()), (hi - lo)[sel].float().mean().item(), rows[sel].float().mean().item(),
        (rows[sel] == 34).float().mean().item(), lo[sel].float().mean().item()))
ro = rows[order]
pw = ro[:nw * 64].view(nw, 64).max(dim=1).values.float()
print("per-wave row limit mean %.1f; waves with full rows %.3f" % (pw.mean().item(), (pw == 34).float().mean().item()))
cells = (u * pw).sum().item() * 64
print("swept cells per unresolved read %.0f" % (cells / order_total))
# classes of the unresolved reads (window word bit 28 = band)
band = ((win >> 28) & 1) == 1
m = 34
cls = {
    "band (row-m only, <= 16 diagonals)": v & band,
    "row-binned partial overlaps (rows < m)": v & ~band & (rows < m),
    "full rows, last column too": v & ~band & (rows >= m) & (scan == 1),
    "full rows, window clamped at column 0": v & ~band & (rows >= m) & (scan == 0) & (lo == 0),
    "full rows, span too wide": v & ~band & (rows >= m) & (scan == 0) & (lo > 0),
}
for name, sel in cls.items():
    k = int(sel.sum())
    if k:
        print("%-45s n=%7d (%.3f of unresolved) window mean %.1f lo mean %.1f" % (
            name, k, k / int(v.sum()), (hi - lo)[sel].float().mean().item(), lo[sel].float().mean().item()))
