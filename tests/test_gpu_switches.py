"""The library's run-time switches pick between launch structures, never between results: the records of
the two-pass pipeline (C2) and of a linked set (C4) with every switch at its non-default value equal the default ones.
The switches are read once per process, so each setting runs in a child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
from atropos_amd import _lib, synth
from atropos_amd.align import Aligner
from atropos_amd.adapters import AsciiSource, LinkedAdapter, LinkedSet, upper_ascii
_lib.set_backend(None)
out = {}
w = synth.workload("C2", 7 << 20, 300000, device="cuda:0")
al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
planes = al.pack(w["reads"], layout="plane64")
rec = al.locate_batch(planes).numpy()[:, :6]
out["c2"] = hashlib.sha256(rec.tobytes()).hexdigest()
out["c2_matched"] = int((rec[:, 1] >= 0).sum())
w = synth.workload("C4", 3 << 20, 400000, device="cuda:0")
reads = upper_ascii(w["reads"])
linked = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                        min_overlap=w["min_overlap"], indel_cost=w["indel_cost"]) for f, b in zip(w["fronts"], w["backs"])]
lset = LinkedSet(linked)
assert lset.fused
batch = AsciiSource(reads).batch(lset.table_kind, lset.table)
wc, front, back = lset._backend.linked_match_batch(lset._handle, batch.packed, batch.lens, batch.nreads, batch.max_len)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (wc, front, back):
    h.update(t.cpu().numpy().tobytes())
out["c4"] = h.hexdigest()
out["c4_back_matched"] = int((back[:, 1] >= 0).sum().item())
print(json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_switches_change_no_record(hip_backend):
    base = _run({})
    assert base["c2_matched"] > 100000 and base["c4_back_matched"] > 100000
    for env in ({"ATR_FUSED_SCAN": "0"}, {"ATR_ONE_WINDOW": "0", "ATR_WINDOW_PRIORITY": "0"},
                {"ATR_JIT": "0"}, {"ATR_JIT": "1", "ATR_SPEC_FLAGS": "-DATR_PIECE_STASH=0"},
                {"ATR_FUSED_SCAN": "0", "ATR_ONE_WINDOW": "0"}):
        assert _run(env) == base, env
