#!/usr/bin/env python3
"""The small-batch regime of the drop-in path: reads/s of C2's aligner against the batch size (1, 1 000 -- what
the unchanged trim command hands over per call, /root/reference/atropos/commands/base.py:179 -- 64 k, 1 M), for
  packed   : atr_locate_batch on a resident tile64 batch (records stay on the device),
  ascii    : pack + locate from an ASCII matrix resident on the device,
  strings  : Python strings in, result tuples out (host -> device -> host), i.e. Aligner.locate_batch(list),
the per-read API: Aligner.locate(str) -> tuple, one GPU batch of one per call; and the pair-wise entry points
(insert aligner, MergeOverlapping's per-pair aligner) on 1 000 pairs.
usage: tools/bench_small.py [json-out]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atropos_amd import synth                          # noqa: E402
from atropos_amd.align import Aligner                  # noqa: E402


def timed(fn, min_reps=5, target_s=0.5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    reps = int(max(min_reps, min(2000, target_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def measure(sizes=(1, 1000, 65536, 1_000_000)):
    w = synth.workload("C2", 0, max(sizes), device="cuda")
    al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    # the aligner's specialised pre-pass built BEFORE anything is timed: under the automatic policy it is compiled once a
    # handle has seen 2 M reads -- inside the timed loop of the 65 536-read point, 0.8 s spread over its repetitions (the
    # 523 us of the first round-5 file)
    if hasattr(al, "prepare"):
        al.prepare(int(w["reads"].shape[1]))
    host = w["reads"].cpu().numpy()
    out = {"batch_sizes": list(sizes), "packed_us": [], "ascii_us": [], "strings_us": [], "strings_all_tuples_made_us": []}
    for n in sizes:
        a = w["reads"][:n].contiguous()
        batch = al.pack(a, layout="auto")
        out["packed_us"].append(timed(lambda: al.locate_batch(batch)) * 1e6)
        out["ascii_us"].append(timed(lambda: al.locate_batch(al.pack(a, layout="auto"))) * 1e6)
        if n <= 65536:
            strings = [bytes(r).decode() for r in host[:n]]
            # .tuples() is a sequence that makes a tuple when it is looked at; the second figure makes them all
            out["strings_us"].append(timed(lambda: al.locate_batch(strings).tuples()[n - 1], min_reps=3) * 1e6)
            out["strings_all_tuples_made_us"].append(timed(lambda: al.locate_batch(strings).tuples().tolist(), min_reps=3) * 1e6)
        else:
            out["strings_us"].append(None)
            out["strings_all_tuples_made_us"].append(None)
    for key in ("packed", "ascii", "strings"):
        out[key + "_reads_per_s"] = [None if t is None else n / (t * 1e-6) for n, t in zip(sizes, out[key + "_us"])]
    one = bytes(host[0]).decode()
    t = timed(lambda: al.locate(one), min_reps=50)
    out["per_read_locate_us"] = t * 1e6
    out["per_read_locate_reads_per_s"] = 1.0 / t
    out["pairs_1000"] = measure_pairs()
    out["per_call_us"] = measure_per_call(host)
    return out


def measure_per_call(c2_host):
    """The per-object calls the one-line module swap of INTEGRATION.md section 1 makes (one GPU call each), and a set of
    four linked adapters on a 1 000-read batch."""
    from atropos_amd.adapters import LinkedAdapter, LinkedSet
    from atropos_amd.align import MultiAligner, compare_prefixes
    from atropos_amd.util import reverse_complement
    w3 = synth.workload("C3", 0, 8)
    r1 = bytes(w3["reads1"][2].cpu().numpy()).decode()
    rc2 = reverse_complement(bytes(w3["reads2"][2].cpu().numpy()).decode())
    q = bytes(c2_host[5]).decode()
    ma = MultiAligner(0.2, 9, 1)
    out = {"MultiAligner.locate_2x150": timed(lambda: ma.locate(rc2, r1), min_reps=50) * 1e6,
           "compare_prefixes": timed(lambda: compare_prefixes(synth.TRUSEQ_34, q), min_reps=50) * 1e6,
           "new_Aligner_150_base_reference_plus_locate": timed(lambda: Aligner(rc2, 0.2, 15).locate(r1), min_reps=50) * 1e6,
           "new_Aligner_34_base_adapter_plus_locate": timed(lambda: Aligner(synth.TRUSEQ_34, 0.1, 14).locate(q), min_reps=50) * 1e6}
    from atropos_amd.adapters import AsciiSource, upper_ascii
    w4 = synth.workload("C4", 0, 1000, device="cuda")
    linked = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w4["max_error_rate"],
                            min_overlap=w4["min_overlap"], indel_cost=w4["indel_cost"]) for f, b in zip(w4["fronts"], w4["backs"])]
    lset = LinkedSet(linked)
    if lset.fused:
        batch = AsciiSource(upper_ascii(w4["reads"])).batch(lset.table_kind, lset.table)
        be = lset._backend
        out["linked_set_of_four_1000_reads"] = timed(lambda: be.linked_match_batch(lset._handle, batch.packed, batch.lens,
                                                                                   batch.nreads, batch.max_len)) * 1e6
    return out


def measure_pairs(n=1000):
    """The pair-wise entry points on the 1 000 pairs the reference's trim command hands over per call (C3 pairs, batches
    resident): InsertAligner.match_insert_batch and PairAligner.locate_batch (MergeOverlapping's aligner)."""
    from atropos_amd import _lib
    from atropos_amd.align import InsertAligner, PairAligner
    w = synth.workload("C3", 0, n, device="cuda")
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    pa = PairAligner(0.2, 15, revcomp_ref=True)
    be = _lib.get_backend()
    b1, b2 = ia.pack(w["reads1"]), ia.pack(w["reads2"], check=True)
    rb, qb = pa._pack(w["reads2"], _lib.TABLE_DNA15, be, True), pa._pack(w["reads1"], _lib.TABLE_DNA15, be, True)
    return {"pairs": n, "read_len": int(w["reads1"].shape[1]),
            "insert_match_us": timed(lambda: ia.match_insert_batch(b1, b2)) * 1e6,
            "pair_locate_us": timed(lambda: pa.locate_batch(rb, qb)) * 1e6,
            "pair_locate_full_sweep_us": timed(lambda: pa.locate_batch(rb, qb, path="full"), min_reps=3) * 1e6}


if __name__ == "__main__":
    res = measure()
    text = json.dumps(res)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write(text + "\n")
