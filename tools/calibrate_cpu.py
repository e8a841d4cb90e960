#!/usr/bin/env python3
"""Container-only calibration promised in BASELINE.md section 3: the reference's own Cython code and the
oracle's C restatement (the "port" CPU baseline bench.py times on the GPU box) on IDENTICAL reads, on
1 and on all cores of this container.  Needs /root/reference (built in a scratch directory by
tests/golden/make_golden.build_reference).

    python tools/calibrate_cpu.py            # prints one JSON object; figures go to BASELINE.md
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

N_SE, N_PE, N_LINKED = 200_000, 20_000, 100_000
_state = {}


def _rows(t):
    return [bytes(x.tolist()).decode("ascii") for x in t]


def _init_worker():
    from make_golden import build_reference
    build_reference("/tmp/atropos_ref_build")


def _cython_locate(args):
    from atropos.align import Aligner
    adapter, reads = args
    al = Aligner(adapter, 0.1, 14, False, False, 3, 1)
    return sum(1 for q in reads if al.locate(q) is not None)


def _cython_insert(args):
    from atropos.align import InsertAligner
    a1, a2, pairs = args
    ia = InsertAligner(a1, a2)
    return sum(1 for x, y in pairs if ia.match_insert(x, y) is not None)


def _cython_linked(args):
    from atropos.adapters import LinkedAdapter
    from atropos.io.seqio import Sequence
    fronts, backs, reads = args
    las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=0.12, min_overlap=3, indel_cost=1)
           for f, b in zip(fronts, backs)]
    hits = 0
    for q in reads:
        s = Sequence(name="r", sequence=q)
        hits += sum(1 for la in las if la.match_to(s) is not None)
    return hits


def _cython_c5(args):
    """C5 as bench.py's CPU leg times it: match_insert (read wildcards) and, for an insert match with errors, the
    liberal correction of the overlap (InsertAdapterCutter.__call__, modifiers.py:397-404, without the trimming)."""
    from atropos.adapters import Adapter, BACK
    from atropos.commands.trim.modifiers import InsertAdapterCutter
    from atropos.io.seqio import Sequence
    from atropos.util import RandomMatchProbability
    a1, a2, quads = args
    akw = dict(max_error_rate=0.2, min_overlap=1, indel_cost=3, match_probability=RandomMatchProbability(), max_rmp=1e-6,
               read_wildcards=True)
    cutter = InsertAdapterCutter(Adapter(a1, BACK, name="a1", **akw), Adapter(a2, BACK, name="a2", **akw), action='trim',
                                 mismatch_action='liberal', read_wildcards=True)
    hits = 0
    for r1, q1, r2, q2 in quads:
        s1, s2 = Sequence(name="p", sequence=r1, qualities=q1), Sequence(name="p", sequence=r2, qualities=q2)
        im = cutter.aligner.match_insert(r1, r2)
        if im is not None:
            hits += 1
            if im[0][5] > 0:
                cutter.correct_errors(s1, s2, im[0], truncate_seqs=True)
    return hits


def timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def main():
    import numpy as np
    from atropos_amd import synth
    from oracle import oracle as O
    _init_worker()
    cores = len(os.sched_getaffinity(0))
    out = {"cores": cores}
    # C2-like
    w = synth.workload("C2", 0, N_SE)
    reads = _rows(w["reads"])
    mat, lens = w["reads"].numpy(), np.full(N_SE, 150, np.int32)
    out["C2_cython_1core"] = N_SE / timed(lambda: _cython_locate((w["adapter"], reads)))
    with mp.Pool(cores, initializer=_init_worker) as pool:
        parts = [(w["adapter"], reads[i::cores]) for i in range(cores)]
        pool.map(_cython_locate, parts)
        out["C2_cython_allcores"] = N_SE / timed(lambda: pool.map(_cython_locate, parts))
    out["C2_port_1thread"] = N_SE / timed(lambda: O.locate_many(w["adapter"], mat, lens, 0.1, 14, False, False, 3, 1, 1))
    out["C2_port_allthreads"] = N_SE / timed(lambda: O.locate_many(w["adapter"], mat, lens, 0.1, 14, False, False, 3, 1, cores))
    # C3-like
    w = synth.workload("C3", 0, N_PE)
    r1, r2 = _rows(w["reads1"]), _rows(w["reads2"])
    pairs = list(zip(r1, r2))
    out["C3_cython_1core_pairs"] = N_PE / timed(lambda: _cython_insert((synth.PE_ADAPTER1, synth.PE_ADAPTER2, pairs)))
    with mp.Pool(cores, initializer=_init_worker) as pool:
        parts = [(synth.PE_ADAPTER1, synth.PE_ADAPTER2, pairs[i::cores]) for i in range(cores)]
        pool.map(_cython_insert, parts)
        out["C3_cython_allcores_pairs"] = N_PE / timed(lambda: pool.map(_cython_insert, parts))
    orc = O.InsertOracle(synth.PE_ADAPTER1, synth.PE_ADAPTER2)
    l = np.full(N_PE, 150, np.int32)
    m1, m2 = w["reads1"].numpy(), w["reads2"].numpy()
    out["C3_port_1thread_pairs"] = N_PE / timed(lambda: O.match_insert_many(orc, m1, l, m2, l, 1))
    out["C3_port_allthreads_pairs"] = N_PE / timed(lambda: O.match_insert_many(orc, m1, l, m2, l, cores))
    # C4-like
    w = synth.workload("C4", 0, N_LINKED)
    reads = _rows(w["reads"])
    mat, lens = w["reads"].numpy(), np.full(N_LINKED, 150, np.int32)
    out["C4_cython_1core"] = N_LINKED / timed(lambda: _cython_linked((w["fronts"], w["backs"], reads)))
    with mp.Pool(cores, initializer=_init_worker) as pool:
        parts = [(w["fronts"], w["backs"], reads[i::cores]) for i in range(cores)]
        pool.map(_cython_linked, parts)
        out["C4_cython_allcores"] = N_LINKED / timed(lambda: pool.map(_cython_linked, parts))
    out["C4_port_1thread"] = N_LINKED / timed(lambda: O.linked_many(w["fronts"], w["backs"], mat, lens, 0.12, 3, 1, True, False, 1))
    out["C4_port_allthreads"] = N_LINKED / timed(lambda: O.linked_many(w["fronts"], w["backs"], mat, lens, 0.12, 3, 1, True, False, cores))
    # C5-like: match + liberal correction (the oracle's orc_correct_errors, round 3)
    n5 = 20_000
    w = synth.workload("C5", 0, n5)
    quads = list(zip(_rows(w["reads1"]), _rows(w["quals1"]), _rows(w["reads2"]), _rows(w["quals2"])))
    out["C5_cython_1core_pairs"] = n5 / timed(lambda: _cython_c5((synth.PE_ADAPTER1, synth.PE_ADAPTER2, quads)))
    with mp.Pool(cores, initializer=_init_worker) as pool:
        parts = [(synth.PE_ADAPTER1, synth.PE_ADAPTER2, quads[i::cores]) for i in range(cores)]
        pool.map(_cython_c5, parts)
        out["C5_cython_allcores_pairs"] = n5 / timed(lambda: pool.map(_cython_c5, parts))
    orc5 = O.InsertOracle(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
    l5 = np.full(n5, 250, np.int32)
    m = [w[k].numpy() for k in ("reads1", "reads2", "quals1", "quals2")]

    def port5(threads):
        a1, a2, b1, b2 = (x.copy() for x in m)
        rec = O.match_insert_many(orc5, a1, l5, a2, l5, threads)
        O.insert_correct_many(rec, a1, b1, l5, a2, b2, l5, "liberal", 1, threads)
    out["C5_port_1thread_pairs"] = n5 / timed(lambda: port5(1))
    out["C5_port_allthreads_pairs"] = n5 / timed(lambda: port5(cores))
    out["C5_port_over_cython_1"] = out["C5_port_1thread_pairs"] / out["C5_cython_1core_pairs"]
    out["C5_port_over_cython_all"] = out["C5_port_allthreads_pairs"] / out["C5_cython_allcores_pairs"]
    for k in ("C2", "C4"):
        out[k + "_port_over_cython_1"] = out[k + "_port_1thread"] / out[k + "_cython_1core"]
        out[k + "_port_over_cython_all"] = out[k + "_port_allthreads"] / out[k + "_cython_allcores"]
    out["C3_port_over_cython_1"] = out["C3_port_1thread_pairs"] / out["C3_cython_1core_pairs"]
    out["C3_port_over_cython_all"] = out["C3_port_allthreads_pairs"] / out["C3_cython_allcores_pairs"]
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) and v < 100 else int(v)) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
