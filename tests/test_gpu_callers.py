"""Boundary objects and callers with the device work done by the real HIP kernels,
against outputs recorded from the reference."""
import pytest

from . import _cases

pytestmark = pytest.mark.gpu


def test_match_to(hip_backend):
    assert _cases.check_match_to_golden() > 3000


def test_linked_adapters_c4(hip_backend):
    assert _cases.check_linked_c4() == 512


def test_adapter_cutter(hip_backend):
    assert _cases.check_cutter_golden() > 2500


def test_insert_adapter_cutter(hip_backend):
    assert _cases.check_insert_cutter_golden() > 1000


def test_reference_caller_kats(hip_backend):
    _cases.check_caller_kats()


def test_linked_sets_fused(hip_backend, oracle):
    """The fused linked-adapter pipeline (atr_linked_match_batch) on random sets of linked adapters:
    every (which, count, front, back) against the reference's rule restated on the oracle."""
    total, fused = _cases.check_linked_sets_against_oracle(oracle, 21, 400, reads_per_round=(1, 64, 65, 200, 700))
    assert total > 40000 and fused > 250


def test_c4_shard_against_oracle(hip_backend, oracle):
    """BASELINE config C4, one GPU's shard in full (12.5 M x 150 bp, four linked adapters, e = 0.12):
    a 250 k-read slice bit-exact against the oracle (LinkedAdapter.match_to = PREFIX match_to, then
    BACK match_to on read[front.rstop:], adapters/__init__.py:671-690), the whole shard against the
    step-wise device path (one kernel pipeline per adapter part, nothing shared with the fused
    kernels but the DP cores), and size-independent properties of all 12.5 M record pairs."""
    import numpy as np
    import torch
    from atropos_amd import synth
    from atropos_amd.adapters import (AsciiSource, LinkedAdapter, LinkedSet, _linked_records_stepwise, upper_ascii)
    n_total = 12_500_000
    chunks = [synth.workload("C4", lo, min(2_500_000, n_total - lo), device="cuda") for lo in range(0, n_total, 2_500_000)]
    w = chunks[0]
    reads = upper_ascii(torch.cat([c["reads"] for c in chunks]))
    del chunks
    las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=w["max_error_rate"],
                         min_overlap=w["min_overlap"], indel_cost=w["indel_cost"]) for f, b in zip(w["fronts"], w["backs"])]
    lset = LinkedSet(las)
    assert lset.fused
    src = AsciiSource(reads)
    which, count, front, back = lset.match_source(src)
    torch.cuda.synchronize()
    # (1) slice parity against the oracle
    lo = 7_654_321
    k = 250_000
    sl = reads[lo:lo + k].cpu().numpy()
    ew, ef, eb = oracle.linked_many(w["fronts"], w["backs"], sl, np.full(k, 150, np.int32), w["max_error_rate"],
                                    w["min_overlap"], w["indel_cost"], True, False, 8)
    assert np.array_equal(which[lo:lo + k].cpu().numpy(), ew[:, 0].astype(np.int32))
    assert np.array_equal(count[lo:lo + k].cpu().numpy(), ew[:, 1].astype(np.int32))
    assert np.array_equal(front[lo:lo + k, :6].cpu().numpy().astype(np.int32), ef)
    assert np.array_equal(back[lo:lo + k, :6].cpu().numpy().astype(np.int32), eb)
    assert (ew[:, 0] >= 0).mean() > 0.7 and (eb[:, 1] >= 0).mean() > 0.3
    # (2) the whole shard against the step-wise device path
    w2, c2, f2, b2 = _linked_records_stepwise(las, src)
    assert torch.equal(which, w2) and torch.equal(count, c2)
    assert torch.equal(front[:, :6], f2[:, :6]) and torch.equal(back[:, :6], b2[:, :6])
    # (3) invariants of every record pair
    fr, bk = front.to(torch.int32), back.to(torch.int32)
    hasf, hasb = fr[:, 1] >= 0, bk[:, 1] >= 0
    assert bool((hasf == (which >= 0)).all()) and bool(((count > 0) == (which >= 0)).all())
    assert bool((hasb <= hasf).all())                                  # no 3' match without a 5' match
    assert int((count > 1).sum().item()) == 0                          # the four 5' parts are mutually exclusive
    f = fr[hasf]
    assert bool((f[:, 0] == 0).all()) and bool((f[:, 2] == 0).all())   # anchored: starts at (0, 0)
    assert bool((f[:, 1] == 20).all())                                 # no STOP_WITHIN_SEQ1: the whole 5' adapter
    assert bool(((f[:, 3] >= 18) & (f[:, 3] <= 22)).all())             # at most k = 2 indels
    assert bool((f[:, 5] <= 2).all()) and bool((f[:, 4] + f[:, 5] >= 20).all())
    b, s = bk[hasb], fr[hasb][:, 3]
    mb = torch.tensor([len(x) for x in w["backs"]], device="cuda", dtype=torch.int32)[which[hasb].long()]
    assert bool((b[:, 0] == 0).all()) and bool(((b[:, 1] > 0) & (b[:, 1] <= mb)).all())
    assert bool(((b[:, 2] >= 0) & (b[:, 2] <= b[:, 3]) & (b[:, 3] <= 150 - s)).all())   # inside read[front.rstop:]
    assert bool(((b[:, 1] < mb) <= (b[:, 3] == 150 - s)).all())        # a partial 3' adapter only at the read end
    assert bool((b[:, 5] <= (b[:, 1].double() * 0.12).floor().to(torch.int32)).all())
    assert bool((bk[~hasb][:, [0, 2, 3, 4, 5]] == 0).all())
    # (4) determinism, and independence of the batch split
    w3, c3, f3, b3 = lset.match_source(src)
    assert torch.equal(which, w3) and torch.equal(front, f3) and torch.equal(back, b3)
    part = lset.match_source(AsciiSource(reads[lo:lo + 100_037].contiguous()))
    assert torch.equal(part[0], which[lo:lo + 100_037]) and torch.equal(part[3], back[lo:lo + 100_037])
    # (5) ragged lengths: the same reads cut to 60 .. 150 bases, a slice against the oracle
    g = torch.Generator(device="cuda").manual_seed(11)
    lens = torch.randint(60, 151, (1_000_000,), generator=g, device="cuda", dtype=torch.int32)
    sub = reads[:1_000_000].contiguous()
    rw, rc, rf, rb = lset.match_source(AsciiSource(sub, lens))
    ew, ef, eb = oracle.linked_many(w["fronts"], w["backs"], sub[:100_000].cpu().numpy(), lens[:100_000].cpu().numpy(),
                                    w["max_error_rate"], w["min_overlap"], w["indel_cost"], True, False, 8)
    assert np.array_equal(rw[:100_000].cpu().numpy(), ew[:, 0].astype(np.int32))
    assert np.array_equal(rf[:100_000, :6].cpu().numpy().astype(np.int32), ef)
    assert np.array_equal(rb[:100_000, :6].cpu().numpy().astype(np.int32), eb)


def test_linked_long_batches_mixed_sets(hip_backend, oracle):
    """Long batches (> 262 144 reads: per-block lists, offsets by atomics, one band launch, window launches by row class)
    of linked sets whose 3' aligners do NOT share a row class / indel mode -- a window launch per adapter, each reading
    its adapter's block of the set's device blob -- and of uniform sets of two and three adapters (one window launch):
    a slice against the oracle, the whole batch against the same reads in short batches (the wavefront-per-read finish
    or the window-word path: other kernels), equal-length and ragged."""
    import numpy as np
    import torch
    from atropos_amd import synth
    from atropos_amd.adapters import AsciiSource, LinkedAdapter, LinkedSet, upper_ascii
    rng = np.random.default_rng(97)

    def rseq(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))

    n = 300_000
    cases = [  # (5' length, 3' lengths, e, indel cost, min_overlap)
        (16, (20, 34, 45, 60), 0.1, 1, 3),          # four row classes
        (20, (34, 33, 12), 0.12, 1, 3),             # two classes
        (18, (34, 34), 0.1, 1, 3),                  # uniform pair: one window launch with two block rows
        (20, (33, 34, 36), 0.12, 1, 5),             # uniform triple
        (14, (30, 34, 40, 24), 0.1, 5, 3),          # indel cost above every k: no-indel windows, mixed classes
    ]
    ran = 0
    for ci, (fl, bls, e, indel, mo) in enumerate(cases):
        fronts, backs = [rseq(fl) for _ in bls], [rseq(b) for b in bls]
        reads = upper_ascii(synth.linked(ci * n, n, 150, fronts, backs, 0xC4C4 + ci, "cuda"))
        las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=e, min_overlap=mo,
                             indel_cost=indel) for f, b in zip(fronts, backs)]
        lset = LinkedSet(las)
        if not lset.fused:
            continue
        ran += 1
        for ragged in (False, True):
            lens = None
            if ragged:
                g = torch.Generator(device="cuda").manual_seed(5 + ci)
                lens = torch.randint(50, 151, (n,), generator=g, device="cuda", dtype=torch.int32)
            which, count, front, back = lset.match_source(AsciiSource(reads, lens))
            k = 60_000
            lo = 123_457
            ew, ef, eb = oracle.linked_many(fronts, backs, reads[lo:lo + k].cpu().numpy(),
                                            np.full(k, 150, np.int32) if lens is None else lens[lo:lo + k].cpu().numpy(),
                                            e, mo, indel, True, False, 8)
            assert np.array_equal(which[lo:lo + k].cpu().numpy(), ew[:, 0].astype(np.int32)), (ci, ragged)
            assert np.array_equal(count[lo:lo + k].cpu().numpy(), ew[:, 1].astype(np.int32)), (ci, ragged)
            assert np.array_equal(front[lo:lo + k, :6].cpu().numpy().astype(np.int32), ef), (ci, ragged)
            assert np.array_equal(back[lo:lo + k, :6].cpu().numpy().astype(np.int32), eb), (ci, ragged)
            assert (eb[:, 1] >= 0).mean() > 0.05 and (ew[:, 0] >= 0).mean() > 0.5
            for a in range(0, n, 100_000):
                part = lset.match_source(AsciiSource(reads[a:a + 100_000].contiguous(), None if lens is None else lens[a:a + 100_000].contiguous()))
                assert torch.equal(part[0], which[a:a + 100_000]) and torch.equal(part[1], count[a:a + 100_000]), (ci, ragged, a)
                assert torch.equal(part[2], front[a:a + 100_000]) and torch.equal(part[3], back[a:a + 100_000]), (ci, ragged, a)
    assert ran >= 4, ran


def test_linked_golden(hip_backend):
    total, fused = _cases.check_linked_golden()
    assert total == 3840 and fused > 60


def test_info_records(hip_backend):
    assert _cases.check_info_records() > 500


def test_c5_head(hip_backend):
    """2048 pairs of C5 with qualities against the reference: object path and device FASTQ pipeline."""
    assert _cases.check_c5_head() == 2048


def test_device_resident_adapters(hip_backend):
    assert _cases.check_device_resident_adapters() > 5000


def test_merge_overlapping(hip_backend):
    assert _cases.check_merge_golden(batch=True) == 1190


def test_long_multi_and_compare(hip_backend):
    """MultiAligner / compare_prefixes / compare_suffixes past 736 (and past 1 024 reference) characters vs the reference"""
    assert _cases.check_long_multi_compare() == 36 * 8 + 30 * 5 + 60


def _linked_group_check(oracle, reads, lens, fronts, backs, e, mo, ic, slice_lo, slice_k):
    """Grouped pipeline (5' parts at pack time, adapter-uniform plane64 sub-batches, the single-aligner two-pass pipeline
    per 3' adapter) against the fused tile64 pipeline on every read and against the oracle on a slice; the slot-order
    outputs against the batch-order ones."""
    import numpy as np
    import torch
    from atropos_amd.adapters import AsciiSource, LinkedAdapter, LinkedSet
    las = [LinkedAdapter(f, b, front_anchored=True, back_anchored=False, max_error_rate=e, min_overlap=mo, indel_cost=ic)
           for f, b in zip(fronts, backs)]
    lset = LinkedSet(las)
    assert lset.fused and lset.group_applies(reads.shape[1])
    groups = lset.pack_groups(reads, lens)
    which, count, front, back = lset.match_groups(groups)
    w2, c2, f2, b2 = lset.match_source(AsciiSource(reads, lens))
    assert torch.equal(which, w2) and torch.equal(count, c2)
    assert torch.equal(front[:, :6], f2[:, :6])
    assert torch.equal(back[:, :6], b2[:, :6])
    n = reads.shape[0]
    # the permutation and its inverse; group g's slots hold reads of adapter g in batch order
    perm, slot_of = groups.perm.cpu().numpy(), groups.slot_of.cpu().numpy()
    wh = which.cpu().numpy()
    assert np.array_equal(slot_of >= 0, wh >= 0)
    has = np.nonzero(slot_of >= 0)[0]
    assert np.array_equal(perm[slot_of[has]], has)
    info = [int(x) for x in groups.info]
    glens = groups.glens.cpu().numpy()
    ln = np.full(n, reads.shape[1], np.int64) if lens is None else lens.cpu().numpy().astype(np.int64)
    rstop = front[:, 3].cpu().numpy().astype(np.int64)
    for g in range(len(fronts)):
        cnt, t0 = info[g], info[4 + g]
        assert cnt == int((wh == g).sum())
        sl = perm[64 * t0:64 * t0 + cnt]
        assert np.array_equal(sl, np.nonzero(wh == g)[0])
        assert np.array_equal(glens[64 * t0:64 * t0 + cnt], (ln - rstop)[sl])
        pad = (-cnt) % 64
        assert (perm[64 * t0 + cnt:64 * t0 + cnt + pad] == -1).all() and (glens[64 * t0 + cnt:64 * t0 + cnt + pad] == 0).all()
    # slot-order records: the raw Aligner.locate of read[rstop:]; accepted ones equal the batch-order record
    slab, _ = lset.match_groups(groups, ordered=False)
    sb = slab[:, :6].cpu().numpy()[slot_of[has]]
    bb = back[:, :6].cpu().numpy()[has]
    acc = bb[:, 1] >= 0
    assert np.array_equal(sb[acc], bb[acc])
    # the oracle on a slice
    k = min(slice_k, n - slice_lo)
    sl = reads[slice_lo:slice_lo + k].cpu().numpy()
    sl_lens = np.full(k, reads.shape[1], np.int32) if lens is None else lens[slice_lo:slice_lo + k].cpu().numpy().astype(np.int32)
    ew, ef, eb = oracle.linked_many(fronts, backs, sl, sl_lens, e, mo, ic, True, False, 8)
    assert np.array_equal(wh[slice_lo:slice_lo + k], ew[:, 0].astype(np.int32))
    assert np.array_equal(front[slice_lo:slice_lo + k, :6].cpu().numpy().astype(np.int32), ef)
    assert np.array_equal(back[slice_lo:slice_lo + k, :6].cpu().numpy().astype(np.int32), eb)
    return int((wh >= 0).sum()), int((back[:, 1] >= 0).sum().item())


def test_linked_groups_c4(hip_backend, oracle):
    """C4's reads through atr_linked_group_pack / atr_linked_group_match (round 6): 3 M reads equal-length, 1 M ragged."""
    import torch
    from atropos_amd import synth
    from atropos_amd.adapters import upper_ascii
    n = 3_000_000
    w = synth.workload("C4", 1_000_000, n, device="cuda")
    reads = upper_ascii(w["reads"])
    nf, nb = _linked_group_check(oracle, reads, None, w["fronts"], w["backs"], w["max_error_rate"], w["min_overlap"],
                                 w["indel_cost"], 1_234_567, 150_000)
    assert nf > 0.7 * n and nb > 0.3 * n
    g = torch.Generator(device="cuda").manual_seed(11)
    sub = reads[:1_000_000].contiguous()
    lens = torch.randint(0, 151, (sub.shape[0],), generator=g, device="cuda", dtype=torch.int32)
    lens = torch.where(torch.rand(sub.shape[0], generator=g, device="cuda") < 0.7, torch.full_like(lens, 150), lens)
    nf, nb = _linked_group_check(oracle, sub, lens, w["fronts"], w["backs"], w["max_error_rate"], w["min_overlap"],
                                 w["indel_cost"], 500_000, 100_000)
    assert nf > 0.5 * sub.shape[0]


def test_linked_groups_small_and_odd(hip_backend, oracle):
    """Short batches, batches that end inside a tile, sets of one to three adapters, a group without reads, 5' parts with
    errors (the queued anchored DP), lower-case free input already folded, reads shorter than the 5' part."""
    import numpy as np
    import torch
    from atropos_amd import synth
    from atropos_amd.adapters import upper_ascii
    rng = np.random.default_rng(5)
    total = 0
    for nad, n in ((4, 1), (4, 63), (4, 65), (1, 1000), (2, 4097), (3, 70_001), (4, 300_017)):
        w = synth.workload("C4", 17, n, device="cuda")
        reads = upper_ascii(w["reads"])
        fronts, backs = w["fronts"][:nad], w["backs"][:nad]
        if nad == 3:                                   # a group nobody belongs to: an adapter no read starts with
            fronts = (fronts[0], "TTTTGGGGCCCCAAAATTGG", fronts[2])
        lens = None
        if n % 2 == 1 and n > 100:
            lens = torch.from_numpy(rng.integers(0, 151, n).astype(np.int32)).cuda()
        nf, _ = _linked_group_check(oracle, reads, lens, fronts, backs, 0.12, 3, 1, 0, min(n, 50_000))
        total += nf
    assert total > 100_000
