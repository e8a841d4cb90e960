"""GPU parity tests of the insert aligner kernel, the general MultiAligner kernel and
compare_prefixes/suffixes through the C ABI: bit-exact against oracle and fixtures."""
import numpy as np
import pytest
import torch

from . import _cases
from .conftest import load_golden

pytestmark = pytest.mark.gpu


def test_golden_insert(hip_backend):
    from atropos_amd.align import InsertAligner
    assert _cases.check_golden_insert(InsertAligner) > 2000


def test_golden_multi_and_compare(hip_backend):
    from atropos_amd import align
    assert _cases.check_golden_multi_compare(align) > 3000


def test_insert_batches(hip_backend, oracle):
    from atropos_amd.align import InsertAligner
    assert _cases.check_insert_batches_against_oracle(InsertAligner, oracle, 17, 54) > 2000


def test_synthetic_heads_c3_c5(hip_backend):
    from atropos_amd import synth
    from atropos_amd.align import InsertAligner
    heads = load_golden("synth_heads.json.gz")
    for name in ("C3", "C5"):
        w = synth.workload(name, 0, heads[name]["count"], device="cuda")
        ia = InsertAligner(w["adapter1"], w["adapter2"], **heads[name]["kw"])
        res = ia.match_insert_batch(w["reads1"], w["reads2"]).results()
        assert [_cases.norm_insert(r) for r in res] == heads[name]["out"]


def _rows(t):
    return [bytes(x.tolist()).decode("ascii") for x in t]


@pytest.mark.parametrize("name,count", [("C3", 10_000_000), ("C5", 2_000_000)])
def test_full_size_insert(hip_backend, oracle, name, count):
    """BASELINE configs C3 (10 M x 2x150) / a 2 M-pair shard of C5 (2x250, read
    wildcards) on the device: a slice bit-exact against the oracle plus
    size-independent properties of every record."""
    from atropos_amd import synth
    from atropos_amd.align import InsertAligner
    kw = dict(read_wildcards=True) if name == "C5" else {}
    n = 150 if name == "C3" else 250
    chunk = 2_000_000
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, **kw)
    orc = oracle.InsertOracle(synth.PE_ADAPTER1, synth.PE_ADAPTER2, **kw)
    nfound = 0
    for lo in range(0, count, chunk):
        w = synth.workload(name, lo, min(chunk, count - lo), device="cuda")
        b1, b2 = ia.pack(w["reads1"]), ia.pack(w["reads2"], check=True)
        rec = ia.match_insert_batch(b1, b2).records
        r = rec.to(torch.int32)
        found = r[:, 0, 1] >= 0
        f = r[found]
        nfound += int(found.sum().item())
        ins, m1, m2 = f[:, 0], f[:, 1], f[:, 2]
        j = ins[:, 3]
        # insert tuple shape (L-j, L, 0, j, j-c, c), error bound floor(j*0.2) and int(0.2*L)
        assert bool((ins[:, 1] == n).all()) and bool((ins[:, 2] == 0).all())
        assert bool((ins[:, 0] == n - j).all()) and bool((ins[:, 4] + ins[:, 5] == j).all())
        assert bool((ins[:, 5] <= (j.double() * 0.2).floor().to(torch.int32)).all())
        has = m1[:, 1] >= 0
        assert bool((has == (m2[:, 1] >= 0)).all())
        assert bool((has == (ins[:, 0] >= 1)).all())                     # overhang >= min_adapter_overlap
        a1, a2 = m1[has], m2[has]
        assert bool((a1[:, 2] == j[has]).all()) and bool((a2[:, 2] == j[has]).all())    # rstart == insert size
        assert bool((a1[:, 3] == n).all()) and bool((a1[:, 0] == 0).all())
        assert bool((a1[:, 4] + a1[:, 5] == a1[:, 1]).all()) and bool((a2[:, 4] + a2[:, 5] == a2[:, 1]).all())
        assert bool((a1[:, 1] == torch.minimum(n - j[has], torch.tensor(len(synth.PE_ADAPTER1), device="cuda"))).all())
        # absent records are all-zero apart from the -1 marker
        assert bool((r[~found][:, :, [0, 2, 3, 4, 5]] == 0).all())
        # determinism + independence of the batch split
        assert torch.equal(rec, ia.match_insert_batch(b1, b2).records)
        part = ia.match_insert_batch(w["reads1"][1000:1000 + 70_001], w["reads2"][1000:1000 + 70_001]).records
        assert torch.equal(part, rec[1000:1000 + 70_001])
        if lo == 0:
            k = 20_000 if name == "C3" else 8_000
            res = ia.match_insert_batch(w["reads1"][:k], w["reads2"][:k]).results()
            r1s, r2s = _rows(w["reads1"][:k].cpu()), _rows(w["reads2"][:k].cpu())
            for x, y, g in zip(r1s, r2s, res):
                exp = orc.match_insert(x, y)
                exp = None if exp is None else [list(exp[0]), None if exp[1] is None else list(exp[1]),
                                                None if exp[2] is None else list(exp[2])]
                assert _cases.norm_insert(g) == exp
        del w, b1, b2, rec, r
    assert 0.40 < nfound / count < 0.60


def test_plane_guided_correction(hip_backend):
    assert _cases.check_plane_guided_correction(n=200_000) == 7 * 200_000


def test_fused_match_correct(hip_backend):
    """atr_insert_match_correct_batch == atr_insert_match_batch + atr_insert_correct_batch, every output"""
    assert _cases.check_fused_match_correct(n=150_000) == 13 * 150_000
    for n in (1, 63, 191):                                  # (a partial last tile, blocks with idle waves)
        assert _cases.check_fused_match_correct(n=n, seed=n) == 13 * n


def test_correct_errors_fixture(hip_backend):
    assert _cases.check_correct_errors_fixture() == 4000


def test_c5_correction_at_size(hip_backend, oracle):
    """BASELINE config C5 at shard-piece size (2 M pairs 2 x 250 bp with qualities, read wildcards):
    insert match + liberal error correction in place (atr_insert_match_batch + atr_insert_correct_batch).
    The head against the reference's outputs (c5_head.json.gz), a slice of the insert matches against the
    oracle, and size-independent properties of all 2 M pairs."""
    import numpy as np
    import torch
    from atropos_amd import synth
    from atropos_amd.align import InsertAligner
    from atropos_amd.modifiers import COMP_TABLE
    from .conftest import load_golden
    n = 2_000_000
    w = synth.workload("C5", 0, n, device="cuda")
    ia = InsertAligner(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
    s1, s2, q1, q2 = (w[k].clone() for k in ("reads1", "reads2", "quals1", "quals2"))
    b1, b2 = ia.pack(s1), ia.pack(s2, check=True)
    res = ia.match_insert_batch(b1, b2)
    rec = res.records
    changed, newlen = hip_backend.insert_correct_batch(rec, s1, q1, None, s2, q2, None, 2, 1, COMP_TABLE, planes1=b1, planes2=b2)
    torch.cuda.synchronize()
    # (0) the fused call (what bench.py times for C5) on the same 2 M pairs: every output equal to the two calls'
    f1, f2, g1, g2 = (w[k].clone() for k in ("reads1", "reads2", "quals1", "quals2"))
    fres, fch, fnl = ia.match_insert_correct_batch(b1, b2, f1, g1, f2, g2, "liberal", 1)
    for x, y in ((fres.records, rec), (f1, s1), (f2, s2), (g1, q1), (g2, q2), (fch, changed), (fnl, newlen)):
        assert torch.equal(x, y)
    del f1, f2, g1, g2, fres, fch, fnl
    ins = rec[:, 0].to(torch.int32)
    found, errs = ins[:, 1] >= 0, ins[:, 5]
    ch = changed.to(torch.int64)
    # (1) the head against the reference: the trimmed outputs are prefixes of the corrected reads
    g = load_golden("c5_head.json.gz")
    h1, hq1, h2, hq2 = (t[:len(g["full"])].cpu().numpy() for t in (s1, q1, s2, q2))
    for k, (a, b) in enumerate(g["full"]):
        if not a[3]:
            continue            # no insert match: the cutter corrects from its adapter matches instead (check_c5_head covers it)
        for got_s, got_q, exp in ((h1[k], hq1[k], a), (h2[k], hq2[k], b)):
            assert bytes(got_s[:len(exp[0])]).decode() == exp[0] and bytes(got_q[:len(exp[1])]).decode() == exp[1], k
        assert int(ch[k, 0]) == a[2] and int(ch[k, 1]) == b[2]
    # (2) a slice of the insert matches against the oracle
    lo, k = 1_234_000, 20_000
    orc = oracle.InsertOracle(synth.PE_ADAPTER1, synth.PE_ADAPTER2, read_wildcards=True)
    lens = np.full(k, 250, np.int32)
    exp = oracle.match_insert_many(orc, w["reads1"][lo:lo + k].cpu().numpy(), lens, w["reads2"][lo:lo + k].cpu().numpy(), lens, 8)
    assert np.array_equal(rec[lo:lo + k, :, :6].cpu().numpy().astype(np.int32), exp)
    # (2b) the corrected bases AND qualities of two slices against the checker's correct_errors (pinned to the
    #      reference by correct_errors_fuzz.json.gz): 40 k pairs, byte for byte, plus the per-read counts
    for lo2 in (lo, 0):
        e1, e2, eq1, eq2 = (np.ascontiguousarray(w[key][lo2:lo2 + k].cpu().numpy()) for key in ("reads1", "reads2", "quals1", "quals2"))
        exp_rec = exp if lo2 == lo else oracle.match_insert_many(orc, e1, lens, e2, lens, 8)
        ech, enl = oracle.insert_correct_many(exp_rec, e1, eq1, lens, e2, eq2, lens, "liberal", 1, 8)
        assert int((ech.sum(axis=1) > 0).sum()) > k // 8
        for got_t, exp_m in ((s1, e1), (s2, e2), (q1, eq1), (q2, eq2)):
            assert np.array_equal(got_t[lo2:lo2 + k].cpu().numpy(), exp_m)
        assert np.array_equal(changed[lo2:lo2 + k].cpu().numpy(), ech) and np.array_equal(newlen[lo2:lo2 + k].cpu().numpy(), enl)
    # (3) properties of all pairs
    assert bool((ch >= 0).all())                                         # no pair failed (KeyError / IndexError / ValueError codes are < 0)
    untouched = ~(found & (errs > 0))
    assert bool((ch[untouched] == 0).all())                              # only pairs with an imperfect insert match are corrected
    assert bool((ch.sum(dim=1) <= errs.to(torch.int64)).all())           # never more changes than mismatches in the overlap
    same1, same2 = (s1 == w["reads1"]), (s2 == w["reads2"])
    assert bool((( ~same1).sum(dim=1) == ch[:, 0]).all()) and bool(((~same2).sum(dim=1) == ch[:, 1]).all())
    cols = torch.arange(250, device="cuda", dtype=torch.int32)[None, :]
    inside1 = (cols >= ins[:, 2:3]) & (cols < ins[:, 3:4])               # read 1 changes only inside [querystart, querystop)
    assert bool((same1 | inside1).all())
    inside2 = (cols >= 250 - ins[:, 1:2]) & (cols < 250 - ins[:, 0:1])   # read 2 only inside [len2 - refstop, len2 - refstart)
    assert bool((same2 | inside2).all())
    assert bool(((q1 == w["quals1"]) | ~same1).all()) and bool(((q2 == w["quals2"]) | ~same2).all())   # qualities move with bases
    assert bool((newlen == 250).all())
    # (4) correcting the corrected reads again: every overlap now has fewer or as many mismatches, and a second
    #     pass changes nothing where the first pass resolved every mismatch
    res2 = ia.match_insert_batch(ia.pack(s1), ia.pack(s2, check=True)).records[:, 0].to(torch.int32)
    both = found & (res2[:, 1] >= 0) & (res2[:, 0] == ins[:, 0]) & (res2[:, 3] == ins[:, 3])
    assert bool((res2[both][:, 5] <= errs[both]).all())
    assert float(both.float().mean().item()) > 0.4


def test_read2_validation(hip_backend):
    from atropos_amd.align import InsertAligner
    _cases.check_read2_validation(InsertAligner("TTAGACATATGG", "CAGTGGAGTATA"))


def test_multi_aligner_against_oracle(hip_backend, oracle):
    """multi_wave_kernel (a wavefront per pair, one Hamming distance per candidate) against the oracle, short and
    MiSeq-length sides."""
    from atropos_amd.align import MultiAligner
    assert _cases.check_multi_against_oracle(MultiAligner, oracle, 21, 60) == 2400
    assert _cases.check_multi_against_oracle(MultiAligner, oracle, 22, 12, npairs=30, top=320) == 360


@pytest.mark.gpu
def test_insert_list_cap_eight_chunks(hip_backend, oracle):
    """pairs with more listed overlap lengths than the kernel's per-width cap (12 for eight chunks, 16 otherwise)"""
    assert _cases.check_insert_list_cap(oracle) == 160
