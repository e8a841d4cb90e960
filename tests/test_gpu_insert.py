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
