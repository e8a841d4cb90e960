"""The run-time specialised pre-pass (csrc/jit.hpp, piece_spec.hip): same records as the generic kernel, the full
sweep and the oracle; the code-object cache."""
import os

import numpy as np
import pytest

from tests import _cases

pytestmark = pytest.mark.gpu


@pytest.fixture()
def jit_on(tmp_path, monkeypatch):
    monkeypatch.setenv("ATR_JIT", "1")
    monkeypatch.setenv("ATR_KCACHE_DIR", str(tmp_path / "kcache"))
    yield tmp_path / "kcache"


def test_specialised_kernel_c2(hip_backend, oracle, jit_on, monkeypatch):
    """C2's aligner, equal-length and ragged batches: specialised == generic == oracle, and the code object lands
    in the cache directory."""
    import torch
    from atropos_amd import synth
    from atropos_amd.align import Aligner
    n = 262144
    w = synth.workload("C2", 1 << 20, n, device="cuda:0")
    al = Aligner(w["adapter"], w["max_error_rate"], 14, False, False, w["min_overlap"], w["indel_cost"])
    assert al.prepare(150), "no specialised kernel (hiprtc?)"
    # (a process that has compiled this aligner's kernel before -- tests/test_gpu_locate.py run first -- serves it
    #  from memory and writes nothing)
    files = sorted(os.listdir(jit_on)) if jit_on.exists() else []
    assert len(files) <= 1 and all(f.endswith(".hsaco") for f in files), files
    planes = al.pack(w["reads"], layout="plane64")
    got = al.locate_batch(planes).numpy()[:, :6].astype(np.int32)
    monkeypatch.setenv("ATR_JIT", "0")
    gen = al.locate_batch(planes).numpy()[:, :6].astype(np.int32)
    monkeypatch.setenv("ATR_JIT", "1")
    assert np.array_equal(got, gen), "specialised pre-pass differs from the generic one"
    reads = w["reads"].cpu().numpy()
    m = 65536
    exp = oracle.locate_many(w["adapter"], reads[:m], np.full(m, 150, np.int32), w["max_error_rate"], 14, False, False,
                             w["min_overlap"], w["indel_cost"], max(2, min(16, os.cpu_count() or 2)))
    assert np.array_equal(got[:m], exp)
    assert (exp[:, 1] >= 0).sum() > 20000
    # ragged: lengths 100 .. 150 (the tail of every row cut off)
    rng = np.random.default_rng(5)
    lens = rng.integers(100, 151, size=m).astype(np.int32)
    rows = [bytes(reads[i, :lens[i]]).decode() for i in range(m)]
    assert al.prepare(150, ragged=True)
    pl = al.pack(rows, layout="plane64")
    assert pl.layout == "plane64" and pl.lens is not None
    got_r = al.locate_batch(pl).numpy()[:, :6].astype(np.int32)
    monkeypatch.setenv("ATR_JIT", "0")
    gen_r = al.locate_batch(pl).numpy()[:, :6].astype(np.int32)
    assert np.array_equal(got_r, gen_r)
    exp_r = oracle.locate_many(w["adapter"], reads[:m], lens, w["max_error_rate"], 14, False, False, w["min_overlap"],
                               w["indel_cost"], max(2, min(16, os.cpu_count() or 2)))
    assert np.array_equal(got_r, exp_r)
    assert len(os.listdir(jit_on)) == 2


def test_specialised_kernel_random_aligners(hip_backend, oracle, jit_on):
    """A kernel per aligner: random adapters of 20 .. 40 bases, error rates, wildcards, min_overlap, read lengths,
    equal-length and ragged batches -- specialised pre-pass == full sweep == one-pass pipeline == oracle."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 77, 14, 400,
                                                 lengths=(70, 100, 128, 150, 150, 160, 180, 200, 224, 250, 260, 288, 300))
    assert total > 2500
    assert len(os.listdir(jit_on)) >= 6
    # round 6: START_WITHIN_SEQ1 (flags 11 / 15)
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 79, 12, 400,
                                                 lengths=(100, 128, 150, 150, 160, 200, 250, 300), mrange=(20, 64),
                                                 flag_choices=(11, 15))
    assert total > 2000
    # round 6: adapters of 41 .. 64 bases (extended NARROW mode, up to eight body pieces, 96-column windows)
    total, refused = _cases.check_piece_pipeline(Aligner, oracle, _lib.AtroposHipError, 78, 12, 400,
                                                 lengths=(100, 128, 150, 150, 160, 200, 250, 300), mrange=(41, 64))
    assert total > 2500


def test_policy_off_and_auto(hip_backend, tmp_path, monkeypatch):
    """ATR_JIT=0: no kernel is built; auto: only on request (prepare) or for a long batch."""
    from atropos_amd.align import Aligner
    monkeypatch.setenv("ATR_KCACHE_DIR", str(tmp_path / "k"))
    monkeypatch.setenv("ATR_JIT", "0")
    al = Aligner("AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC", 0.1, 14, False, False, 3, 1)
    assert not al.prepare(150)
    monkeypatch.delenv("ATR_JIT")
    reads = ["ACGT" * 25] * 70000
    al.locate_batch(al.pack(reads, layout="plane64"))
    assert not os.path.exists(tmp_path / "k") or not os.listdir(tmp_path / "k")      # a 70 k batch does not compile anything
    assert al.prepare(100) and len(os.listdir(tmp_path / "k")) == 1


def test_certificates_on_device(hip_backend, oracle, jit_on):
    """The same pressure on the device, every aligner with its run-time specialised kernel (the certificates are
    constants of that kernel) -- and once more on the generic kernel."""
    from atropos_amd import _lib
    from atropos_amd.align import Aligner
    assert _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, 41, 10, count=3000) > 15000
    assert _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, 45, 8, count=2000, mrange=(41, 64)) > 8000
    import os
    os.environ["ATR_JIT"] = "0"
    try:
        assert _cases.check_certificates(Aligner, oracle, _lib.AtroposHipError, 43, 10, count=3000) > 15000
    finally:
        os.environ["ATR_JIT"] = "1"


def test_fused_ascii_entry(hip_backend, oracle, monkeypatch):
    """atr_locate_ascii_planes_batch (round 6): ASCII rows -> bit planes in registers -> pass A in ONE kernel.  Records and
    the packed batch it leaves behind equal atr_pack_planes + atr_locate_planes_batch: equal-length and ragged batches,
    batches that end inside a tile, rows wider than the reads, a matrix that does not start on a 16-byte boundary, other
    adapters / error rates, soft-masked and non-ACGT bytes; the switch that turns the fused kernel off; the oracle."""
    import torch
    from atropos_amd import synth
    from atropos_amd.align import Aligner
    rng = np.random.default_rng(23)
    n = 400_003
    w = synth.workload("C2", 5 << 20, n, device="cuda:0")
    reads = w["reads"]
    cases = [(w["adapter"], 0.1, 3), (synth.TRUSEQ_33, 0.12, 5), (synth.LINKED_BACKS[1], 0.08, 1)]
    checked = 0
    for adapter, e, mo in cases:
        al = Aligner(adapter, e, 14, False, False, mo, 1)
        for variant in ("equal", "ragged", "wide_rows", "unaligned", "dirty"):
            mat, lens, max_len = reads, None, 150
            if variant == "ragged":
                lens = torch.from_numpy(rng.integers(0, 151, n).astype(np.int32)).cuda()
            elif variant == "wide_rows":                              # 150-base reads in rows of 160 bytes
                mat = torch.zeros((n, 160), dtype=torch.uint8, device="cuda:0")
                mat[:, :150] = reads
                mat = mat[:, :150]                                    # (a view: stride 160)
            elif variant == "unaligned":                              # the first row starts 150 bytes into the buffer; n - 1 rows
                mat = reads[1:]
            elif variant == "dirty":
                mat = reads.clone()
                idx = torch.from_numpy(rng.integers(0, n * 150, 200_000)).cuda()
                mat.view(-1)[idx] = torch.from_numpy(rng.choice(np.frombuffer(b"acgtNnRYX.-*", np.uint8), 200_000)).cuda()
            k = mat.shape[0]
            res, left = al.locate_ascii(mat, lens, max_len)
            got = res.records
            ref_batch = al.pack(mat if lens is None else mat, layout="plane64") if lens is None else None
            if lens is None:
                want = al.locate_batch(ref_batch).records
                assert torch.equal(left.packed[:ref_batch.packed.numel()], ref_batch.packed), (adapter, variant)
            else:
                from atropos_amd.batch import ReadBatch
                ref_batch = ReadBatch.from_ascii(mat, lens, 150, al.table_kind, al._table, planes=True)
                want = al.locate_batch(ref_batch).records
                assert torch.equal(left.packed[:ref_batch.packed.numel()], ref_batch.packed), (adapter, variant)
            assert torch.equal(got, want), (adapter, variant)
            # the same through locate_batch's own routing (a long uint8 tensor on the device) and with the fused kernel off
            if variant == "equal":
                assert torch.equal(al.locate_batch(mat).records, want)
                sl = mat[:60_000].cpu().numpy()
                exp = oracle.locate_many(adapter, sl, np.full(len(sl), 150, np.int32), e, 14, False, False, mo, 1, 8)
                assert np.array_equal(got[:60_000, :6].cpu().numpy().astype(np.int32), exp)
            checked += k
    monkeypatch.setenv("ATR_JIT", "0")                                 # no fused kernel: the two-kernel form behind the same entry
    al = Aligner(w["adapter"], 0.1, 14, False, False, 3, 1)
    res, _ = al.locate_ascii(reads[:100_000].contiguous())
    assert torch.equal(res.records, al.locate_batch(al.pack(reads[:100_000].contiguous(), layout="plane64")).records)
    assert checked > 5_000_000
