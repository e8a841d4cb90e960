"""Device-resident FASTQ pipeline (index -> trim stages -> filters -> formatter) against the
output text of the reference's `atropos trim` command, with the device work done by the CPU
twins of the kernels (tests/emu/emu_fastq.cpp + the alignment emulation)."""
from . import _cases


def test_trim_pipeline_reference_cli_cases(emu_backend):
    assert _cases.check_trim_golden() >= 86


def test_trim_pipeline_through_the_two_pass_prepass(emu_backend, monkeypatch):
    """The same reference outputs with every batch, however short, packed as bit planes where the two-pass pre-pass
    takes the adapter (ragged batches: the reads come out of quality trimming / earlier adapters)."""
    from atropos_amd import _lib
    calls = []
    real = emu_backend.locate_planes_batch
    monkeypatch.setattr(_lib, "PLANES_MIN_READS", 1)
    monkeypatch.setattr(emu_backend, "locate_planes_batch", lambda *a: (calls.append(a[3]), real(*a))[1])
    assert _cases.check_trim_golden() >= 86
    assert len(calls) >= 10


def test_trim_file_chunking(emu_backend, tmp_path):
    counts = _cases.check_fastq_chunking(tmp_path)
    assert counts["keep"] > 0 and counts["too_short"] > 0


def test_pipeline_rejects_what_it_does_not_cover(emu_backend):
    import pytest
    from atropos_amd.trim import pipeline_from_args
    pipe = pipeline_from_args("-g ^ACGTACGT --no-indels")                # anchored without indels: compare_prefixes path
    assert pipe.trim_bytes(b"@r\nACGTACGTAA\n+\nIIIIIIIIII\n") == b"@r\nAA\n+\nII\n"
    with pytest.raises(NotImplementedError):
        pipeline_from_args("-a AAAA...TTTT -a GGGG")                     # linked + plain adapters mixed
    with pytest.raises(SystemExit):
        pipeline_from_args("-a ACGT --trim-primer")                      # modifier outside the pipeline
    with pytest.raises(NotImplementedError):
        pipeline_from_args("--aligner insert -a ACGTACGTAC -A ACGTACGTAC --length-tag length=")   # not with the insert aligner
    with pytest.raises(NotImplementedError):
        pipeline_from_args("-a ^ACGT...TTTT --info-file x")             # info file with a linked adapter


def test_paired_pipeline_reference_cli_cases(emu_backend):
    assert _cases.check_trim_golden_paired() >= 47


def test_paired_file_chunking(emu_backend, tmp_path):
    counts = _cases.check_paired_file_chunking(tmp_path)
    assert counts["keep"] > 0 and counts["too_short"] > 0


def test_fastq_reader_fuzz_vs_reference(emu_backend):
    total, errors = _cases.check_fastq_reader_golden()
    assert total == 300 and errors > 40


def test_paired_merge_slice_against_oracle(emu_backend, oracle):
    """MergeOverlapping restated on the checker (no code shared with the kernels) -- the CPU-tier size of
    tests/test_gpu_fastq.py::test_large_paired_merge_slice_against_oracle"""
    from .test_gpu_fastq import check_merge_slice_against_oracle
    assert check_merge_slice_against_oracle(oracle, 400, 2) > 150


def test_quality_trim_fixture(emu_backend):
    """Row f4 against the reference's outputs (qualtrim_fuzz.json.gz) -- CPU-tier twin of the GPU test of the same name"""
    from .test_gpu_fastq import check_quality_trim_fixture
    assert check_quality_trim_fixture() > 7000


def test_batch_slice_against_oracle_with_quality_trimming(emu_backend, oracle):
    from .test_gpu_fastq import check_slice_against_oracle
    trimmed, qtrimmed = check_slice_against_oracle(oracle, 3000, 1, "-q 15,20 --nextseq-trim 20 --trim-n")
    assert trimmed > 800 and qtrimmed > 1000


def test_chunked_reader_read_ahead_and_carry(emu_backend, tmp_path):
    """ChunkedFastqReader (round 6: file reads run READ_AHEAD chunks ahead of the carry): every record comes out once and
    in order -- over more chunks than staging buffers, with a caller that takes fewer records than a chunk holds (the
    surplus is carried over, as in a paired run), with records left over when the file is exhausted, and with a last line
    that has no line end."""
    import numpy as np
    from atropos_amd.fastq import ChunkedFastqReader
    rng = np.random.default_rng(11)
    recs = []
    for i in range(700):
        n = int(rng.integers(20, 90))
        seq = "".join("ACGT"[k] for k in rng.integers(0, 4, n))
        recs.append("@read%d some text\n%s\n+\n%s\n" % (i, seq, "I" * n))
    text = "".join(recs)[:-1]                                  # (no line end behind the last quality line)
    path = tmp_path / "in.fastq"
    path.write_bytes(text.encode())
    for take_all in (True, False):
        reader = ChunkedFastqReader(str(path), 4096, emu_backend)
        assert len(reader.buf) == ChunkedFastqReader.READ_AHEAD + 1
        got, chunks = [], 0
        try:
            while True:
                batch = reader.next_batch()
                chunks += 1
                if take_all:
                    head, consumed = batch, None
                else:                                           # two records fewer than the chunk holds, when it has them
                    head, consumed = batch.head(max(len(batch) - 2, min(len(batch), 1)))
                data = bytes(head.data[:head.nbytes].cpu().numpy().tobytes())
                nrec = len(head)
                end = reader.consumed if consumed is None else consumed
                got.append(data[:end])
                assert data[:end].count(b"\n") == 4 * nrec
                if reader.advance(consumed):
                    break
                assert chunks < 400
        finally:
            reader.close()
        assert chunks > 2 * (ChunkedFastqReader.READ_AHEAD + 1)
        assert b"".join(got) == (text + "\n").encode()
