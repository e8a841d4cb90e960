"""Device-resident FASTQ pipeline on the real HIP kernels (through the C ABI): against the
output text of the reference's `atropos trim` command (committed golden cases), and -- at a
size where every kernel runs many blocks -- against the CPU twins of the kernels."""
import numpy as np
import pytest

from . import _cases

pytestmark = pytest.mark.gpu


def test_trim_pipeline_reference_cli_cases(hip_backend):
    assert _cases.check_trim_golden() >= 48


def test_trim_file_chunking(hip_backend, tmp_path):
    counts = _cases.check_fastq_chunking(tmp_path)
    assert counts["keep"] > 0 and counts["too_short"] > 0


def _big_fastq(nreads, seed):
    """nreads x 150 bp C2 reads with decaying qualities, ragged tails and N ends."""
    from atropos_amd import synth
    reads = synth.workload("C2", 0, nreads, device="cpu")["reads"].numpy()
    rng = np.random.RandomState(seed)
    n = reads.shape[1]
    qual = np.clip(38 - (np.arange(n)[None, :] * rng.uniform(0, 0.25, size=(nreads, 1))).astype(np.int64)
                   + rng.randint(-3, 4, size=(nreads, n)), 2, 40).astype(np.uint8) + 33
    lens = np.where(rng.rand(nreads) < 0.1, rng.randint(0, n + 1, size=nreads), n)
    nend = rng.rand(nreads) < 0.1
    parts = []
    for i in range(nreads):
        L = int(lens[i])
        seq = bytes(reads[i, :L])
        if nend[i] and L > 6:
            seq = b"NN" + seq[2:L - 3] + b"NNN"
        parts.append(b"@r%d len=%d\n%s\n+\n%s\n" % (i, L, seq, bytes(qual[i, :L])))
    return b"".join(parts)


@pytest.mark.parametrize("args", [
    "-a AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -q 15,20 --trim-n -m 20",
    "-b AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -n 2 --mask-adapter --max-n 0.2",
    "-a ^ACGT...AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC -e 0.15 -u 2 --discard-untrimmed",
])
def test_large_batch_equals_cpu_twin(hip_backend, args):
    from atropos_amd import _lib
    from atropos_amd.trim import pipeline_from_args
    from tests.emu.backend import EmuBackend
    data = _big_fastq(60000, 5)
    got = pipeline_from_args(args).trim_bytes(data)
    prev = _lib.set_backend(EmuBackend())
    try:
        want = pipeline_from_args(args).trim_bytes(data)
    finally:
        _lib.set_backend(prev)
    assert len(got) == len(want) and got == want, _cases._first_diff(got, want)
    assert len(got) > 20000


def test_paired_pipeline_reference_cli_cases(hip_backend):
    assert _cases.check_trim_golden_paired() >= 20


def test_fastq_reader_fuzz_vs_reference(hip_backend):
    total, errors = _cases.check_fastq_reader_golden()
    assert total == 300 and errors > 40
