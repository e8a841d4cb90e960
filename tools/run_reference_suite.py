#!/usr/bin/env python3
"""The boundary claim of INTEGRATION.md section 1, run: the REFERENCE'S OWN test-suite against this
implementation of `atropos.align`.

Build container only (needs /root/reference; nothing of it enters this repository or travels to the
GPU box).  The reference tree is copied to a scratch directory, its Cython modules are built there, ONE
line is changed -- atropos/align/__init__.py:6, the import of the native module, is pointed at
`atropos_amd.align` (the swap INTEGRATION.md describes) -- and `pytest tests/` of the reference runs in a
subprocess with this repository on the path.  Without a GPU the alignment kernels run through the CPU
twin of tests/emu (installed by a conftest.py written into the scratch copy); with --hip they run on the
GPU (`atropos_amd._lib.get_backend()`).

    python tools/run_reference_suite.py [--hip] [--keep] [pytest args ...]

Prints pytest's summary line; exit code = pytest's.  Expected here: 221 passed, 1 skipped -- the same as
the unpatched reference.
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
ORIGINAL = "from atropos.align._align import Aligner, MultiAligner, compare_prefixes, locate"
SWAPPED = "from atropos_amd.align import Aligner, MultiAligner, compare_prefixes, locate"

CONFTEST_EMU = '''\
# written by tools/run_reference_suite.py: the alignment kernels through the CPU twin (no GPU in this container)
import importlib.util
import sys
sys.path.insert(0, %r)
from atropos_amd import _lib
# (by path: the reference's own `tests` package has the name of this repository's)
_spec = importlib.util.spec_from_file_location("atropos_amd_emu_backend", %r)
_emu = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_emu)
_lib.set_backend(_emu.EmuBackend(), _test_double=True)
''' % (ROOT, os.path.join(ROOT, "tests", "emu", "backend.py"))

CONFTEST_HIP = '''\
# written by tools/run_reference_suite.py: the alignment kernels on the GPU
import sys
sys.path.insert(0, %r)
from atropos_amd import _lib
assert _lib.get_backend().name == "hip"
''' % ROOT


def main(argv):
    hip = "--hip" in argv
    keep = "--keep" in argv
    extra = [a for a in argv if a not in ("--hip", "--keep")]
    if not os.path.isdir(REFERENCE):
        raise SystemExit("%s is not here: this harness runs in the build container only" % REFERENCE)
    scratch = tempfile.mkdtemp(prefix="atropos_ref_suite_")
    try:
        for item in ("atropos", "tests", "setup.py", "versioneer.py", "setup.cfg", "README.md", "pytest.ini"):
            src = os.path.join(REFERENCE, item)
            if os.path.isdir(src):
                shutil.copytree(src, os.path.join(scratch, item))
            elif os.path.exists(src):
                shutil.copy(src, scratch)
        # the reference's other Cython modules (_qualtrim, _seqio) and _align itself (imported by nobody after the swap)
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "-i"], cwd=scratch,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        init = os.path.join(scratch, "atropos", "align", "__init__.py")
        text = open(init).read()
        if text.count(ORIGINAL) != 1:
            raise SystemExit("atropos/align/__init__.py does not hold the import line INTEGRATION.md names")
        with open(init, "w") as fh:
            fh.write(text.replace(ORIGINAL, SWAPPED))
        with open(os.path.join(scratch, "tests", "conftest.py"), "w") as fh:
            fh.write(CONFTEST_HIP if hip else CONFTEST_EMU)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([scratch, ROOT, os.environ.get("PYTHONPATH", "")]))
        cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "tests"] + extra
        proc = subprocess.run(cmd, cwd=scratch, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        lines = proc.stdout.strip().splitlines()
        print("\n".join(lines[-15:] if proc.returncode else lines[-3:]))
        print("backend: %s; swapped line: %s" % ("hip" if hip else "CPU twin (tests/emu)", SWAPPED))
        return proc.returncode
    finally:
        if keep:
            print("scratch copy kept at", scratch)
        else:
            shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
