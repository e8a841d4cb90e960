"""Loader for libatropos_hip.so (the C ABI in include/atropos_hip.h) and the
backend object the Python layer talks to.

There is exactly one product backend: the HIP library running on an MI355X.  If the
shared library has not been built, or no GPU is visible, every entry point raises --
there is no CPU fallback.  (The CPU test-suite injects a lock-step emulation of the
kernel built from the same per-lane source via ``set_backend``; that object lives
under tests/ and is never imported from here.)
"""
import contextlib
import ctypes as C
import os
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ATROPOS_HIP_LIB") or os.path.join(_HERE, "libatropos_hip.so")   # (override: kernel experiments)

ATR_OK = 0
ERRORS = {-1: "invalid argument", -2: "unsupported by the device kernels", -3: "HIP runtime error",
          -4: "out of memory", -5: "no HIP device"}
TABLE_DNA15, TABLE_ACGT, TABLE_IUPAC, TABLE_CUSTOM = 0, 1, 2, 3
MAX_REF_LEN = 128
MAX_READ_LEN = 736
MAX_LONG_READ_LEN = 32736                # Aligner.locate / locate_batch: reads up to here (the full sweep with a rolling origin base)
PLANES_MIN_READS = 65536                 # Aligner.pack: batches from here on are packed as bit planes when the two-pass
                                         # pre-pass takes the aligner (atr_locate_planes_batch)
WAVE_MAX_READS = 32768                   # atr_locate_batch: short batches take the wavefront-per-read kernel
LOCATE_PATHS = {"auto": 0, "full": 1, "filtered": 2, "wave": 3}   # ATR_LOCATE_* of include/atropos_hip.h
PAIRS_PATHS = {"auto": 0, "full": 1, "fast": 2, "wave": 3}        # ATR_PAIRS_*
PAIRS_MAX_LEN = 320
INSERT_MAX_ADAPTER = 128
INSERT_MAX_READ = 320


class InsertConfig(C.Structure):
    """atr_insert_config (include/atropos_hip.h)."""
    _fields_ = [
        ("adapter1", C.c_char_p), ("alen1", C.c_int),
        ("adapter2", C.c_char_p), ("alen2", C.c_int),
        ("insert_max_rmp", C.c_double), ("adapter_max_rmp", C.c_double),
        ("min_insert_overlap", C.c_int), ("max_insert_mismatch_frac", C.c_double),
        ("min_adapter_overlap", C.c_int), ("max_adapter_mismatch_frac", C.c_double),
        ("adapter_check_cutoff", C.c_int),
        ("adapter_wildcards", C.c_int), ("read_wildcards", C.c_int),
        ("rmp_insert", C.c_void_p), ("rmp_adapter", C.c_void_p), ("rmp_ld", C.c_int),
        ("max_mismatch_by_alen", C.c_void_p), ("n_mismatch", C.c_int),
    ]

class LinkedAdapterSpec(C.Structure):
    """atr_linked_adapter (include/atropos_hip.h)."""
    _fields_ = [
        ("front", C.c_void_p), ("back", C.c_void_p),
        ("front_exact_shortcut", C.c_int), ("back_exact_shortcut", C.c_int),
        ("d_front_rmp", C.c_void_p), ("d_back_rmp", C.c_void_p), ("front_rmp_ld", C.c_int), ("back_rmp_ld", C.c_int),
        ("front_max_rmp", C.c_double), ("back_max_rmp", C.c_double),
    ]


LINKED_MAX_ADAPTERS = 4

# prototypes of every symbol include/atropos_hip.h declares
PROTOTYPES = {
    "atr_version": (C.c_int, []),
    "atr_device_count": (C.c_int, []),
    "atr_last_error": (C.c_char_p, []),
    "atr_translate_table": (C.c_int, [C.c_int, C.c_char_p]),
    "atr_packed_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "atr_pack_reads": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_planes_count_uncoded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_pack_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_multi_locate_work_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "atr_multi_locate_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p]),
    "atr_compare_batch": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_locate_debug_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "atr_locate_debug": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_compare_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_adapter_postfilter": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                         C.c_double, C.c_int, C.c_void_p]),
    "atr_correct_errors_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_insert_correct_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_insert_match_correct_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                 C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_insert_aligner_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "atr_insert_aligner_destroy": (None, [C.c_void_p]),
    "atr_insert_match_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int, C.c_void_p, C.c_void_p]),
    "atr_insert_match_batch_coded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_case_sensitive_table": (C.c_int, [C.c_char_p]),
    "atr_aligner_create": (C.c_int, [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.POINTER(C.c_void_p)]),
    "atr_aligner_destroy": (None, [C.c_void_p]),
    "atr_aligner_set_min_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "atr_aligner_set_indel_cost": (C.c_int, [C.c_void_p, C.c_int]),
    "atr_aligner_query_table": (C.c_int, [C.c_void_p, C.c_char_p]),
    "atr_locate_work_bytes": (C.c_size_t, [C.c_int64]),
    "atr_locate_work_unresolved": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "atr_locate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "atr_locate_ascii_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_locate_one": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_insert_match_one": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_multi_locate_one": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_compare_one": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_locate_planes_applies": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "atr_aligner_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "atr_locate_ascii_planes_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_locate_planes_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_locate_batch_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "atr_linked_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "atr_linked_destroy": (None, [C.c_void_p]),
    "atr_linked_query_table": (C.c_int, [C.c_void_p]),
    "atr_linked_work_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "atr_linked_match_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_linked_group_applies": (C.c_int, [C.c_void_p, C.c_int]),
    "atr_linked_group_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "atr_linked_group_work_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "atr_linked_group_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_char_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "atr_linked_group_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_locate_pairs_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "atr_locate_pairs_need_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "atr_locate_pairs_path_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_int, C.c_void_p, C.c_void_p]),
    "atr_locate_pairs_long_work_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "atr_locate_pairs_long_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "atr_locate_pair_one": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_locate_pairs_full_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p]),
    "atr_fastq_work_bytes": (C.c_size_t, [C.c_int64]),
    "atr_fastq_count_lines": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_fastq_index": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "atr_pack_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_pair_filter_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "atr_insert_plan_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_clip_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "atr_quality_trim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]),
    "atr_nend_trim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p]),
    "atr_match_trim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    "atr_read_filter_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "atr_merge_work_bytes": (C.c_size_t, [C.c_int64]),
    "atr_merge_plan_batch": (C.c_int, [C.c_void_p] * 7 + [C.c_int64] + [C.c_void_p] * 5),
    "atr_merge_emit_batch": (C.c_int, [C.c_void_p] * 11 + [C.c_int64, C.c_int, C.c_int, C.c_char_p] + [C.c_void_p] * 5),
    "atr_fastq_emit_work_bytes": (C.c_size_t, [C.c_int64]),
    "atr_fastq_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

FASTQ_ERR_AT, FASTQ_ERR_PLUS, FASTQ_ERR_NAME2, FASTQ_ERR_LENGTH = 1, 2, 3, 4
DEST_KEEP, DEST_TOO_SHORT, DEST_TOO_LONG, DEST_TOO_MANY_N, DEST_TRIMMED, DEST_UNTRIMMED = range(6)
INT64_MAX = (1 << 63) - 1


class AtroposHipError(RuntimeError):
    pass


class AtroposUnsupported(AtroposHipError):
    """ATR_ERR_UNSUPPORTED: the request is outside the device kernels' envelope."""


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library and attach prototypes.  Raises if it is missing."""
    if not os.path.exists(path):
        raise AtroposHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C atropos_amd/csrc`). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def _check(lib, rc, what):
    if rc >= 0:
        return rc
    if rc == -1:
        raise ValueError("%s: invalid argument" % what)
    if rc == -4:
        raise MemoryError(what)
    detail = ERRORS.get(rc, "error %d" % rc)
    if rc == -2:
        raise AtroposUnsupported("%s: %s" % (what, detail))
    if rc == -3:
        detail += ": " + (lib.atr_last_error() or b"").decode("ascii", "replace")
    raise AtroposHipError("%s: %s" % (what, detail))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _PinnedUpload(object):
    """What HipBackend.stage_host_bytes hands to ReadBatch.from_ascii: `.to(device)` starts the DMA from the page-locked
    buffer and records the event the next staging waits for."""

    def __init__(self, host, backend):
        self.host, self.backend = host, backend
        self.dtype, self.shape = host.dtype, host.shape

    def dim(self):
        return self.host.dim()

    def to(self, device):
        dev = self.host.to(device, non_blocking=True)
        self.backend._pinned_event.record(torch.cuda.current_stream(self.backend.device))
        return dev


class LinkedGroups(object):
    """What atr_linked_group_pack leaves on the device for one batch: the per-adapter plane64 sub-batches of
    read[front.rstop:] (``grouped``, ``glens``), the permutation (``perm`` slot -> read, ``slot_of`` read -> slot), the 5'
    results (``which``, ``front``), the host ``info`` block (reads and first tile of every group) and the workspace."""
    __slots__ = ("nreads", "max_len", "grouped", "glens", "perm", "slot_of", "which", "front", "info", "work")

    def group_reads(self):
        return [int(self.info[g]) for g in range(LINKED_MAX_ADAPTERS)]


class HipBackend(object):
    """Thin object view of the C ABI; all buffers are torch tensors on one GPU."""

    name = "hip"

    def __init__(self, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise AtroposHipError("no HIP device visible to torch; the alignment kernels need an MI355X")
        ndev = _check(self.lib, self.lib.atr_device_count(), "atr_device_count")
        if ndev < 1:
            raise AtroposHipError("atr_device_count() == 0: no HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._works = {}                  # the kernels' scratch, one per stream calls are issued on (Aligner.locate_stream)
        self._side = []

    @property
    def _work(self):
        return self._works.get(torch.cuda.current_stream(self.device).cuda_stream)

    @_work.setter
    def _work(self, tensor):
        # one scratch per stream that issues calls, most recently used last; at most WORKS_KEPT of them are kept (a
        # 10 M-read workspace is ~1 GB: a backend whose callers come and go on fresh streams -- worker_context() -- would
        # otherwise pin one per stream it ever saw).  Dropping an entry frees nothing a kernel still
        # reads: the tensor's memory goes back to torch's caching allocator, which orders re-use on the stream it
        # was allocated on.
        key = torch.cuda.current_stream(self.device).cuda_stream
        self._works.pop(key, None)
        self._works[key] = tensor
        while len(self._works) > self.WORKS_KEPT:
            self._works.pop(next(iter(self._works)))

    WORKS_KEPT = 4

    def last_unresolved(self, nreads, n_adapters=1):
        """Reads the pre-pass of the last filtered call on the current stream left to the exact DP kernels
        (atr_locate_work_unresolved); None when no such call used this stream's workspace."""
        work = self._work
        if work is None:
            return None
        out = C.c_int64(0)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_work_unresolved(_ptr(work), int(nreads), int(n_adapters), self._stream(),
                                                                  C.byref(out)), "atr_locate_work_unresolved")
        return int(out.value)

    def stage_host_bytes(self, mat):
        """A uint8 ndarray [n, w] -> the same bytes in a page-locked host tensor (kept and re-used): the upload that
        follows is one DMA.  The previous upload from the buffer is waited for first."""
        need = int(mat.size)
        buf = getattr(self, "_pinned", None)
        if buf is None or buf.numel() < need:
            buf = self._pinned = torch.empty((max(need, 1 << 20),), dtype=torch.uint8).pin_memory()
            self._pinned_event = None
        if self._pinned_event is not None:
            self._pinned_event.synchronize()
        view = buf[:need].view(mat.shape)
        np.copyto(view.numpy(), mat)
        self._pinned_event = torch.cuda.Event()
        return _PinnedUpload(view, self)

    def side_streams(self, count):
        """``count`` streams of this backend's device, created once (Aligner.locate_stream issues consecutive batches
        on them in turn; every stream gets a workspace of its own)."""
        while len(self._side) < count:
            self._side.append(torch.cuda.Stream(device=self.device))
        return self._side[:count]

    # -- helpers ---------------------------------------------------------------
    @contextlib.contextmanager
    def worker_context(self):
        """Device + a stream of its own for a host thread that drives this GPU next to other
        threads driving theirs; the stream is synchronised on exit."""
        stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.device(self.device), torch.cuda.stream(stream):
            yield stream
            stream.synchronize()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # -- C ABI -----------------------------------------------------------------
    def translate_table(self, kind):
        buf = C.create_string_buffer(256)
        _check(self.lib, self.lib.atr_translate_table(kind, buf), "atr_translate_table")
        return buf.raw

    def packed_bytes(self, nreads, max_len):
        return self.lib.atr_packed_bytes(nreads, max_len)

    def pack_reads(self, ascii_2d, lens, max_len, table, count_invalid=False, starts=None, planes=False):
        """ascii_2d: uint8 [nreads, >=max_len] on self.device (row stride arbitrary);
        lens: int32 [nreads] or None; starts: int32 [nreads] or None (pack read[start:]);
        table: 256 bytes; planes: plane64 layout (the insert aligner's) instead of tile64.
        Returns the packed uint8 tensor (and, with count_invalid, the number of reads
        holding a byte the table maps to 0)."""
        nreads = ascii_2d.shape[0]
        packed = self.empty((max(self.packed_bytes(nreads, max_len), 16),), torch.uint8)
        invalid = torch.zeros((1,), dtype=torch.int32, device=self.device) if count_invalid else None
        if nreads and max_len:
            with torch.cuda.device(self.device):
                fn = self.lib.atr_pack_planes if planes else self.lib.atr_pack_reads
                _check(self.lib, fn(_ptr(ascii_2d), ascii_2d.stride(0), _ptr(lens), _ptr(starts), nreads, max_len, table,
                                    _ptr(packed), _ptr(invalid), self._stream()), "atr_pack_reads")
        return (packed, int(invalid.item())) if count_invalid else packed

    def planes_count_uncoded(self, planes, lens, other_lens, nreads, max_len):
        """Number of reads of a plane64 buffer with an uncoded base among their first
        min(lens, other_lens) bases (atr_planes_count_uncoded); either lens may be None (max_len)."""
        count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        if nreads and max_len:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_planes_count_uncoded(_ptr(planes), _ptr(lens), _ptr(other_lens), nreads, max_len,
                                                                   _ptr(count), self._stream()), "atr_planes_count_uncoded")
        return int(count.item())

    def multi_locate_batch(self, refs, ref_lens, queries, query_lens, e, flags, min_overlap, max_matches,
                           max_ref_len, out_stride):
        """refs/queries: uint8 [npairs, width] raw ASCII on the device; returns
        (records int16 [npairs, out_stride, 8], counts int32 [npairs])."""
        npairs = refs.shape[0]
        out = self.empty((npairs, out_stride, 8), torch.int16)
        counts = self.empty((npairs,), torch.int32)
        work = self.empty((max(self.lib.atr_multi_locate_work_bytes(npairs, max_ref_len), 4),), torch.uint8)
        if npairs:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_multi_locate_batch(
                    _ptr(refs), refs.stride(0), _ptr(ref_lens), _ptr(queries), queries.stride(0), _ptr(query_lens),
                    npairs, e, flags, min_overlap, max_matches, max_ref_len, _ptr(work), _ptr(out), _ptr(counts),
                    out_stride, self._stream()), "atr_multi_locate_batch")
        return out, counts

    def compare_batch(self, ref, queries, lens, max_len, wildcard_ref, wildcard_query, suffix):
        """ref: bytes; queries: uint8 [n, width] raw ASCII on the device."""
        n = queries.shape[0]
        out = self.empty((n, 8), torch.int16)
        if n:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_compare_batch(ref, len(ref), _ptr(queries), queries.stride(0), _ptr(lens),
                                                            n, max_len, int(wildcard_ref), int(wildcard_query),
                                                            int(suffix), _ptr(out), self._stream()),
                       "atr_compare_batch")
        return out

    def adapter_postfilter(self, records, m, min_overlap, max_error_rate, rmp, max_rmp, accept_full):
        """In-place acceptance test of Adapter.match_to on int16 [n, 8] records; rmp: float64
        [ld, ld] device tensor or None."""
        if records.shape[0]:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_adapter_postfilter(
                    _ptr(records), records.shape[0], m, min_overlap, max_error_rate, _ptr(rmp),
                    0 if rmp is None else rmp.shape[1], 0.0 if max_rmp is None else max_rmp, int(accept_full),
                    self._stream()), "atr_adapter_postfilter")
        return records

    def correct_errors_batch(self, seq1, qual1, lens1, seq2, qual2, lens2, insert, mask, action, min_qual_diff,
                             truncate, comp):
        """In-place error correction of the overlaps (uint8 [n, width] ASCII tensors on
        the device, same row stride); returns (changed int32 [n, 2], newlen int32 [n, 2])."""
        n = seq1.shape[0]
        changed = self.empty((n, 2), torch.int32)
        newlen = self.empty((n, 2), torch.int32)
        if n:
            if seq1.stride(0) != seq2.stride(0):
                raise ValueError("both reads need the same row stride")
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_correct_errors_batch(
                    _ptr(seq1), _ptr(qual1), _ptr(lens1), _ptr(seq2), _ptr(qual2), _ptr(lens2), seq1.stride(0),
                    _ptr(insert), _ptr(mask), n, seq1.shape[1], action, min_qual_diff, int(truncate), comp,
                    _ptr(changed), _ptr(newlen), self._stream()), "atr_correct_errors_batch")
        return changed, newlen

    def insert_correct_batch(self, records, seq1, qual1, lens1, seq2, qual2, lens2, action, min_qual_diff, comp,
                             changed=None, newlen=None, planes1=None, planes2=None):
        """Error correction of the pairs whose insert match (records of insert_match_batch) has
        errors, in place on uint8 [n, width] ASCII tensors; returns (changed, newlen) int32 [n, 2].
        planes1 / planes2: the plane64 ReadBatches the records were computed from (optional; the
        kernel then only visits the positions where the reads disagree)."""
        n = seq1.shape[0]
        changed = self.empty((n, 2), torch.int32) if changed is None else changed
        newlen = self.empty((n, 2), torch.int32) if newlen is None else newlen
        if n:
            if seq1.stride(0) != seq2.stride(0):
                raise ValueError("both reads need the same row stride")
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_insert_correct_batch(
                    _ptr(records), None if planes1 is None else _ptr(planes1.packed),
                    None if planes2 is None else _ptr(planes2.packed), 0 if planes1 is None else planes1.max_len,
                    _ptr(seq1), _ptr(qual1), _ptr(lens1), _ptr(seq2), _ptr(qual2), _ptr(lens2),
                    seq1.stride(0), n, seq1.shape[1], action, min_qual_diff, comp, _ptr(changed), _ptr(newlen),
                    self._stream()), "atr_insert_correct_batch")
        return changed, newlen

    def insert_aligner_create(self, cfg):
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_insert_aligner_create(C.addressof(cfg), C.byref(h)),
                   "atr_insert_aligner_create")
        return h

    def insert_aligner_destroy(self, h):
        self.lib.atr_insert_aligner_destroy(h)

    def insert_match_batch(self, h, packed1, lens1, packed2, lens2, npairs, max_len, cased=False):
        """cased: both reads are packed with ``case_sensitive_table()`` (soft-masked reads)."""
        out = self.empty((npairs, 3, 8), torch.int16)
        if npairs:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_insert_match_batch_coded(
                    h, _ptr(packed1), _ptr(lens1), _ptr(packed2), _ptr(lens2), npairs, max_len, 1 if cased else 0, _ptr(out),
                    self._stream()), "atr_insert_match_batch_coded")
        return out

    def insert_match_correct_batch(self, h, planes1, planes2, seq1, qual1, seq2, qual2, action, min_qual_diff, comp,
                                   changed=None, newlen=None):
        """insert_match_batch + insert_correct_batch as ONE kernel (atr_insert_match_correct_batch): the reads'
        planes are streamed once.  planes1 / planes2: plane64 ReadBatches whose lengths are also the lengths of
        the matrix rows.  Returns (records int16 [n, 3, 8], changed, newlen)."""
        n = planes1.nreads
        if planes1.max_len != planes2.max_len:
            raise ValueError("both read batches must be packed with the same max_len")
        if seq1.stride(0) != seq2.stride(0):
            raise ValueError("both reads need the same row stride")
        out = self.empty((n, 3, 8), torch.int16)
        changed = self.empty((n, 2), torch.int32) if changed is None else changed
        newlen = self.empty((n, 2), torch.int32) if newlen is None else newlen
        if n:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_insert_match_correct_batch(
                    h, _ptr(planes1.packed), _ptr(planes1.lens), _ptr(planes2.packed), _ptr(planes2.lens), n, planes1.max_len,
                    _ptr(out), _ptr(seq1), _ptr(qual1), _ptr(seq2), _ptr(qual2), seq1.stride(0), action, min_qual_diff, comp,
                    _ptr(changed), _ptr(newlen), self._stream()), "atr_insert_match_correct_batch")
        return out, changed, newlen

    def case_sensitive_table(self):
        buf = C.create_string_buffer(256)
        _check(self.lib, self.lib.atr_case_sensitive_table(buf), "atr_case_sensitive_table")
        return buf.raw

    def aligner_create(self, ref, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost):
        h = C.c_void_p()
        _check(self.lib, self.lib.atr_aligner_create(ref, len(ref), e, flags, int(wildcard_ref),
                                                     int(wildcard_query), min_overlap, indel_cost, C.byref(h)),
               "atr_aligner_create")
        return h

    def aligner_destroy(self, h):
        self.lib.atr_aligner_destroy(h)

    def aligner_set_min_overlap(self, h, v):
        _check(self.lib, self.lib.atr_aligner_set_min_overlap(h, v), "atr_aligner_set_min_overlap")

    def aligner_set_indel_cost(self, h, v):
        _check(self.lib, self.lib.atr_aligner_set_indel_cost(h, v), "atr_aligner_set_indel_cost")

    def aligner_query_table(self, h):
        buf = C.create_string_buffer(256)
        kind = _check(self.lib, self.lib.atr_aligner_query_table(h, buf), "atr_aligner_query_table")
        return kind, buf.raw

    def locate_batch(self, h, packed, lens, nreads, max_len, filtered=True, path=None):
        """path: one of LOCATE_PATHS ("auto": the fastest applicable kernels -- a wavefront per read for short
        batches, the filtered pipeline (bit-parallel pre-pass + windowed DP) for long ones; "full", "filtered",
        "wave": that kernel family); the records are identical on every path.  ``filtered=False`` is "full"."""
        if path is None:
            path = "auto" if filtered else "full"
        code = LOCATE_PATHS[path]
        out = self.empty((nreads, 8), torch.int16)
        if nreads:
            work = None
            if code in (0, 2) and (code == 2 or nreads > WAVE_MAX_READS):
                need = self.lib.atr_locate_work_bytes(nreads)
                if self._work is None or self._work.numel() < need:
                    self._work = self.empty((need,), torch.uint8)
                work = self._work
            elif code == 0:
                work = self._work if self._work is not None else self._small_work()
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_locate_batch_path(h, _ptr(packed), _ptr(lens), nreads, max_len, _ptr(out),
                                                                _ptr(work), code, self._stream()), "atr_locate_batch")
        return out

    def locate_planes_applies(self, h, max_len, ragged=False):
        """Is this aligner on reads of max_len bases (ragged: of at most max_len) inside the envelope of the
        two-pass pre-pass (plane64 reads, atr_locate_planes_batch)?"""
        return bool(self.lib.atr_locate_planes_applies(h, int(max_len), int(bool(ragged))))

    def aligner_prepare(self, h, max_len, ragged=False):
        """atr_aligner_prepare: build / load the pre-pass kernel specialised for this aligner and read length on this
        backend's device.  True: ready; False: there is none (the generic kernel serves the calls)."""
        with torch.cuda.device(self.device):
            rc = self.lib.atr_aligner_prepare(h, int(max_len), int(bool(ragged)))
        if rc not in (0, -2):                             # (ATR_ERR_UNSUPPORTED: no specialised kernel, not an error)
            _check(self.lib, rc, "atr_aligner_prepare")
        return rc == 0

    def locate_ascii_planes_batch(self, h, ascii_2d, lens, max_len, planes=None):
        """atr_locate_ascii_planes_batch: a long batch of ASCII rows -> (records int16 [n, 8], the packed plane64 buffer the
        call wrote) in one pre-pass launch + the DP kernels."""
        nreads = ascii_2d.shape[0]
        out = self.empty((nreads, 8), torch.int16)
        if planes is None:
            planes = self.empty((max(self.packed_bytes(nreads, max_len), 16),), torch.uint8)
        if nreads:
            need = self.lib.atr_locate_work_bytes(nreads)
            if self._work is None or self._work.numel() < need:
                self._work = self.empty((need,), torch.uint8)
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_locate_ascii_planes_batch(h, _ptr(ascii_2d), ascii_2d.stride(0), _ptr(lens), nreads,
                                                                        int(max_len), _ptr(planes), _ptr(out), _ptr(self._work),
                                                                        self._stream()), "atr_locate_ascii_planes_batch")
        return out, planes

    def locate_planes_batch(self, h, planes, lens, nreads, max_len):
        """Batched locate on a plane64 batch (atr_locate_planes_batch); lens None: equal-length reads."""
        out = self.empty((nreads, 8), torch.int16)
        if nreads:
            need = self.lib.atr_locate_work_bytes(nreads)
            if self._work is None or self._work.numel() < need:
                self._work = self.empty((need,), torch.uint8)
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_locate_planes_batch(h, _ptr(planes), _ptr(lens), nreads, max_len, _ptr(out), _ptr(self._work),
                                                                  self._stream()), "atr_locate_planes_batch")
        return out

    def _tls_buffer(self, name, make, fits=None):
        """The result buffer of a one-object call, one per host thread: ctypes releases the GIL during the call and
        the C side stages per thread, so two threads on one backend must not share the record they read back."""
        tls = self.__dict__.get("_tls")
        if tls is None:
            import threading
            tls = self.__dict__.setdefault("_tls", threading.local())
        buf = getattr(tls, name, None)
        if buf is None or (fits is not None and not fits(buf)):
            buf = make()
            setattr(tls, name, buf)
        return buf

    def locate_one(self, h, query):
        """``Aligner.locate`` of ONE read (bytes) -- what the module swap of INTEGRATION.md section 1 calls per
        read: atr_locate_one (the kernel reads the read from a page-locked staging buffer and writes the record
        into one; one launch, one synchronisation, no allocation).  Returns the six numbers or None."""
        rec = self._tls_buffer("one_rec", lambda: (C.c_int16 * 8)())
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_one(h, query, len(query), C.addressof(rec), self._stream()), "atr_locate_one")
        return None if rec[1] < 0 else (rec[0], rec[1], rec[2], rec[3], rec[4], rec[5])

    def multi_locate_one(self, ref, query, e, flags, min_overlap, max_matches):
        """``MultiAligner.locate`` of ONE pair of byte strings (atr_multi_locate_one); list of 6-tuples or None."""
        cap = max_matches + len(ref) + 2              # the last-column scan appends past max_matches (_align.pyx:750-763)
        buf = self._tls_buffer("multi_buf", lambda: (C.c_int16 * (cap * 8))(), lambda b: len(b) >= cap * 8)
        cnt = self._tls_buffer("multi_cnt", C.c_int32)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_multi_locate_one(ref, len(ref), query, len(query), e, flags, min_overlap, max_matches,
                                                           C.addressof(buf), cap, C.addressof(cnt), self._stream()),
                   "atr_multi_locate_one")
        c = cnt.value
        return None if c == 0 else [tuple(buf[8 * t:8 * t + 6]) for t in range(c)]

    def compare_one(self, ref, query, wildcard_ref, wildcard_query, suffix):
        """compare_prefixes / compare_suffixes of one pair of byte strings (atr_compare_one); the 6-tuple."""
        rec = self._tls_buffer("cmp_rec", lambda: (C.c_int16 * 8)())
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_compare_one(ref, len(ref), query, len(query), int(wildcard_ref), int(wildcard_query),
                                                      int(suffix), C.addressof(rec), self._stream()), "atr_compare_one")
        return (rec[0], rec[1], rec[2], rec[3], rec[4], rec[5])

    def locate_pair_one(self, ref_codes, revcomp_ref, query_codes, e, flags, wildcard_ref, wildcard_query, min_overlap,
                        indel_cost):
        """``Aligner(ref, ...).locate(query)`` for ONE pair of translated byte strings (atr_locate_pair_one)."""
        rec = self._tls_buffer("pair_rec", lambda: (C.c_int16 * 8)())
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_pair_one(ref_codes, len(ref_codes), int(revcomp_ref), query_codes,
                                                          len(query_codes), e, flags, int(wildcard_ref), int(wildcard_query),
                                                          min_overlap, indel_cost, C.addressof(rec), self._stream()),
                   "atr_locate_pair_one")
        return None if rec[1] < 0 else (rec[0], rec[1], rec[2], rec[3], rec[4], rec[5])

    def insert_match_one(self, h, seq1, seq2):
        """``InsertAligner.match_insert`` of ONE pair of byte strings (upper-case IUPAC letters only: the caller checks);
        the three records as a flat list of 24 int16 values (atr_insert_match_one)."""
        rec = self._tls_buffer("ins_rec", lambda: (C.c_int16 * 24)())
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_insert_match_one(h, seq1, len(seq1), seq2, len(seq2), C.addressof(rec), self._stream()),
                   "atr_insert_match_one")
        return rec

    def locate_ascii_batch(self, h, ascii_2d, lens, max_len):
        """atr_locate_ascii_batch: a short batch of ASCII rows (uint8 [n, width] on the device, row stride a multiple of
        four) through the wavefront-per-read kernel without packing; int16 [n, 8] records."""
        n = ascii_2d.shape[0]
        out = self.empty((n, 8), torch.int16)
        if n:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_locate_ascii_batch(h, _ptr(ascii_2d), ascii_2d.stride(0), _ptr(lens), n, max_len,
                                                                 _ptr(out), self._stream()), "atr_locate_ascii_batch")
        return out

    def _small_work(self):
        """Scratch for the short batches that do not take the wave kernel (references of more than 64 bases never
        use it, anchored prefixes only need it to be there)."""
        self._work = self.empty((self.lib.atr_locate_work_bytes(WAVE_MAX_READS),), torch.uint8)
        return self._work

    # -- linked adapters (one fused pipeline for the whole set) ---------------------
    def linked_create(self, specs):
        """specs: list of LinkedAdapterSpec.  Returns the set handle; AtroposHipError
        ("unsupported") when the set is outside the fused kernels' envelope."""
        arr = (LinkedAdapterSpec * len(specs))(*specs)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_linked_create(C.addressof(arr), len(specs), C.byref(h)), "atr_linked_create")
        return h

    def linked_destroy(self, h):
        self.lib.atr_linked_destroy(h)

    def linked_match_batch(self, h, packed, lens, nreads, max_len):
        """(which int8 [n, 2] = (first matching adapter | -1, number of matching 5' parts),
        front int16 [n, 8], back int16 [n, 8]) -- include/atropos_hip.h, atr_linked_match_batch."""
        which = self.empty((nreads, 2), torch.int8)
        front = self.empty((nreads, 8), torch.int16)
        back = self.empty((nreads, 8), torch.int16)
        if nreads:
            need = self.lib.atr_linked_work_bytes(h, nreads)
            if self._work is None or self._work.numel() < need:
                self._work = self.empty((need,), torch.uint8)
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_linked_match_batch(h, _ptr(packed), _ptr(lens), nreads, max_len, _ptr(which),
                                                                 _ptr(front), _ptr(back), _ptr(self._work), self._stream()),
                       "atr_linked_match_batch")
        return which, front, back

    def linked_group_applies(self, h, max_len):
        return bool(self.lib.atr_linked_group_applies(h, int(max_len)))

    def linked_group_pack(self, h, ascii_2d, lens, max_len, table):
        """atr_linked_group_pack: the 5' parts decided from the ASCII rows, read[front.rstop:] packed as bit planes into
        a sub-batch per adapter.  Returns a ``LinkedGroups`` (device buffers + the host info block); waits for the stream."""
        nreads = ascii_2d.shape[0]
        g = LinkedGroups()
        g.nreads, g.max_len = nreads, int(max_len)
        g.grouped = self.empty((max(self.lib.atr_linked_group_bytes(nreads, max_len), 16),), torch.uint8)
        g.glens = self.empty((nreads + 64 * LINKED_MAX_ADAPTERS,), torch.int32)
        g.perm = self.empty((nreads + 64 * LINKED_MAX_ADAPTERS,), torch.int32)
        g.slot_of = self.empty((nreads,), torch.int32)
        g.which = self.empty((nreads, 2), torch.int8)
        g.front = self.empty((nreads, 8), torch.int16)
        g.info = (C.c_int64 * 8)()
        g.work = self.empty((max(self.lib.atr_linked_group_work_bytes(h, nreads), 16),), torch.uint8)
        if nreads:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_linked_group_pack(h, _ptr(ascii_2d), ascii_2d.stride(0), _ptr(lens), nreads, int(max_len),
                                                                table, _ptr(g.grouped), _ptr(g.glens), _ptr(g.perm), _ptr(g.slot_of),
                                                                _ptr(g.which), _ptr(g.front), g.info, _ptr(g.work), self._stream()),
                       "atr_linked_group_pack")
        return g

    def linked_group_match(self, h, g, ordered=True):
        """atr_linked_group_match on a ``LinkedGroups``: (slab int16 [slots, 8] raw 3' records in slot order, back int16
        [n, 8] in batch order after the acceptance test -- None unless ``ordered``)."""
        slots = g.glens.shape[0]
        slab = self.empty((slots, 8), torch.int16)
        back = self.empty((g.nreads, 8), torch.int16) if ordered else None
        if g.nreads:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_linked_group_match(h, _ptr(g.grouped), _ptr(g.glens), g.info, g.max_len, _ptr(g.slot_of),
                                                                 _ptr(g.which), g.nreads, _ptr(slab), _ptr(back), _ptr(g.work),
                                                                 self._stream()), "atr_linked_group_match")
        return slab, back

    def locate_debug(self, h, packed, m, n):
        """atr_locate_debug: (cost matrix int32 [m + 1, n + 1] on the host, INT32_MIN = not computed; the record)."""
        work = self.empty((max(self.lib.atr_locate_debug_bytes(h, n), 16),), torch.uint8)
        out = self.empty((1, 8), torch.int16)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_debug(h, _ptr(packed), n, _ptr(work), _ptr(out), self._stream()),
                   "atr_locate_debug")
        cells = (m + 1) * (n + 1)
        return work[:4 * cells].view(torch.int32).reshape(m + 1, n + 1).cpu(), out.cpu()

    def compare_packed(self, h, packed, lens, nreads, max_len, suffix):
        """atr_compare_packed: compare_prefixes / compare_suffixes of aligner ``h``'s reference against
        reads packed for it.  int16 [nreads, 8]."""
        out = self.empty((nreads, 8), torch.int16)
        if nreads:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_compare_packed(h, _ptr(packed), _ptr(lens), nreads, max_len, int(suffix),
                                                             _ptr(out), self._stream()), "atr_compare_packed")
        return out

    def locate_pairs_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                           query_max_len, npairs, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost,
                           need=None, path="auto"):
        """Aligner.locate with a per-pair reference; both sides tile64-packed.  int16 [npairs, 8].
        need: int32 [npairs] or None -- alignments with fewer matches may come back as None
        (atr_locate_pairs_need_batch).  path: PAIRS_PATHS ("auto": a wavefront per pair for short batches, the
        cost / threat / band pipeline or the full sweep for long ones; "full", "fast", "wave": that family)."""
        out = self.empty((npairs, 8), torch.int16)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_pairs_path_batch(
                _ptr(ref_packed), _ptr(ref_lens), ref_max_len, int(revcomp_ref), _ptr(query_packed), _ptr(query_lens),
                query_max_len, npairs, e, flags, int(wildcard_ref), int(wildcard_query), min_overlap, indel_cost,
                _ptr(need), PAIRS_PATHS[path], _ptr(out), self._stream()), "atr_locate_pairs_path_batch")
        return out

    def locate_pairs_long_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                                query_max_len, npairs, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost):
        """Pairs with a side beyond PAIRS_MAX_LEN (atr_locate_pairs_long_batch: 64-bit cells, the DP column in a
        workspace allocated here).  The caller keeps npairs * ref_max_len small (PairAligner chunks)."""
        out = self.empty((npairs, 8), torch.int16)
        if npairs:
            work = self.empty((max(self.lib.atr_locate_pairs_long_work_bytes(npairs, ref_max_len), 16),), torch.uint8)
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_locate_pairs_long_batch(
                    _ptr(ref_packed), _ptr(ref_lens), ref_max_len, int(revcomp_ref), _ptr(query_packed), _ptr(query_lens),
                    query_max_len, npairs, e, flags, int(wildcard_ref), int(wildcard_query), min_overlap, indel_cost,
                    _ptr(out), _ptr(work), self._stream()), "atr_locate_pairs_long_batch")
        return out

    def locate_pairs_full_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                                query_max_len, npairs, e, flags, min_overlap, indel_cost):
        """The same records by the full-matrix sweep alone (atr_locate_pairs_full_batch)."""
        out = self.empty((npairs, 8), torch.int16)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_locate_pairs_full_batch(
                _ptr(ref_packed), _ptr(ref_lens), ref_max_len, int(revcomp_ref), _ptr(query_packed), _ptr(query_lens),
                query_max_len, npairs, e, flags, 0, 0, min_overlap, indel_cost, _ptr(out), self._stream()),
                "atr_locate_pairs_full_batch")
        return out

    # -- device-resident FASTQ batch --------------------------------------------
    def fastq_index(self, data, nbytes):
        """data: uint8 device tensor holding nbytes of FASTQ text (16-byte aligned, readable up
        to the next multiple of 16).  Returns (records uint32 [nrec, 8], nlines, error word)."""
        work = self.empty((max(self.lib.atr_fastq_work_bytes(nbytes), 16),), torch.uint8)
        info = self.empty((2,), torch.int64)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_fastq_count_lines(_ptr(data), nbytes, _ptr(work), _ptr(info), self._stream()),
                   "atr_fastq_count_lines")
            nlines = int(info[0].item())
            line_ends = self.empty((max(nlines, 1),), torch.int32)
            records = self.empty((nlines // 4, 8), torch.int32)
            _check(self.lib, self.lib.atr_fastq_index(_ptr(data), nbytes, _ptr(work), _ptr(line_ends), nlines,
                                                      _ptr(records), C.c_void_p(info.data_ptr() + 8), self._stream()),
                   "atr_fastq_index")
            err = int(info[1].item())
        return records, line_ends, nlines, err

    def pack_records(self, data, records, begin, end, max_len, table, count_invalid=False, planes=False):
        n = records.shape[0]
        packed = self.empty((max(self.packed_bytes(n, max_len), 16),), torch.uint8)
        lens = self.empty((n,), torch.int32)
        invalid = torch.zeros((1,), dtype=torch.int32, device=self.device) if count_invalid else None
        if n:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.atr_pack_records(_ptr(data), _ptr(records), _ptr(begin), _ptr(end), n, max_len,
                                                           table, int(planes), _ptr(packed), _ptr(lens), _ptr(invalid),
                                                           self._stream()), "atr_pack_records")
        return (packed, lens, int(invalid.item())) if count_invalid else (packed, lens)

    def clip_batch(self, records, begin, end, front, back):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_clip_batch(_ptr(records), _ptr(begin), _ptr(end), begin.shape[0], front, back,
                                                     self._stream()), "atr_clip_batch")

    def quality_trim_batch(self, data, records, begin, end, cutoff_front, cutoff_back, base, nextseq):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_quality_trim_batch(_ptr(data), _ptr(records), _ptr(begin), _ptr(end),
                                                             begin.shape[0], cutoff_front, cutoff_back, base,
                                                             int(nextseq), self._stream()), "atr_quality_trim_batch")

    def nend_trim_batch(self, data, records, begin, end, ubegin=None, uend=None):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_nend_trim_batch(_ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin),
                                                          _ptr(uend), begin.shape[0], self._stream()),
                   "atr_nend_trim_batch")

    def match_trim_batch(self, matches, front, default_front, begin, end, active, matched):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_match_trim_batch(_ptr(matches), _ptr(front), default_front, _ptr(begin),
                                                           _ptr(end), _ptr(active), _ptr(matched), begin.shape[0],
                                                           self._stream()), "atr_match_trim_batch")

    def read_filter_batch(self, data, records, begin, end, ubegin, uend, matched, min_len, max_len, max_n,
                          discard_trimmed, discard_untrimmed, masks=False):
        """The destination byte per read, or (masks=True) the bit mask of the filters that fire."""
        out = self.empty((begin.shape[0],), torch.uint8)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_read_filter_batch(
                _ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin), _ptr(uend), _ptr(matched),
                begin.shape[0], min_len, max_len, max_n, int(discard_trimmed), int(discard_untrimmed),
                None if masks else _ptr(out), _ptr(out) if masks else None, self._stream()), "atr_read_filter_batch")
        return out

    def pair_filter_batch(self, mask1, mask2, min_affected):
        dest = self.empty((mask1.shape[0],), torch.uint8)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_pair_filter_batch(_ptr(mask1), _ptr(mask2), mask1.shape[0], min_affected,
                                                            _ptr(dest), self._stream()), "atr_pair_filter_batch")
        return dest

    def insert_plan_batch(self, insert, fb1, fb2, batch1, batch2, begin1, end1, begin2, end2, uend1, uend2,
                          min_insert_len, symmetric, trim_action, correct_action=-1, min_qual_difference=1, comp=None):
        """InsertAdapterCutter's decision logic (+ optional in-place error correction of the two
        FASTQ chunks).  Returns (matched1, matched2, corrected int32 [n, 2], error word)."""
        n = begin1.shape[0]
        m1, m2 = self.empty((n,), torch.uint8), self.empty((n,), torch.uint8)
        corrected = self.empty((n, 2), torch.int32)
        err = self.empty((1,), torch.int64)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_insert_plan_batch(
                _ptr(insert), _ptr(fb1), _ptr(fb2), _ptr(batch1.data), _ptr(batch1.records), _ptr(batch2.data),
                _ptr(batch2.records), _ptr(begin1), _ptr(end1), _ptr(begin2), _ptr(end2), _ptr(uend1), _ptr(uend2), n,
                min_insert_len, int(symmetric), trim_action, correct_action, min_qual_difference, comp, _ptr(m1),
                _ptr(m2), _ptr(corrected), _ptr(err), self._stream()), "atr_insert_plan_batch")
        return m1, m2, corrected, int(err.item())

    def merge_batch(self, align, need, insert_matched, batch1, batch2, begin1, end1, begin2, end2, correct_action=-1,
                    min_qual_difference=1, comp=None):
        """MergeOverlapping after the alignments (atr_merge_plan_batch + atr_merge_emit_batch): returns
        (kind uint8 [n] -- 0: the pair stays a pair --, the FASTQ text of the merged reads in input order,
        corrected int32 [n, 2], error word).  With a mismatch action the two chunks are corrected in place."""
        n = begin1.shape[0]
        kind = self.empty((n,), torch.uint8)
        offsets = self.empty((n + 1,), torch.int64)
        corrected = self.empty((n, 2), torch.int32)
        corrected.zero_()
        err = self.empty((1,), torch.int64)
        work = self.empty((max(self.lib.atr_merge_work_bytes(n), 16),), torch.uint8)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.atr_merge_plan_batch(
                _ptr(align), _ptr(need), _ptr(batch1.records), _ptr(begin1), _ptr(end1), _ptr(begin2), _ptr(end2), n,
                _ptr(kind), _ptr(offsets), _ptr(work), _ptr(err), self._stream()), "atr_merge_plan_batch")
            total = int(offsets[n].item())
            out = self.empty((max(total, 1),), torch.uint8)
            if total and int(err.item()) == INT64_MAX:
                _check(self.lib, self.lib.atr_merge_emit_batch(
                    _ptr(align), _ptr(kind), _ptr(insert_matched), _ptr(batch1.data), _ptr(batch1.records),
                    _ptr(batch2.data), _ptr(batch2.records), _ptr(begin1), _ptr(end1), _ptr(begin2), _ptr(end2), n,
                    correct_action, min_qual_difference, comp, _ptr(offsets), _ptr(corrected), _ptr(err), _ptr(out),
                    self._stream()), "atr_merge_emit_batch")
        return kind, out[:total], corrected, int(err.item())

    def fastq_emit(self, data, records, begin, end, ubegin, uend, dest, which):
        """Formatted FASTQ text (uint8 device tensor) of the records with dest == which."""
        n = records.shape[0]
        offsets = self.empty((n + 1,), torch.int64)
        work = self.empty((max(self.lib.atr_fastq_emit_work_bytes(n), 16),), torch.uint8)
        with torch.cuda.device(self.device):
            hint = int(data.numel() // max(n, 1))
            args = (_ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin), _ptr(uend), _ptr(dest), which, n,
                    hint, _ptr(offsets), _ptr(work))
            _check(self.lib, self.lib.atr_fastq_emit(*args, None, self._stream()), "atr_fastq_emit")
            total = int(offsets[n].item())
            out = self.empty((max(total, 1),), torch.uint8)
            if total:
                _check(self.lib, self.lib.atr_fastq_emit(*args, _ptr(out), self._stream()), "atr_fastq_emit")
        return out[:total]


_backend = None
_thread_local = threading.local()


def get_backend():
    """The backend of the calling thread (``thread_backend``: one per GPU in a multi-device
    process), else the process-wide one, created on first use.  Raises (never falls back)
    when the HIP library or the GPU is missing."""
    global _backend
    mine = getattr(_thread_local, "backend", None)
    if mine is not None:
        return mine
    if _backend is None:
        _backend = HipBackend()
    return _backend


@contextlib.contextmanager
def thread_backend(backend):
    """Make ``backend`` the backend of the calling thread: objects built inside the block
    (aligners, adapters) live on its GPU.  Used by the single-process multi-device drivers
    (atropos_amd.shard), one host thread per device."""
    prev = getattr(_thread_local, "backend", None)
    _thread_local.backend = backend
    try:
        yield backend
    finally:
        _thread_local.backend = prev


def set_backend(backend, _test_double=False):
    """Install the process-wide backend: a ``HipBackend`` (bench.py pins one device per rank with it)
    or None (forget the current one).  Returns the previous one.  Anything else is refused -- the
    product never computes on a stand-in; the CPU test-suite injects its kernel emulation by saying so
    explicitly (``_test_double=True``, tests/ only)."""
    global _backend
    if backend is not None and not isinstance(backend, HipBackend) and not _test_double:
        raise TypeError("set_backend: only a HipBackend can be installed")
    prev, _backend = _backend, backend
    return prev
