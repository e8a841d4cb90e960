"""TEST INFRASTRUCTURE: a backend object with the interface of
``atropos_amd._lib.HipBackend`` that runs the lock-step CPU emulation of the gfx950
kernels (tests/emu/emu_locate.cpp, compiled from the product's own per-lane source
with -DATR_HOST_EMU).  Installed by the CPU test-suite through
``atropos_amd._lib.set_backend``; never importable from the product package."""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, "libemu_locate.so")
_CPPS = [os.path.join(_HERE, f) for f in ("emu_locate.cpp", "emu_insert.cpp", "emu_misc.cpp", "emu_fastq.cpp")]
_SRCS = _CPPS + [
    os.path.join(_ROOT, "atropos_amd", "csrc", f)
    for f in ("locate_core.hpp", "aligner_host.hpp", "insert_core.hpp", "insert_host.hpp", "misc_core.hpp",
              "filter_core.hpp", "piece_core.hpp", "fastq_core.hpp", "pairs_core.hpp", "pairs_fast_core.hpp", "linked_core.hpp", "linked_host.hpp")] + [
    os.path.join(_ROOT, "include", "atropos_hip.h")]


def build():
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DATR_HOST_EMU",
                               "-I" + os.path.join(_ROOT, "include"),
                               "-I" + os.path.join(_ROOT, "atropos_amd", "csrc"),
                               ] + _CPPS + ["-o", _SO])
    return _SO


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check(rc, what):
    if rc >= 0:
        return rc
    if rc == -1:
        raise ValueError("%s: invalid argument" % what)
    if rc == -4:
        raise MemoryError(what)
    from atropos_amd._lib import AtroposHipError, AtroposUnsupported, ERRORS
    raise (AtroposUnsupported if rc == -2 else AtroposHipError)("%s: %s" % (what, ERRORS.get(rc, "error %d" % rc)))


class EmuBackend(object):
    name = "emu"

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.emu_aligner_create.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(C.c_void_p)]
        L.emu_aligner_destroy.argtypes = [C.c_void_p]
        L.emu_aligner_destroy.restype = None
        L.emu_aligner_set_min_overlap.argtypes = [C.c_void_p, C.c_int]
        L.emu_aligner_set_indel_cost.argtypes = [C.c_void_p, C.c_int]
        L.emu_aligner_query_table.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_packed_bytes.argtypes = [C.c_int64, C.c_int]
        L.emu_packed_bytes.restype = C.c_size_t
        L.emu_pack_reads.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p,
                                     C.c_void_p, C.c_void_p]
        L.emu_multi_locate_work_bytes.argtypes = [C.c_int64, C.c_int]
        L.emu_multi_locate_work_bytes.restype = C.c_size_t
        L.emu_multi_locate_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int]
        L.emu_compare_batch.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.emu_adapter_postfilter.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                             C.c_double, C.c_int]
        L.emu_correct_errors_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_char_p, C.c_void_p, C.c_void_p]
        L.emu_insert_aligner_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.emu_insert_aligner_destroy.argtypes = [C.c_void_p]
        L.emu_insert_aligner_destroy.restype = None
        L.emu_insert_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_int, C.c_int, C.c_void_p]
        L.emu_locate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]
        self.device = torch.device("cpu")
        from atropos_amd import _lib
        self._kinds = {}
        # fixed tables come from the same aligner_host.hpp code via throw-away aligners
        for kind, args in ((_lib.TABLE_DNA15, (b"A", 0, 0)), (_lib.TABLE_ACGT, (b"A", 1, 0)),
                           (_lib.TABLE_IUPAC, (b"A", 0, 1))):
            h = self.aligner_create(args[0], 0.1, 15, args[1], args[2], 1, 1)
            k, tab = self.aligner_query_table(h)
            assert k == kind
            self._kinds[kind] = tab
            self.aligner_destroy(h)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def translate_table(self, kind):
        return self._kinds[kind]

    def packed_bytes(self, nreads, max_len):
        return self.lib.emu_packed_bytes(nreads, max_len)

    def pack_reads(self, ascii_2d, lens, max_len, table, count_invalid=False, starts=None, planes=False):
        nreads = ascii_2d.shape[0]
        packed = torch.zeros((max(self.packed_bytes(nreads, max_len), 16),), dtype=torch.uint8)
        invalid = torch.zeros((1,), dtype=torch.int32) if count_invalid else None
        if nreads and max_len:
            fn = self.lib.emu_pack_planes if planes else self.lib.emu_pack_reads
            fn.argtypes = self.lib.emu_pack_reads.argtypes
            _check(fn(_ptr(ascii_2d), ascii_2d.stride(0), _ptr(lens), _ptr(starts), nreads,
                      max_len, table, _ptr(packed), _ptr(invalid)), "emu_pack_reads")
        return (packed, int(invalid.item())) if count_invalid else packed

    def locate_debug(self, h, packed, m, n):
        matrix = torch.zeros((m + 1, n + 1), dtype=torch.int32)
        out = torch.zeros((1, 8), dtype=torch.int16)
        self.lib.emu_locate_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.lib.emu_locate_debug(h, _ptr(packed), n, _ptr(matrix), _ptr(out)), "emu_locate_debug")
        return matrix, out

    def compare_packed(self, h, packed, lens, nreads, max_len, suffix):
        out = torch.zeros((nreads, 8), dtype=torch.int16)
        if nreads:
            self.lib.emu_compare_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
            _check(self.lib.emu_compare_packed(h, _ptr(packed), _ptr(lens), nreads, max_len, int(suffix), _ptr(out)),
                   "emu_compare_packed")
        return out

    def planes_count_uncoded(self, planes, lens, other_lens, nreads, max_len):
        count = torch.zeros((1,), dtype=torch.int32)
        if nreads and max_len:
            self.lib.emu_planes_count_uncoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
            _check(self.lib.emu_planes_count_uncoded(_ptr(planes), _ptr(lens), _ptr(other_lens), nreads, max_len, _ptr(count)),
                   "emu_planes_count_uncoded")
        return int(count.item())

    def multi_locate_batch(self, refs, ref_lens, queries, query_lens, e, flags, min_overlap, max_matches,
                           max_ref_len, out_stride):
        npairs = refs.shape[0]
        out = torch.zeros((npairs, out_stride, 8), dtype=torch.int16)
        counts = torch.zeros((npairs,), dtype=torch.int32)
        work = torch.zeros((max(self.lib.emu_multi_locate_work_bytes(npairs, max_ref_len), 4),), dtype=torch.uint8)
        if npairs:
            _check(self.lib.emu_multi_locate_batch(_ptr(refs), refs.stride(0), _ptr(ref_lens), _ptr(queries),
                                                   queries.stride(0), _ptr(query_lens), npairs, e, flags, min_overlap,
                                                   max_matches, max_ref_len, _ptr(work), _ptr(out), _ptr(counts),
                                                   out_stride), "atr_multi_locate_batch")
        return out, counts

    def compare_batch(self, ref, queries, lens, max_len, wildcard_ref, wildcard_query, suffix):
        n = queries.shape[0]
        out = torch.zeros((n, 8), dtype=torch.int16)
        if n:
            _check(self.lib.emu_compare_batch(ref, len(ref), _ptr(queries), queries.stride(0), _ptr(lens), n, max_len,
                                              int(wildcard_ref), int(wildcard_query), int(suffix), _ptr(out)),
                   "atr_compare_batch")
        return out

    def adapter_postfilter(self, records, m, min_overlap, max_error_rate, rmp, max_rmp, accept_full):
        if records.shape[0]:
            _check(self.lib.emu_adapter_postfilter(_ptr(records), records.shape[0], m, min_overlap, max_error_rate,
                                                   _ptr(rmp), 0 if rmp is None else rmp.shape[1],
                                                   0.0 if max_rmp is None else max_rmp, int(accept_full)),
                   "atr_adapter_postfilter")
        return records

    def correct_errors_batch(self, seq1, qual1, lens1, seq2, qual2, lens2, insert, mask, action, min_qual_diff,
                             truncate, comp):
        n = seq1.shape[0]
        changed = torch.zeros((n, 2), dtype=torch.int32)
        newlen = torch.zeros((n, 2), dtype=torch.int32)
        if n:
            assert seq1.stride(0) == seq2.stride(0)
            _check(self.lib.emu_correct_errors_batch(_ptr(seq1), _ptr(qual1), _ptr(lens1), _ptr(seq2), _ptr(qual2),
                                                     _ptr(lens2), seq1.stride(0), _ptr(insert), _ptr(mask), n,
                                                     seq1.shape[1], action, min_qual_diff, int(truncate), comp,
                                                     _ptr(changed), _ptr(newlen)), "atr_correct_errors_batch")
        return changed, newlen

    def insert_correct_batch(self, records, seq1, qual1, lens1, seq2, qual2, lens2, action, min_qual_diff, comp,
                             changed=None, newlen=None, planes1=None, planes2=None):
        n = seq1.shape[0]
        changed = torch.zeros((n, 2), dtype=torch.int32) if changed is None else changed
        newlen = torch.zeros((n, 2), dtype=torch.int32) if newlen is None else newlen
        fn = self.lib.emu_insert_correct_batch
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p,
                       C.c_void_p]
        if n:
            _check(fn(_ptr(records), None if planes1 is None else _ptr(planes1.packed),
                      None if planes2 is None else _ptr(planes2.packed), 0 if planes1 is None else planes1.max_len,
                      _ptr(seq1), _ptr(qual1), _ptr(lens1), _ptr(seq2), _ptr(qual2), _ptr(lens2), seq1.stride(0), n,
                      seq1.shape[1], action, min_qual_diff, comp, _ptr(changed), _ptr(newlen)), "atr_insert_correct_batch")
        return changed, newlen

    def insert_aligner_create(self, cfg):
        h = C.c_void_p()
        _check(self.lib.emu_insert_aligner_create(C.addressof(cfg), C.byref(h)), "atr_insert_aligner_create")
        return h

    def insert_aligner_destroy(self, h):
        self.lib.emu_insert_aligner_destroy(h)

    def insert_match_batch(self, h, packed1, lens1, packed2, lens2, npairs, max_len, cased=False):
        out = torch.zeros((npairs, 3, 8), dtype=torch.int16)
        if npairs:
            _check(self.lib.emu_insert_match_batch(h, _ptr(packed1), _ptr(lens1), _ptr(packed2), _ptr(lens2), npairs,
                                                   max_len, int(cased), _ptr(out)), "atr_insert_match_batch")
        return out

    def insert_match_correct_batch(self, h, planes1, planes2, seq1, qual1, seq2, qual2, action, min_qual_diff, comp,
                                   changed=None, newlen=None):
        # the test double has no fused kernel: the two steps of the contract one after the other
        out = self.insert_match_batch(h, planes1.packed, planes1.lens, planes2.packed, planes2.lens, planes1.nreads, planes1.max_len)
        changed, newlen = self.insert_correct_batch(out, seq1, qual1, planes1.lens, seq2, qual2, planes2.lens, action, min_qual_diff,
                                                    comp, changed, newlen, planes1=planes1, planes2=planes2)
        return out, changed, newlen

    def case_sensitive_table(self):
        buf = C.create_string_buffer(256)
        _check(self.lib.emu_case_sensitive_table(buf), "emu_case_sensitive_table")
        return buf.raw

    def aligner_create(self, ref, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost):
        h = C.c_void_p()
        _check(self.lib.emu_aligner_create(ref, len(ref), e, flags, int(wildcard_ref), int(wildcard_query),
                                           min_overlap, indel_cost, C.byref(h)), "atr_aligner_create")
        return h

    def aligner_destroy(self, h):
        self.lib.emu_aligner_destroy(h)

    def aligner_set_min_overlap(self, h, v):
        _check(self.lib.emu_aligner_set_min_overlap(h, v), "atr_aligner_set_min_overlap")

    def aligner_set_indel_cost(self, h, v):
        _check(self.lib.emu_aligner_set_indel_cost(h, v), "atr_aligner_set_indel_cost")

    def aligner_query_table(self, h):
        buf = C.create_string_buffer(256)
        kind = _check(self.lib.emu_aligner_query_table(h, buf), "atr_aligner_query_table")
        return kind, buf.raw

    def locate_batch(self, h, packed, lens, nreads, max_len, filtered=True, path=None):
        from atropos_amd._lib import LOCATE_PATHS
        if path is None:
            path = "auto" if filtered else "full"
        out = torch.zeros((nreads, 8), dtype=torch.int16)
        if nreads:
            _check(self.lib.emu_locate_batch(h, _ptr(packed), _ptr(lens), nreads, max_len, _ptr(out), LOCATE_PATHS[path]),
                   "atr_locate_batch")
        return out

    def locate_planes_applies(self, h, max_len, ragged=False):
        self.lib.emu_locate_planes_all_widths.argtypes = [C.c_void_p, C.c_int, C.c_int]
        return bool(self.lib.emu_locate_planes_all_widths(h, int(max_len), int(bool(ragged))))

    def locate_planes_batch(self, h, planes, lens, nreads, max_len):
        out = torch.zeros((nreads, 8), dtype=torch.int16)
        if nreads:
            self.lib.emu_locate_planes_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
            _check(self.lib.emu_locate_planes_batch(h, _ptr(planes), _ptr(lens), nreads, max_len, _ptr(out)), "atr_locate_planes_batch")
        return out

    def linked_create(self, specs):
        from atropos_amd._lib import LinkedAdapterSpec
        arr = (LinkedAdapterSpec * len(specs))(*specs)
        h = C.c_void_p()
        self.lib.emu_linked_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib.emu_linked_create(C.addressof(arr), len(specs), C.byref(h)), "atr_linked_create")
        return h

    def linked_destroy(self, h):
        self.lib.emu_linked_destroy.argtypes = [C.c_void_p]
        self.lib.emu_linked_destroy.restype = None
        self.lib.emu_linked_destroy(h)

    def linked_match_batch(self, h, packed, lens, nreads, max_len):
        which = torch.zeros((nreads, 2), dtype=torch.int8)
        front = torch.zeros((nreads, 8), dtype=torch.int16)
        back = torch.zeros((nreads, 8), dtype=torch.int16)
        if nreads:
            self.lib.emu_linked_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                                        C.c_void_p, C.c_void_p]
            _check(self.lib.emu_linked_match_batch(h, _ptr(packed), _ptr(lens), nreads, max_len, _ptr(which), _ptr(front),
                                                   _ptr(back)), "atr_linked_match_batch")
        return which, front, back

    def locate_pairs_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                           query_max_len, npairs, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost,
                           need=None, path="auto"):
        from atropos_amd._lib import PAIRS_PATHS
        out = torch.zeros((npairs, 8), dtype=torch.int16)
        C.c_int.in_dll(self.lib, "emu_pairs_path").value = PAIRS_PATHS[path]
        _check(self.lib.emu_locate_pairs_need_batch(
            _ptr(ref_packed), _ptr(ref_lens), ref_max_len, int(revcomp_ref), _ptr(query_packed), _ptr(query_lens),
            query_max_len, C.c_int64(npairs), C.c_double(e), flags, int(wildcard_ref), int(wildcard_query), min_overlap,
            indel_cost, _ptr(need), _ptr(out)), "emu_locate_pairs_need_batch")
        return out

    def locate_pairs_long_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                                query_max_len, npairs, e, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost):
        out = torch.zeros((npairs, 8), dtype=torch.int16)
        _check(self.lib.emu_locate_pairs_long_batch(
            _ptr(ref_packed), _ptr(ref_lens), ref_max_len, int(revcomp_ref), _ptr(query_packed), _ptr(query_lens),
            query_max_len, C.c_int64(npairs), C.c_double(e), flags, int(wildcard_ref), int(wildcard_query), min_overlap,
            indel_cost, _ptr(out)), "atr_locate_pairs_long_batch")
        return out

    def locate_pairs_full_batch(self, ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                                query_max_len, npairs, e, flags, min_overlap, indel_cost):
        return self.locate_pairs_batch(ref_packed, ref_lens, ref_max_len, revcomp_ref, query_packed, query_lens,
                                       query_max_len, npairs, e, flags, False, False, min_overlap, indel_cost, path="full")

    # -- device-resident FASTQ batch (CPU twin) ----------------------------------
    def fastq_index(self, data, nbytes):
        L = self.lib
        info = torch.zeros((2,), dtype=torch.int64)
        _check(L.emu_fastq_count_lines(_ptr(data), C.c_int64(nbytes), _ptr(info)), "emu_fastq_count_lines")
        nlines = int(info[0])
        line_ends = torch.zeros((max(nlines, 1),), dtype=torch.int32)
        records = torch.zeros((nlines // 4, 8), dtype=torch.int32)
        _check(L.emu_fastq_index(_ptr(data), C.c_int64(nbytes), _ptr(line_ends), C.c_int64(nlines), _ptr(records),
                                 C.c_void_p(info.data_ptr() + 8)), "emu_fastq_index")
        return records, line_ends, nlines, int(info[1])

    def pack_records(self, data, records, begin, end, max_len, table, count_invalid=False, planes=False):
        n = records.shape[0]
        packed = torch.zeros((max(self.packed_bytes(n, max_len), 16),), dtype=torch.uint8)
        lens = torch.zeros((n,), dtype=torch.int32)
        invalid = torch.zeros((1,), dtype=torch.int32)
        _check(self.lib.emu_pack_records(_ptr(data), _ptr(records), _ptr(begin), _ptr(end), C.c_int64(n), max_len, table,
                                         int(planes), _ptr(packed), _ptr(lens), _ptr(invalid)), "emu_pack_records")
        return (packed, lens, int(invalid.item())) if count_invalid else (packed, lens)

    def clip_batch(self, records, begin, end, front, back):
        _check(self.lib.emu_clip_batch(_ptr(records), _ptr(begin), _ptr(end), C.c_int64(begin.shape[0]), front, back),
               "emu_clip_batch")

    def quality_trim_batch(self, data, records, begin, end, cutoff_front, cutoff_back, base, nextseq):
        _check(self.lib.emu_quality_trim_batch(_ptr(data), _ptr(records), _ptr(begin), _ptr(end),
                                               C.c_int64(begin.shape[0]), cutoff_front, cutoff_back, base, int(nextseq)),
               "emu_quality_trim_batch")

    def nend_trim_batch(self, data, records, begin, end, ubegin=None, uend=None):
        _check(self.lib.emu_nend_trim_batch(_ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin), _ptr(uend),
                                            C.c_int64(begin.shape[0])), "emu_nend_trim_batch")

    def match_trim_batch(self, matches, front, default_front, begin, end, active, matched):
        _check(self.lib.emu_match_trim_batch(_ptr(matches), _ptr(front), default_front, _ptr(begin), _ptr(end),
                                             _ptr(active), _ptr(matched), C.c_int64(begin.shape[0])),
               "emu_match_trim_batch")

    def read_filter_batch(self, data, records, begin, end, ubegin, uend, matched, min_len, max_len, max_n,
                          discard_trimmed, discard_untrimmed, masks=False):
        out = torch.zeros((begin.shape[0],), dtype=torch.uint8)
        _check(self.lib.emu_read_filter_batch(_ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin), _ptr(uend),
                                              _ptr(matched), C.c_int64(begin.shape[0]), min_len, max_len,
                                              C.c_double(max_n), int(discard_trimmed), int(discard_untrimmed),
                                              None if masks else _ptr(out), _ptr(out) if masks else None),
               "emu_read_filter_batch")
        return out

    def pair_filter_batch(self, mask1, mask2, min_affected):
        dest = torch.zeros((mask1.shape[0],), dtype=torch.uint8)
        _check(self.lib.emu_pair_filter_batch(_ptr(mask1), _ptr(mask2), C.c_int64(mask1.shape[0]), min_affected,
                                              _ptr(dest)), "emu_pair_filter_batch")
        return dest

    def insert_plan_batch(self, insert, fb1, fb2, batch1, batch2, begin1, end1, begin2, end2, uend1, uend2,
                          min_insert_len, symmetric, trim_action, correct_action=-1, min_qual_difference=1, comp=None):
        n = begin1.shape[0]
        m1, m2 = torch.zeros((n,), dtype=torch.uint8), torch.zeros((n,), dtype=torch.uint8)
        corrected = torch.zeros((n, 2), dtype=torch.int32)
        err = torch.zeros((1,), dtype=torch.int64)
        _check(self.lib.emu_insert_plan_batch(
            _ptr(insert), _ptr(fb1), _ptr(fb2), _ptr(batch1.data), _ptr(batch1.records), _ptr(batch2.data),
            _ptr(batch2.records), _ptr(begin1), _ptr(end1), _ptr(begin2), _ptr(end2), _ptr(uend1), _ptr(uend2),
            C.c_int64(n), min_insert_len, int(symmetric), trim_action, correct_action, min_qual_difference, comp,
            _ptr(m1), _ptr(m2), _ptr(corrected), _ptr(err)), "emu_insert_plan_batch")
        return m1, m2, corrected, int(err.item())

    def merge_batch(self, align, need, insert_matched, batch1, batch2, begin1, end1, begin2, end2, correct_action=-1,
                    min_qual_difference=1, comp=None):
        n = begin1.shape[0]
        kind = torch.zeros((n,), dtype=torch.uint8)
        offsets = torch.zeros((n + 1,), dtype=torch.int64)
        corrected = torch.zeros((n, 2), dtype=torch.int32)
        err = torch.zeros((1,), dtype=torch.int64)
        args = (_ptr(align), _ptr(need), _ptr(insert_matched), _ptr(batch1.data), _ptr(batch1.records), _ptr(batch2.data),
                _ptr(batch2.records), _ptr(begin1), _ptr(end1), _ptr(begin2), _ptr(end2), C.c_int64(n), correct_action,
                min_qual_difference, comp, _ptr(kind), _ptr(offsets), _ptr(corrected), _ptr(err))
        _check(self.lib.emu_merge_batch(*args, None), "emu_merge_batch")
        total = int(offsets[n])
        out = torch.zeros((max(total, 1),), dtype=torch.uint8)
        if total and int(err.item()) == (1 << 63) - 1:
            _check(self.lib.emu_merge_batch(*args, _ptr(out)), "emu_merge_batch")
        return kind, out[:total], corrected, int(err.item())

    def fastq_emit(self, data, records, begin, end, ubegin, uend, dest, which):
        n = records.shape[0]
        offsets = torch.zeros((n + 1,), dtype=torch.int64)
        args = (_ptr(data), _ptr(records), _ptr(begin), _ptr(end), _ptr(ubegin), _ptr(uend), _ptr(dest), which,
                C.c_int64(n), _ptr(offsets))
        _check(self.lib.emu_fastq_emit(*args, None), "emu_fastq_emit")
        total = int(offsets[n])
        out = torch.zeros((max(total, 1),), dtype=torch.uint8)
        if total:
            _check(self.lib.emu_fastq_emit(*args, _ptr(out)), "emu_fastq_emit")
        return out[:total]
