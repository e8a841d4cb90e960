# coding: utf-8
"""The callers of the alignment path (reference atropos/commands/trim/modifiers.py:
AdapterCutter :91-195, ErrorCorrectorMixin :201-357, InsertAdapterCutter :359-509).

Same classes and per-read ``__call__`` behaviour as the reference; in addition every
cutter has a batched twin (``call_batch``) that runs each alignment stage once over the
whole batch on the GPU and then does the per-read bookkeeping (trimming, statistics,
``match``/``match_info``) from the result arrays -- SURVEY section 8(f1).
``MergeOverlapping`` (:864-931, section 8(f3)) is here too: its per-pair aligner is a batched
GPU call.  Quality / NextSeq / N-end trimming, clipping and the read filters live in the
device-resident FASTQ pipeline (``atropos_amd.trim``); the remaining modifiers of the reference
(bisulfite trimmers, name editors, ...) are cheap string operations off the alignment path
and out of scope.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import align
from .align import InsertAligner, Match
from .util import BASE_COMPLEMENTS, reverse_complement

_ACTIONS = {'N': 0, 'conservative': 1, 'liberal': 2}


def _comp_table():
    t = bytearray(256)
    for base, comp in BASE_COMPLEMENTS.items():
        t[ord(base)] = ord(comp)
    return bytes(t)


COMP_TABLE = _comp_table()


class Modifier(object):
    """A read modifier: ``modifier(read) -> read``; ``summarize()`` feeds the run summary."""

    name = property(lambda self: type(self).__name__)

    def summarize(self):
        return dict()


class ReadPairModifier(Modifier):
    """A modifier of read PAIRS: ``modifier(read1, read2) -> (read1, read2)``."""

    def __call__(self, read1, read2):
        raise NotImplementedError("%s does not implement __call__" % self.name)


class AdapterCutter(Modifier):
    """Find the best-matching of several adapters in each read and remove it, up to ``times`` times
    (reference commands/trim/modifiers.py:91-195).  ``action``: 'trim', 'mask' (the removed bases
    become N) or None (only record the match).  The work happens in ``call_batch``: per round ONE GPU
    call per adapter over all reads that are still being trimmed; ``cutter(read)`` is a batch of one."""

    def __init__(self, adapters=None, times=1, action='trim'):
        self.adapters = adapters or []
        self.times, self.action = times, action
        self.with_adapters = 0

    def __call__(self, read):
        return self.call_batch([read])[0]

    def call_batch(self, reads):
        out = list(reads)
        found = {i: [] for i, read in enumerate(reads) if len(read) > 0}      # read -> its successive matches
        current = {i: reads[i] for i in found}
        active = sorted(found)
        for _ in range(self.times):
            if not active:
                break
            batch = [current[i] for i in active]
            best = [None] * len(batch)
            for adapter in self.adapters:
                for k, match in enumerate(adapter.match_to_batch(batch)):
                    # most matches wins, the earlier adapter on ties (:107-122)
                    if match is not None and (best[k] is None or match.matches > best[k].matches):
                        best[k] = match
            again = []
            for k, i in enumerate(active):
                if best[k] is not None:
                    found[i].append(best[k])
                    current[i] = best[k].adapter.trimmed(best[k])
                    again.append(i)
            active = again                     # a read without a match leaves the rounds (:133-139)
        for i, matches in found.items():
            out[i] = self._record(reads[i], current[i], matches)
        return out

    def _record(self, original, trimmed, matches):
        """Apply ``action`` and attach match / match_info, given the successive matches of one read."""
        if not matches:
            trimmed.match = trimmed.match_info = None
            return trimmed
        if len(trimmed) >= len(original):
            raise AssertionError("Trimmed read isn't shorter than original")
        result = trimmed
        if self.action == 'mask':
            # N for every removed base, on the side it was removed from.  The removed length is asked
            # from the adapter again, which -- as in the reference (:161) -- counts the match a second
            # time in the adapter's statistics.
            head = tail = 0
            for match in sorted(matches, key=lambda m: -m.astart):
                gone = len(match.read.sequence) - len(match.adapter.trimmed(match).sequence)
                head, tail = (head + gone, tail) if match.front else (head, tail + gone)
            result.sequence = 'N' * head + trimmed.sequence + 'N' * tail
            result.qualities = matches[0].read.qualities
            if len(result.sequence) != len(original):
                raise AssertionError("masked read has a different length")
        elif self.action is None:
            result = original
        result.match = matches[-1]
        result.match_info = [match.get_info_record() for match in matches]
        self.with_adapters += 1
        return result

    def summarize(self):
        per_adapter = OrderedDict((adapter.name, adapter.summarize()) for adapter in self.adapters)
        return {"records_with_adapters": self.with_adapters, "adapters": per_adapter}


class ErrorCorrectorMixin(object):
    """Error correction of the overlapping part of a read pair.

    Args:
        mismatch_action: what to do on a mismatch between the overlapping portions of
            read1 and read2: 'liberal', 'conservative' or 'N'.
        min_qual_difference: minimum base-quality difference required to trust one read
            over the other.
    """

    def __init__(self, mismatch_action=None, min_qual_difference=1):
        self.mismatch_action, self.corrected_pairs, self.corrected_bp = mismatch_action, 0, [0, 0]
        # quality lead read 1 needs over read 2 (and read 2 over read 1, as a negative number) to win a mismatch
        self.r1r2_min_qual_difference, self.r2r1_min_qual_difference = min_qual_difference, -min_qual_difference

    def correct_errors(self, read1, read2, insert_match, truncate_seqs=False):
        """Correct errors in one pair of overlapping reads (a batch of one on the device)."""
        self.correct_errors_batch([read1], [read2], [insert_match], truncate_seqs)

    def correct_errors_batch(self, reads1, reads2, insert_matches, truncate_seqs=False):
        """Batched ``correct_errors``: ``insert_matches[i]`` is the insert-match tuple of
        pair i, or None to leave the pair alone.  Reads are updated in place (sequence,
        qualities, ``corrected``), the counters ``corrected_pairs``/``corrected_bp`` too."""
        idx = [i for i, im in enumerate(insert_matches)
               if im is not None and not (reads1[i].corrected > 0 or reads2[i].corrected > 0)]
        if not idx:
            return
        r1s, r2s = [reads1[i] for i in idx], [reads2[i] for i in idx]
        has_quals = [bool(a.qualities and b.qualities) for a, b in zip(r1s, r2s)]
        if self.mismatch_action in ('liberal', 'conservative') and not all(has_quals):
            raise ValueError("Cannot perform quality-based error correction on reads lacking quality information")
        for flag in (True, False):          # pairs with / without qualities go in separate launches
            sel = [k for k, h in enumerate(has_quals) if h == flag]
            if sel:
                self._correct_on_device([r1s[k] for k in sel], [r2s[k] for k in sel],
                                        [insert_matches[idx[k]] for k in sel], flag, truncate_seqs)

    def _correct_on_device(self, r1s, r2s, ims, with_quals, truncate_seqs):
        be = _lib.get_backend()
        n = len(r1s)
        width = max(max(len(r) for r in r1s), max(len(r) for r in r2s), 1)

        def mat(strings):
            m = np.zeros((n, width), dtype=np.uint8)
            for k, s in enumerate(strings):
                b = s.encode('latin-1')
                m[k, :len(b)] = np.frombuffer(b, dtype=np.uint8)
            return torch.from_numpy(m).to(be.device)

        s1, s2 = mat([r.sequence for r in r1s]), mat([r.sequence for r in r2s])
        q1 = mat([r.qualities for r in r1s]) if with_quals else None
        q2 = mat([r.qualities for r in r2s]) if with_quals else None
        l1 = torch.tensor([len(r) for r in r1s], dtype=torch.int32, device=be.device)
        l2 = torch.tensor([len(r) for r in r2s], dtype=torch.int32, device=be.device)
        im = torch.tensor([[int(v) for v in t[:4]] for t in ims], dtype=torch.int16, device=be.device)
        changed, newlen = be.correct_errors_batch(
            s1, q1, l1, s2, q2, l2, im, None, _ACTIONS[self.mismatch_action], self.r1r2_min_qual_difference,
            truncate_seqs, COMP_TABLE)
        changed, newlen = changed.cpu().numpy(), newlen.cpu().numpy()
        s1, s2 = s1.cpu().numpy(), s2.cpu().numpy()
        if with_quals:
            q1, q2 = q1.cpu().numpy(), q2.cpu().numpy()
        for k in range(n):
            c1, c2 = int(changed[k, 0]), int(changed[k, 1])
            if c1 < 0:          # the exception the reference raises for this pair
                exc = {-1: KeyError, -2: IndexError, -3: ValueError}[c1]
                raise exc("error correction of pair %r: %s" % (r1s[k].name, {
                    -1: "base without a complement", -2: "overlap outside a read",
                    -3: "Cannot determine the mode of an empty sequence"}[c1]))
            if not (c1 or c2):
                continue
            self.corrected_pairs += 1
            for read, seqs, quals, cnt, num, ln in ((r1s[k], s1, q1, c1, 0, int(newlen[k, 0])),
                                                    (r2s[k], s2, q2, c2, 1, int(newlen[k, 1]))):
                if not cnt:
                    continue
                self.corrected_bp[num] += cnt
                read.corrected = cnt
                read.sequence = seqs[k, :ln].tobytes().decode('latin-1')
                if with_quals:
                    read.qualities = quals[k, :ln].tobytes().decode('latin-1')

    def summarize(self):
        return {"records_corrected": self.corrected_pairs, "bp_corrected": self.corrected_bp}


def _mirrored(match, read_len):
    """A lone adapter match of one read carried over to its mate of length ``read_len`` (the
    adapter is assumed to start at the same position in both reads): None when the mate is too
    short to reach it; ends at the mate's end when the mate is the shorter read ('matches' and
    'errors' are then those of the original, as in the reference :421-431)."""
    if match.rstart > read_len:
        return None
    twin = match.copy()
    overhang = twin.rstop - read_len
    if overhang < 0:
        twin.astop += overhang
        twin.rstop = read_len
    return twin


class InsertAdapterCutter(ReadPairModifier, ErrorCorrectorMixin):
    """AdapterCutter that uses InsertAligner to first try to identify the insert overlap
    before falling back to semi-global adapter alignment (reference modifiers.py:359-509).

    Args:
        adapter1, adapter2: Adapters.
        action: 'trim', 'mask' or None.
        mismatch_action: see ErrorCorrectorMixin.
        symmetric: assume the adapter appears at the same place in both reads.
        min_insert_overlap: minimum overlap of the reads for an insert match.
        aligner_args: further arguments of InsertAligner.

    ``call_batch`` does the work in stages over all pairs (insert matching, the two adapters'
    own alignments for the pairs without an insert match, error correction, trimming);
    ``cutter(read1, read2)`` is a batch of one.
    """

    def __init__(self, adapter1, adapter2, action='trim', mismatch_action=None, symmetric=True,
                 min_insert_overlap=1, **aligner_args):
        ErrorCorrectorMixin.__init__(self, mismatch_action)
        self.adapter1, self.adapter2 = adapter1, adapter2
        self.min_insert_len = aligner_args["min_insert_overlap"] = min_insert_overlap
        self.aligner = InsertAligner(*(a.sequence for a in (adapter1, adapter2)), **aligner_args)
        self.action, self.symmetric = action, symmetric
        self.with_adapters = [0, 0]

    def _plan(self, lengths, insert_result, fallback):
        """What to do with one pair: ``(overlap to error-correct or None, match1, match2)``, from the
        insert aligner's result or -- without one -- the two adapters' own matches (:397-446)."""
        fix = self.mismatch_action
        overlap, correct = None, False
        if insert_result:
            overlap, match1, match2 = insert_result
            correct = fix is not None and overlap[5] > 0
        else:
            match1, match2 = fallback
            # both adapters found at the same position: the reads overlap completely before it
            if fix and match1 and match2 and match1.rstart == match2.rstart:
                overlap, correct = self._overlap_before(match1, lengths), True
        if self.symmetric and (match1 is None) != (match2 is None):
            if match2 is None:
                match2 = _mirrored(match1, lengths[1])
            else:
                match1 = _mirrored(match2, lengths[0])
            if fix and not overlap and match1 and match2:
                overlap, correct = self._overlap_before(match1, lengths), True
        return (overlap if correct else None), match1, match2

    @staticmethod
    def _overlap_before(match1, lengths):
        """The insert-match coordinates implied by an adapter starting at match1.rstart in both reads."""
        return (lengths[1] - match1.rstart, lengths[1], 0, match1.rstart)

    def __call__(self, read1, read2):
        return self.call_batch([read1], [read2])[0]

    def call_batch(self, reads1, reads2):
        """Returns the list of (read1, read2) results."""
        out = [(a, b) for a, b in zip(reads1, reads2)]
        act = [i for i, (a, b) in enumerate(out) if min(len(a), len(b)) >= self.min_insert_len]     # :392-394
        if not act:
            return out
        results = self.aligner.match_insert_batch([reads1[i].sequence for i in act],
                                                  [reads2[i].sequence for i in act]).results()
        miss = [k for k, r in enumerate(results) if r is None]
        fb1 = self.adapter1.match_to_batch([reads1[act[k]] for k in miss]) if miss else []
        fb2 = self.adapter2.match_to_batch([reads2[act[k]] for k in miss]) if miss else []
        fallback = dict(zip(miss, zip(fb1, fb2)))
        plans = []
        for k, i in enumerate(act):
            reads1[i].insert_overlap = reads2[i].insert_overlap = (results[k] is not None)
            plans.append(self._plan([len(reads1[i]), len(reads2[i])], results[k], fallback.get(k)))
        self.correct_errors_batch([reads1[i] for i in act], [reads2[i] for i in act], [p[0] for p in plans],
                                  truncate_seqs=True)
        for (_, match1, match2), i in zip(plans, act):
            out[i] = (self.trim(reads1[i], self.adapter1, match1, 0), self.trim(reads2[i], self.adapter2, match2, 1))
        return out

    def trim(self, read, adapter, match, read_idx):
        """Trim (or mask, or just annotate) one read of a pair according to its match."""
        if match is None:
            read.match = read.match_info = None
            return read
        match.adapter, match.read, match.front = adapter, read, False
        result = read
        if self.action is not None and match.rstart < len(read):
            result = adapter.trimmed(match)
            if self.action == 'mask':
                result.sequence = result.sequence + 'N' * (len(read) - len(result))
                result.qualities = read.qualities
        result.match, result.match_info = match, [match.get_info_record()]
        self.with_adapters[read_idx] += 1
        return result

    def summarize(self):
        parts = (self.adapter1, self.adapter2)
        summary = {"records_with_adapters": self.with_adapters, "adapters": tuple({a.name: a.summarize()} for a in parts)}
        return dict(summary, **ErrorCorrectorMixin.summarize(self)) if self.mismatch_action else summary


class MergeOverlapping(ReadPairModifier, ErrorCorrectorMixin):
    """Merge overlapping reads; the merged read is stored in read1 and read2 becomes None
    (reference: commands/trim/modifiers.py:864-931).  The per-pair alignment
    ``Aligner(reverse_complement(read2), error_rate, flags).locate(read1)`` runs on the GPU for
    the whole batch (``PairAligner``, atr_locate_pairs_batch); reverse complement included."""

    def __init__(self, min_overlap=0.9, error_rate=0.1, mismatch_action=None):
        ErrorCorrectorMixin.__init__(self, mismatch_action)
        # above 1: a number of bases; up to 1: a fraction of the shorter read
        self.min_overlap = min_overlap if min_overlap <= 1 else int(min_overlap)
        self.error_rate, self._aligners = error_rate, {}

    def _aligner(self, flags):
        if flags not in self._aligners:
            self._aligners[flags] = align.PairAligner(self.error_rate, flags, revcomp_ref=True)
        return self._aligners[flags]

    def __call__(self, read1, read2):
        return self.call_batch([read1], [read2])[0]

    def _required_overlap(self, len1, len2):
        """The overlap in bases a pair of these lengths needs (a fraction <= 1 is relative to the shorter read)."""
        if self.min_overlap > 1:
            return self.min_overlap
        return max(2, round(self.min_overlap * min(len1, len2)))

    def call_batch(self, reads1, reads2):
        """Batched twin of ``__call__``: returns the list of (read1, read2-or-None) pairs."""
        out = [(r1, r2) for r1, r2 in zip(reads1, reads2)]
        plan = {}                                            # flags -> [(index, required overlap, insert matched)]
        for i, (read1, read2) in enumerate(out):
            need = self._required_overlap(len(read1.sequence), len(read2.sequence))
            if min(len(read1.sequence), len(read2.sequence)) < need:
                continue
            insert_matched = bool(read1.insert_overlap and read2.insert_overlap)
            # an insert overlap with a 3' overhang was already found: constrain the alignment
            flags = (align.START_WITHIN_SEQ1 | align.STOP_WITHIN_SEQ2) if insert_matched else align.SEMIGLOBAL
            plan.setdefault(flags, []).append((i, need, insert_matched))
        for flags, todo in plan.items():
            idx = [t[0] for t in todo]
            for i in idx:
                reverse_complement(out[i][1].sequence)       # KeyError on a base without complement, as the reference
            # reference = reverse_complement(read2) (formed on the device), query = read1
            # (need: an alignment with fewer matches is not looked at below -- the library may then report None)
            alignments = self._aligner(flags).locate_batch([out[i][1].sequence for i in idx],
                                                           [out[i][0].sequence for i in idx],
                                                           need=[t[1] for t in todo]).tuples()
            for (i, need, insert_matched), alignment in zip(todo, alignments):
                if alignment and alignment[4] >= need:
                    out[i] = self._merge(out[i][0], out[i][1], alignment, insert_matched)
        return out

    def _merge(self, read1, read2, alignment, insert_matched):
        """Join read 1 and the reverse complement of read 2 along ``alignment`` (read 2 is the
        reference, read 1 the query); the merged read replaces read 1 (:896-931)."""
        ref_start, ref_stop, qry_start, qry_stop, _matches, errors = alignment
        mate = reverse_complement(read2.sequence)            # as it is BEFORE any correction, like the reference (:896)
        # only correct errors if that was not already done by the insert aligner
        if self.mismatch_action and errors > 0 and not insert_matched:
            self.correct_errors(read1, read2, alignment)
        both_quals = bool(read1.qualities and read2.qualities)
        mate_quals = read2.qualities[::-1] if read2.qualities is not None else None
        if ref_start == 0 and ref_stop == len(read2.sequence):
            pass                                                 # read 2 lies inside read 1
        elif qry_start == 0 and qry_stop == len(read1.sequence):
            read1.sequence = mate                                # read 1 lies inside read 2
            read1.qualities = read2.qualities[::-1]              # TypeError without qualities, like the reference
        elif qry_start > 0:                                      # read 2 continues read 1 to the right
            read1.sequence = read1.sequence + mate[ref_stop:]
            if both_quals:
                read1.qualities = read1.qualities + mate_quals[ref_stop:]
        elif ref_start > 0:                                      # read 2 starts left of read 1
            read1.sequence = mate + read1.sequence[qry_stop:]
            if both_quals:
                read1.qualities = mate_quals + read1.qualities[qry_stop:]
        else:
            raise ValueError("Invalid alignment while trying to merge read {}: {}".format(
                read1.name, ",".join(map(str, alignment))))
        read1.merged = True
        return (read1, None)
