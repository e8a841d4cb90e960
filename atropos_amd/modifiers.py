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
    """Base class of the read modifiers."""

    @property
    def name(self):
        return self.__class__.__name__

    def summarize(self):
        return {}


class ReadPairModifier(Modifier):
    def __call__(self, read1, read2):
        raise NotImplementedError()


class AdapterCutter(Modifier):
    """Repeatedly find one of multiple adapters in reads; the search is repeated ``times``
    times.  ``action``: 'trim', 'mask' (replace the adapter by N) or None."""

    def __init__(self, adapters=None, times=1, action='trim'):
        self.adapters = adapters or []
        self.times = times
        self.action = action
        self.with_adapters = 0

    def _best_match(self, read):
        """The adapter whose match has the most matches (first one wins ties)."""
        best = None
        for adapter in self.adapters:
            match = adapter.match_to(read)
            if match is None:
                continue
            if best is None or match.matches > best.matches:
                best = match
        return best

    def _finish(self, read, trimmed_read, matches):
        """Everything __call__ does once the list of consecutive matches is known."""
        if not matches:
            trimmed_read.match = None
            trimmed_read.match_info = None
            return trimmed_read
        assert len(trimmed_read) < len(read), "Trimmed read isn't shorter than original"
        if self.action == 'trim':
            pass
        elif self.action == 'mask':
            masked_sequence = trimmed_read.sequence
            for match in sorted(matches, reverse=True, key=lambda m: m.astart):
                nstr = 'N' * (len(match.read.sequence) - len(match.adapter.trimmed(match).sequence))
                if match.front:
                    masked_sequence = nstr + masked_sequence
                else:
                    masked_sequence += nstr
            trimmed_read.sequence = masked_sequence
            trimmed_read.qualities = matches[0].read.qualities
            assert len(trimmed_read.sequence) == len(read)
        elif self.action is None:
            trimmed_read = read
        trimmed_read.match = matches[-1]
        trimmed_read.match_info = [match.get_info_record() for match in matches]
        self.with_adapters += 1
        return trimmed_read

    def __call__(self, read):
        """Cut the best-matching adapter(s) from one read; returns the modified read."""
        if len(read) == 0:
            return read
        matches = []
        trimmed_read = read
        for _ in range(self.times):
            match = self._best_match(trimmed_read)
            if match is None:
                break
            matches.append(match)
            trimmed_read = match.adapter.trimmed(match)
        return self._finish(read, trimmed_read, matches)

    def call_batch(self, reads):
        """Batched twin of ``__call__``: for each of the ``times`` rounds every adapter is
        matched against all still-active reads in one GPU call (``match_to_batch``), the
        best adapter per read is chosen with the reference's rule, and the reads are
        trimmed on the host."""
        out = list(reads)
        active = [i for i, r in enumerate(reads) if len(r) > 0]
        current = {i: reads[i] for i in active}
        matches = {i: [] for i in active}
        for _ in range(self.times):
            if not active:
                break
            batch = [current[i] for i in active]
            best = [None] * len(batch)
            for adapter in self.adapters:
                for k, match in enumerate(adapter.match_to_batch(batch)):
                    if match is None:
                        continue
                    if best[k] is None or match.matches > best[k].matches:
                        best[k] = match
            still = []
            for k, i in enumerate(active):
                if best[k] is None:
                    continue
                matches[i].append(best[k])
                current[i] = best[k].adapter.trimmed(best[k])
                still.append(i)
            active = still
        for i in matches:
            out[i] = self._finish(reads[i], current[i], matches[i])
        return out

    def summarize(self):
        adapters_summary = OrderedDict()
        for adapter in self.adapters:
            adapters_summary[adapter.name] = adapter.summarize()
        return dict(records_with_adapters=self.with_adapters, adapters=adapters_summary)


class ErrorCorrectorMixin(object):
    """Error correction of the overlapping part of a read pair.

    Args:
        mismatch_action: what to do on a mismatch between the overlapping portions of
            read1 and read2: 'liberal', 'conservative' or 'N'.
        min_qual_difference: minimum base-quality difference required to trust one read
            over the other.
    """

    def __init__(self, mismatch_action=None, min_qual_difference=1):
        self.mismatch_action = mismatch_action
        self.r1r2_min_qual_difference = min_qual_difference
        self.r2r1_min_qual_difference = -1 * min_qual_difference
        self.corrected_pairs = 0
        self.corrected_bp = [0, 0]

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
        return dict(records_corrected=self.corrected_pairs, bp_corrected=self.corrected_bp)


class InsertAdapterCutter(ReadPairModifier, ErrorCorrectorMixin):
    """AdapterCutter that uses InsertAligner to first try to identify the insert overlap
    before falling back to semi-global adapter alignment.

    Args:
        adapter1, adapter2: Adapters.
        action: 'trim', 'mask', 'lower' or None.
        mismatch_action: see ErrorCorrectorMixin.
        symmetric: assume the adapter appears at the same place in both reads.
        min_insert_overlap: minimum overlap of the reads for an insert match.
        aligner_args: further arguments of InsertAligner.
    """

    def __init__(self, adapter1, adapter2, action='trim', mismatch_action=None, symmetric=True,
                 min_insert_overlap=1, **aligner_args):
        ErrorCorrectorMixin.__init__(self, mismatch_action)
        self.adapter1 = adapter1
        self.adapter2 = adapter2
        self.aligner = InsertAligner(adapter1.sequence, adapter2.sequence, min_insert_overlap=min_insert_overlap,
                                     **aligner_args)
        self.min_insert_len = min_insert_overlap
        self.action = action
        self.symmetric = symmetric
        self.with_adapters = [0, 0]

    # -- the decision logic between the alignment stages (modifiers.py:397-446) ----------
    def _plan(self, read_lengths, match, fallback):
        """Given the insert match (or None) and, for the fallback, the two adapter matches,
        return (insert_match_for_correction or None, adapter_match1, adapter_match2)."""
        insert_match = None
        correct_errors = False
        if match:
            insert_match, adapter_match1, adapter_match2 = match
            correct_errors = self.mismatch_action is not None and insert_match[5] > 0
        else:
            adapter_match1, adapter_match2 = fallback
            # complementary adapter matches: perform error correction
            if (self.mismatch_action and adapter_match1 and adapter_match2 and
                    adapter_match1.rstart == adapter_match2.rstart):
                insert_match = (read_lengths[1] - adapter_match1.rstart, read_lengths[1], 0, adapter_match1.rstart)
                correct_errors = True
        # exactly one of the two alignments failed and symmetric: duplicate the good one
        if self.symmetric and sum(bool(m) for m in (adapter_match1, adapter_match2)) == 1:

            def create_symmetric_match(match, read_len):
                if match.rstart > read_len:
                    return None
                match = match.copy()
                # unequal read lengths: end the match at the read end ('matches'/'errors'
                # are then off, as in the reference)
                if match.rstop < read_len:
                    match.astop -= (read_len - match.rstop)
                    match.rstop = read_len
                return match

            if adapter_match1:
                adapter_match2 = create_symmetric_match(adapter_match1, read_lengths[1])
            else:
                adapter_match1 = create_symmetric_match(adapter_match2, read_lengths[0])
            if self.mismatch_action and not insert_match and adapter_match1 and adapter_match2:
                insert_match = (read_lengths[1] - adapter_match1.rstart, read_lengths[1], 0, adapter_match1.rstart)
                correct_errors = True
        return (insert_match if correct_errors else None), adapter_match1, adapter_match2

    def __call__(self, read1, read2):
        read_lengths = [len(r) for r in (read1, read2)]
        if any(l < self.min_insert_len for l in read_lengths):
            return (read1, read2)
        match = self.aligner.match_insert(read1.sequence, read2.sequence)
        read1.insert_overlap = read2.insert_overlap = (match is not None)
        fallback = None
        if not match:
            fallback = (self.adapter1.match_to(read1), self.adapter2.match_to(read2))
        to_correct, adapter_match1, adapter_match2 = self._plan(read_lengths, match, fallback)
        if to_correct is not None:
            self.correct_errors(read1, read2, to_correct, truncate_seqs=True)
        return (self.trim(read1, self.adapter1, adapter_match1, 0), self.trim(read2, self.adapter2, adapter_match2, 1))

    def call_batch(self, reads1, reads2):
        """Batched twin of ``__call__``: stage 1 insert matching of all pairs in one GPU
        call, stage 2 the two adapters' semi-global alignment over the pairs without an
        insert match, stage 3 error correction of the flagged pairs in one GPU call, then
        per-read trimming on the host.  Returns the list of (read1, read2) results."""
        n = len(reads1)
        out = [None] * n
        act = [i for i in range(n) if not any(len(r) < self.min_insert_len for r in (reads1[i], reads2[i]))]
        act_set = set(act)
        for i in range(n):
            if i not in act_set:
                out[i] = (reads1[i], reads2[i])
        if not act:
            return out
        width = max(max(len(reads1[i]), len(reads2[i])) for i in act)
        b1 = self.aligner.pack([reads1[i].sequence for i in act] + ["A" * width])
        b2 = self.aligner.pack([reads2[i].sequence for i in act] + ["A" * width], check=True)
        results = self.aligner.match_insert_batch(b1, b2).results()[:-1]
        miss = [k for k, r in enumerate(results) if r is None]
        fb1 = self.adapter1.match_to_batch([reads1[act[k]] for k in miss]) if miss else []
        fb2 = self.adapter2.match_to_batch([reads2[act[k]] for k in miss]) if miss else []
        fallback = {k: (a, b) for k, a, b in zip(miss, fb1, fb2)}
        plans = []
        for k, i in enumerate(act):
            reads1[i].insert_overlap = reads2[i].insert_overlap = (results[k] is not None)
            plans.append(self._plan([len(reads1[i]), len(reads2[i])], results[k], fallback.get(k)))
        self.correct_errors_batch([reads1[i] for i in act], [reads2[i] for i in act], [p[0] for p in plans],
                                  truncate_seqs=True)
        for k, i in enumerate(act):
            out[i] = (self.trim(reads1[i], self.adapter1, plans[k][1], 0),
                      self.trim(reads2[i], self.adapter2, plans[k][2], 1))
        return out

    def trim(self, read, adapter, match, read_idx):
        """Trim an adapter from a read according to the match."""
        if not match:
            read.match = None
            read.match_info = None
            return read
        match.adapter = adapter
        match.read = read
        match.front = False
        if self.action is None or match.rstart >= len(read):
            trimmed_read = read
        else:
            trimmed_read = adapter.trimmed(match)
            if self.action == 'mask':
                masked_sequence = trimmed_read.sequence
                masked_sequence += 'N' * (len(read) - len(trimmed_read))
                trimmed_read.sequence = masked_sequence
                trimmed_read.qualities = read.qualities
        trimmed_read.match = match
        trimmed_read.match_info = [match.get_info_record()]
        self.with_adapters[read_idx] += 1
        return trimmed_read

    def summarize(self):
        adapters_summary = tuple({adapter.name: adapter.summarize()} for adapter in (self.adapter1, self.adapter2))
        summary = dict(records_with_adapters=self.with_adapters, adapters=adapters_summary)
        if self.mismatch_action:
            summary.update(ErrorCorrectorMixin.summarize(self))
        return summary


class MergeOverlapping(ReadPairModifier, ErrorCorrectorMixin):
    """Merge overlapping reads; the merged read is stored in read1 and read2 becomes None
    (reference: commands/trim/modifiers.py:864-931).  The per-pair alignment
    ``Aligner(reverse_complement(read2), error_rate, flags).locate(read1)`` runs on the GPU for
    the whole batch (``PairAligner``, atr_locate_pairs_batch); reverse complement included."""

    def __init__(self, min_overlap=0.9, error_rate=0.1, mismatch_action=None):
        ErrorCorrectorMixin.__init__(self, mismatch_action)
        self.min_overlap = int(min_overlap) if min_overlap > 1 else min_overlap
        self.error_rate = error_rate
        self._aligners = {}

    def _aligner(self, flags):
        if flags not in self._aligners:
            self._aligners[flags] = align.PairAligner(self.error_rate, flags, revcomp_ref=True)
        return self._aligners[flags]

    def __call__(self, read1, read2):
        return self.call_batch([read1], [read2])[0]

    def call_batch(self, reads1, reads2):
        """Batched twin of ``__call__``: returns the list of (read1, read2-or-None) pairs."""
        out = [(r1, r2) for r1, r2 in zip(reads1, reads2)]
        plan = {}                                            # flags -> [(index, min_overlap)]
        for i, (read1, read2) in enumerate(out):
            len1, len2 = len(read1.sequence), len(read2.sequence)
            min_overlap = self.min_overlap
            if min_overlap <= 1:
                min_overlap = max(2, round(self.min_overlap * min(len1, len2)))
            if len1 < min_overlap or len2 < min_overlap:
                continue
            insert_matched = read1.insert_overlap and read2.insert_overlap
            # an insert overlap with a 3' overhang was already found: constrain the alignment
            flags = (align.START_WITHIN_SEQ1 | align.STOP_WITHIN_SEQ2) if insert_matched else align.SEMIGLOBAL
            plan.setdefault(flags, []).append((i, min_overlap, bool(insert_matched)))
        for flags, todo in plan.items():
            idx = [t[0] for t in todo]
            for i in idx:
                reverse_complement(out[i][1].sequence)       # KeyError on a base without complement, as the reference
            # reference = reverse_complement(read2) (formed on the device), query = read1
            alignments = self._aligner(flags).locate_batch([out[i][1].sequence for i in idx],
                                                           [out[i][0].sequence for i in idx]).tuples()
            for (i, min_overlap, insert_matched), alignment in zip(todo, alignments):
                if alignment:
                    out[i] = self._merge(out[i][0], out[i][1], alignment, min_overlap, insert_matched)
        return out

    def _merge(self, read1, read2, alignment, min_overlap, insert_matched):
        len1, len2 = len(read1.sequence), len(read2.sequence)
        read2_rc = reverse_complement(read2.sequence)
        r2_start, r2_stop, r1_start, r1_stop, matches, errors = alignment
        if matches < min_overlap:
            return (read1, read2)
        # only correct errors if that was not already done by the insert aligner
        if self.mismatch_action and errors > 0 and not insert_matched:
            self.correct_errors(read1, read2, alignment)
        if r2_start == 0 and r2_stop == len2:
            pass                                             # r2 is fully contained in r1
        elif r1_start == 0 and r1_stop == len1:
            read1.sequence = read2_rc                        # r1 is fully contained in r2
            read1.qualities = "".join(reversed(read2.qualities))
        elif r1_start > 0:
            read1.sequence += read2_rc[r2_stop:]
            if read1.qualities and read2.qualities:
                read1.qualities += "".join(reversed(read2.qualities))[r2_stop:]
        elif r2_start > 0:
            read1.sequence = read2_rc + read1.sequence[r1_stop:]
            if read1.qualities and read2.qualities:
                read1.qualities = "".join(reversed(read2.qualities)) + read1.qualities[r1_stop:]
        else:
            raise ValueError("Invalid alignment while trying to merge read {}: {}".format(
                read1.name, ",".join(str(i) for i in alignment)))
        read1.merged = True
        return (read1, None)

