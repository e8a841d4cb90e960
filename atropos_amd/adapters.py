# coding: utf-8
"""Adapters: the boundary objects that call the aligner (reference
atropos/adapters/__init__.py: ADAPTER_TYPES :41-73, AdapterParser :80-229, Adapter
:231-505, LinkedMatch/LinkedAdapter :612-745, parse_braces :933-970).

Same classes, arguments and behaviour; ``match_to`` keeps its per-read form and gains a
batched twin (``match_to_batch``) that sends the whole batch through the GPU kernels in
one call and applies the reference's post-filters to the result records.
ColorspaceAdapter and AdapterCache are out of scope.
"""
import itertools
import re
from collections import defaultdict

import numpy as np
import torch

from . import _lib
from . import align
from .batch import ReadBatch
from .align import Match
from .util import IUPAC_BASES, GC_BASES, rmp_table

START_WITHIN_SEQ1, START_WITHIN_SEQ2 = align.START_WITHIN_SEQ1, align.START_WITHIN_SEQ2
STOP_WITHIN_SEQ1, STOP_WITHIN_SEQ2 = align.STOP_WITHIN_SEQ1, align.STOP_WITHIN_SEQ2


class AdapterType(object):
    """Adapter type name, description and alignment flags."""

    def __init__(self, name, desc, *flags):
        self.name = name
        self.desc = desc
        self.flags = flags[0]
        for flag in flags[1:]:
            self.flags |= flag

    def asdict(self):
        return dict(name=self.name, desc=self.desc, flags=self.flags)


ADAPTER_TYPES = dict(
    back=AdapterType('back', "regular 3'", START_WITHIN_SEQ2, STOP_WITHIN_SEQ2, STOP_WITHIN_SEQ1),
    front=AdapterType('front', "regular 5'", START_WITHIN_SEQ2, STOP_WITHIN_SEQ2, START_WITHIN_SEQ1),
    prefix=AdapterType('prefix', "anchored 5'", STOP_WITHIN_SEQ2),
    suffix=AdapterType('suffix', "anchored 3'", START_WITHIN_SEQ2),
    anywhere=AdapterType('anywhere', "variable 5'/3'", align.SEMIGLOBAL),
    linked=AdapterType('linked', 'linked', 'linked'))

BACK = ADAPTER_TYPES['back'].flags              # 14
FRONT = ADAPTER_TYPES['front'].flags            # 11
PREFIX = ADAPTER_TYPES['prefix'].flags          # 8
SUFFIX = ADAPTER_TYPES['suffix'].flags          # 2
ANYWHERE = ADAPTER_TYPES['anywhere'].flags      # 15
LINKED = ADAPTER_TYPES['linked'].flags


def where_int_to_dict(where):
    for adapter_type in ADAPTER_TYPES.values():
        if where == adapter_type.flags:
            return adapter_type.asdict()
    raise ValueError("Invalid WHERE value: {}".format(where))


ADAPTER_ID_GENERATOR = itertools.count(1)


def _generate_adapter_name():
    return str(next(ADAPTER_ID_GENERATOR))


def parse_braces(sequence):
    """Replace all occurrences of ``x{n}`` (x any character) with n occurrences of x;
    ``ValueError`` if the expression cannot be parsed.  ``TGA{5}CT`` -> ``TGAAAAACT``."""
    result = ''
    state = None            # None | a plain token | '{' | repeat count (int)
    for token in re.split(r'(\{|\})', sequence):
        if token == '':
            continue
        if state is None:
            if token == '{':
                raise ValueError('"{" must be used after a character')
            if token == '}':
                raise ValueError('"}" cannot be used here')
            state = token
            result += token
        elif state == '{':
            state = int(token)
            if not 0 <= state <= 10000:
                raise ValueError('Value {} invalid'.format(state))
        elif isinstance(state, int):
            if token != '}':
                raise ValueError('"}" expected')
            result = result[:-1] + result[-1] * state
            state = None
        else:
            if token != '{':
                raise ValueError('Expected "{"')
            state = '{'
    if isinstance(state, int) or state == '{':
        raise ValueError("Unterminated expression")
    return result


def _extract_name_from_spec(spec):
    fields = spec.split('=', 1)
    name = None
    if len(fields) > 1:
        name, spec = fields
        name = name.strip()
    return name, spec.strip()


def _seq_of(read):
    return read if isinstance(read, str) else read.sequence


class AsciiSource(object):
    """Read source of the device-resident matchers: an upper-case ASCII matrix on the GPU
    (optionally sliced per read by ``starts``), packed on demand once per translate table."""

    def __init__(self, ascii_upper, lens=None, starts=None, batch=None):
        self.ascii_upper, self.lens, self.starts = ascii_upper, lens, starts
        self.n = ascii_upper.shape[0]
        self._batches = {}
        if batch is not None and starts is None:
            self._batches[(batch.table_kind, batch.table)] = batch

    def batch(self, table_kind, table):
        key = (table_kind, bytes(table))
        if key not in self._batches:
            self._batches[key] = ReadBatch.from_ascii(self.ascii_upper, self.lens, None, table_kind, table,
                                                      _lib.get_backend(), starts=self.starts)
        return self._batches[key]

    def ascii(self):
        if self.starts is not None:
            raise NotImplementedError("sliced reads with anchored no-indel adapters")
        return self.ascii_upper, self.lens

    def sliced(self, starts):
        """The source of the reads ``read[start:]``."""
        if self.starts is not None:
            starts = starts + self.starts
        return AsciiSource(self.ascii_upper, self.lens, starts)


class Adapter(object):
    """An adapter knows how to match itself to a read: where it may sit within the read
    and how wildcard characters are interpreted.

    Args (as in the reference): sequence (upper-cased, U -> T, ``x{n}`` expanded), where
    (BACK, FRONT, PREFIX, SUFFIX or ANYWHERE), max_error_rate, min_overlap,
    read_wildcards, adapter_wildcards, name, indels, indel_cost, match_probability
    (``callable(matches, size)``), max_rmp, gc_content.
    """

    def __init__(self, sequence, where, max_error_rate=0.1, min_overlap=3, read_wildcards=False,
                 adapter_wildcards=True, name=None, indels=True, indel_cost=1, match_probability=None, max_rmp=None,
                 gc_content=0.5, alphabet=None):
        if len(sequence) == 0:
            raise ValueError("Empty adapter sequence")
        sequence = parse_braces(sequence.upper().replace('U', 'T'))
        seq_set = set(sequence)
        if seq_set <= set('ACGT'):
            adapter_wildcards = False
        if adapter_wildcards and not seq_set <= IUPAC_BASES:
            raise ValueError("Invalid character(s) in adapter sequence: {}".format(','.join(seq_set - IUPAC_BASES)))
        if alphabet is not None:
            for character in sequence:
                if character not in alphabet:
                    raise ValueError("Character {!r} is not in the alphabet".format(character))
        self.debug = False
        self.name = _generate_adapter_name() if name is None else name
        self.sequence = sequence
        self.where = where
        self.max_error_rate = max_error_rate
        self.min_overlap = min(min_overlap, len(self.sequence))
        self.match_probability = match_probability
        self.max_rmp = max_rmp
        self.gc_content = gc_content
        self.indels = indels
        self.adapter_wildcards = adapter_wildcards
        self.read_wildcards = read_wildcards
        trimmers = {FRONT: self._trimmed_front, PREFIX: self._trimmed_front, BACK: self._trimmed_back,
                    SUFFIX: self._trimmed_back, ANYWHERE: self._trimmed_anywhere}
        self.trimmed = trimmers[where]
        self._front_flag = None if where == ANYWHERE else where not in (BACK, SUFFIX)
        # statistics about the removed sequences
        self.lengths_front = defaultdict(int)
        self.lengths_back = defaultdict(int)
        self.errors_front = defaultdict(lambda: defaultdict(int))
        self.errors_back = defaultdict(lambda: defaultdict(int))
        self.adjacent_bases = {'A': 0, 'C': 0, 'G': 0, 'T': 0, '': 0}
        self.aligner = align.Aligner(self.sequence, self.max_error_rate, flags=self.where,
                                     wildcard_ref=self.adapter_wildcards, wildcard_query=self.read_wildcards)
        self.aligner.min_overlap = self.min_overlap
        # when indels are disallowed the aligner simply prices them out (:316-322)
        self._indel_cost = indel_cost if self.indels else 100000
        self.aligner.indel_cost = self._indel_cost
        self._exact_aligner = None
        self._rmp_cache = None
        self._rmp_device = None

    def __repr__(self):
        return ('<Adapter(name="{name}", sequence="{sequence}", where={where}, max_error_rate={max_error_rate}, '
                'min_overlap={min_overlap}, read_wildcards={read_wildcards}, '
                'adapter_wildcards={adapter_wildcards}, indels={indels})>').format(**vars(self))

    def enable_debug(self):
        self.debug = True
        self.aligner.enable_debug()

    # ------------------------------------------------------------------ matching
    def _exact_position(self, read_seq):
        """The exact-match shortcut of match_to (:351-367): first literal occurrence."""
        if self.where == PREFIX:
            return 0 if read_seq.startswith(self.sequence) else -1
        if self.where == SUFFIX:
            return len(read_seq) - len(self.sequence) if read_seq.endswith(self.sequence) else -1
        return read_seq.find(self.sequence)

    def _accept(self, astart, astop, matches, errors):
        """The post-filter of match_to (:386-398); note the DIVISION, not the DP's product."""
        size = astop - astart
        return ((size >= self.min_overlap and errors / size <= self.max_error_rate) and
                (self.max_rmp is None or self.match_probability(matches, size) <= self.max_rmp))

    def match_to(self, read):
        """Attempt to match this adapter to the given read; returns a Match or None if
        the criteria (minimum overlap, maximum error rate, random-match probability) are
        not met."""
        read_seq = read.sequence.upper()
        # try to find an exact match first unless wildcards are allowed
        if not self.adapter_wildcards:
            pos = self._exact_position(read_seq)
            if pos >= 0:
                seqlen = len(self.sequence)
                return Match(0, seqlen, pos, pos + seqlen, seqlen, 0, self._front_flag, self, read)
        # approximate matching
        if not self.indels and self.where in (PREFIX, SUFFIX):
            compare = align.compare_prefixes if self.where == PREFIX else align.compare_suffixes
            alignment = compare(self.sequence, read_seq, wildcard_ref=self.adapter_wildcards,
                                wildcard_query=self.read_wildcards)
        else:
            alignment = self.aligner.locate(read_seq)
        if alignment:
            astart, astop, rstart, rstop, matches, errors = alignment
            if self._accept(astart, astop, matches, errors):
                return Match(astart, astop, rstart, rstop, matches, errors, self._front_flag, self, read)
        return None

    def _rmp_by_size(self):
        if self._rmp_cache is None:
            self._rmp_cache = rmp_table(self.match_probability, len(self.sequence))
        return self._rmp_cache

    def locate_records(self, reads_upper):
        """Raw device step of the batched match: the alignment records (int16 [n, 8]
        numpy) for already upper-cased reads (list of str or uint8 [n, width] tensor)
        plus, when the literal exact-match shortcut cannot be read off those records, the
        records of the literal search."""
        if not self.indels and self.where in (PREFIX, SUFFIX):
            rec = align.compare_batch(self.sequence, reads_upper, self.adapter_wildcards, self.read_wildcards,
                                      suffix=(self.where == SUFFIX)).cpu().numpy()
        else:
            rec = self.aligner.locate_batch(reads_upper).numpy()
        exact = None
        if not self.adapter_wildcards and self.read_wildcards:
            # with read wildcards the DP may prefer an earlier wildcard match over the first
            # literal occurrence the shortcut returns: search literally as well
            if self._exact_aligner is None:
                self._exact_aligner = align.Aligner(self.sequence, 0.0, flags=self.where,
                                                    min_overlap=len(self.sequence))
            exact = self._exact_aligner.locate_batch(reads_upper).numpy()
        return rec, exact

    def match_to_batch(self, reads):
        """Batched ``match_to``: one GPU call for the whole batch, then the reference's
        shortcut/post-filter rules on the records.  ``reads``: list of read objects (or
        plain str).  Returns a list of Match / None."""
        seqs = [_seq_of(r).upper() for r in reads]
        if not seqs:
            return []
        rec, exact = self.locate_records(seqs)
        m = len(self.sequence)
        rec = rec.astype(np.int64)
        found = rec[:, 1] >= 0
        astart, astop, matches, errors = rec[:, 0], rec[:, 1], rec[:, 4], rec[:, 5]
        size = np.where(found, astop - astart, 1)
        with np.errstate(divide='ignore', invalid='ignore'):
            ok = found & (size >= self.min_overlap) & (errors / size <= self.max_error_rate)
        if self.max_rmp is not None:
            table = self._rmp_by_size()
            probs = table[np.clip(size, 0, m), np.clip(matches, 0, m)]
            ok &= probs <= self.max_rmp
        use_exact = np.zeros(len(seqs), dtype=bool)
        if not self.adapter_wildcards:
            if exact is None:
                # literal compare mode: an exact full-length occurrence is what the DP returns
                # for it (most matches, zero errors, leftmost); the shortcut bypasses the filters
                if not self.indels and self.where in (PREFIX, SUFFIX):
                    full = found & (matches == m) & (errors == 0) & (astop - astart == m)
                else:
                    full = found & (matches == m) & (errors == 0)
                ok |= full
            else:
                use_exact = exact[:, 1] >= 0
        out = []
        for i, read in enumerate(reads):
            if use_exact[i]:
                row = exact[i]
            elif ok[i]:
                row = rec[i]
            else:
                out.append(None)
                continue
            out.append(Match(int(row[0]), int(row[1]), int(row[2]), int(row[3]), int(row[4]), int(row[5]),
                             self._front_flag, self, None if isinstance(read, str) else read))
        return out

    # ------------------------------------------------------------------ device-resident twin
    def match_records(self, ascii_upper, lens=None, starts=None, batch=None):
        """``match_to`` for a batch that never leaves the GPU.  ``ascii_upper``: uint8
        [n, width] tensor of UPPER-CASE ASCII reads on the device (``upper_ascii()``),
        ``lens``/``starts``: optional int32 tensors (the read is ``row[start:len]``);
        ``batch``: the same reads already packed with this aligner's table (saves the pack).
        Returns the int16 [n, 8] record tensor with ``refstop = -1`` wherever ``match_to``
        would return None: alignment, exact-match shortcut and post-filters
        (adapters/__init__.py:338-400) are all evaluated on the device."""
        return self.match_source(AsciiSource(ascii_upper, lens, starts, batch))

    def match_source(self, source):
        """``match_records`` over any read source: an object with ``n``, ``batch(table_kind,
        table)`` (the reads 4-bit packed with that table) and ``ascii()`` ((uint8 [n, width]
        upper-case matrix, lens) -- only anchored no-indel adapters need it)."""
        be = _lib.get_backend()
        m = len(self.sequence)
        if not self.indels and self.where in (PREFIX, SUFFIX):
            ascii_upper, lens = source.ascii()
            rec = align.compare_batch(self.sequence, ascii_upper, self.adapter_wildcards, self.read_wildcards,
                                      suffix=(self.where == SUFFIX), lens=lens)
        else:
            rec = self.aligner.locate_batch(source.batch(self.aligner.table_kind, self.aligner._table)).records
        rmp_t = None
        if self.max_rmp is not None:
            if self._rmp_device is None or self._rmp_device.device != be.device:
                self._rmp_device = torch.from_numpy(np.ascontiguousarray(self._rmp_by_size())).to(be.device)
            rmp_t = self._rmp_device
        # the shortcut applies when the literal compare mode makes "full length, zero errors"
        # exactly what str.find / startswith / endswith would have reported
        accept_full = (not self.adapter_wildcards) and (not self.read_wildcards)
        rec = be.adapter_postfilter(rec, m, self.min_overlap, float(self.max_error_rate), rmp_t, self.max_rmp,
                                    accept_full)
        if not self.adapter_wildcards and self.read_wildcards:
            # the literal first occurrence wins over whatever the wildcard DP found
            if self._exact_aligner is None:
                self._exact_aligner = align.Aligner(self.sequence, 0.0, flags=self.where, min_overlap=m)
            ex = self._exact_aligner.locate_batch(
                source.batch(self._exact_aligner.table_kind, self._exact_aligner._table)).records
            rec = torch.where((ex[:, 1] >= 0)[:, None], ex, rec)
        return rec

    # ------------------------------------------------------------------ trimming
    def _trimmed_anywhere(self, match):
        return self._trimmed_front(match) if match.front else self._trimmed_back(match)

    def _trimmed_front(self, match):
        self.lengths_front[match.rstop] += 1
        self.errors_front[match.rstop][match.errors] += 1
        return match.read[match.rstop:]

    def _trimmed_back(self, match):
        self.lengths_back[len(match.read) - match.rstart] += 1
        self.errors_back[len(match.read) - match.rstart][match.errors] += 1
        adjacent_base = match.read.sequence[match.rstart - 1:match.rstart]
        if adjacent_base not in 'ACGT':
            adjacent_base = ''
        self.adjacent_bases[adjacent_base] += 1
        return match.read[:match.rstart]

    def __len__(self):
        return len(self.sequence)

    def random_match_probabilities(self):
        """Probability that the first i bases of this adapter match a random sequence
        (indels not taken into account), for i = 0..len."""
        seq = self.sequence[::-1] if self._front_flag else self.sequence
        base_probs = (self.gc_content / 2.0, (1 - self.gc_content) / 2.0)
        probabilities = [1.0] + ([0] * len(seq))
        c_bases = frozenset(GC_BASES if self.adapter_wildcards else 'GC')
        cur_p = 1.0
        for idx, base in enumerate(seq, 1):
            cur_p *= base_probs[0 if base in c_bases else 1]
            probabilities[idx] = cur_p
        return probabilities

    def summarize(self):
        total_front = sum(self.lengths_front.values())
        total_back = sum(self.lengths_back.values())
        where = self.where
        assert (where in (ANYWHERE, LINKED) or (where in (BACK, SUFFIX) and total_front == 0) or
                (where in (FRONT, PREFIX) and total_back == 0))
        stats = dict(adapter_class=self.__class__.__name__, total_front=total_front, total_back=total_back,
                     total=total_front + total_back, match_probabilities=self.random_match_probabilities(),
                     where=where_int_to_dict(where), sequence=self.sequence, max_error_rate=self.max_error_rate)
        if where in (ANYWHERE, FRONT, PREFIX):
            stats["lengths_front"] = dict(self.lengths_front)
            stats["errors_front"] = {k: dict(v) for k, v in self.errors_front.items()}
        if where in (ANYWHERE, BACK, SUFFIX):
            stats["lengths_back"] = dict(self.lengths_back)
            stats["errors_back"] = {k: dict(v) for k, v in self.errors_back.items()}
        if where in (BACK, SUFFIX):
            stats["adjacent_bases"] = dict(self.adjacent_bases)
        return stats


class LinkedMatch(object):
    """A match of a LinkedAdapter: the front match and (possibly None) the back match."""

    def __init__(self, front_match, back_match, adapter):
        self.front_match = front_match
        self.back_match = back_match
        self.adapter = adapter
        assert front_match is not None

    def get_info_record(self):
        if self.back_match:
            return self.back_match.get_info_record()
        return self.front_match.get_info_record()


class LinkedAdapter(object):
    """An adapter with linked front (anchored 5') and back (3') sequences; the back
    adapter is only searched for when the front adapter was found, in the read with the
    front match removed."""

    def __init__(self, front_sequence, back_sequence, front_anchored=True, back_anchored=False, name=None, **kwargs):
        assert front_anchored and not back_anchored
        where1 = PREFIX if front_anchored else FRONT
        where2 = SUFFIX if back_anchored else BACK
        self.front_anchored = front_anchored
        self.back_anchored = back_anchored
        self.where = LINKED
        self.name = _generate_adapter_name() if name is None else name
        self.front_adapter = Adapter(front_sequence, where=where1, name=None, **kwargs)
        self.back_adapter = Adapter(back_sequence, where=where2, name=None, **kwargs)

    def enable_debug(self):
        self.front_adapter.enable_debug()
        self.back_adapter.enable_debug()

    def match_to(self, read):
        front_match = self.front_adapter.match_to(read)
        if front_match is None:
            return None
        read = read[front_match.rstop:]
        back_match = self.back_adapter.match_to(read)
        return LinkedMatch(front_match, back_match, self)

    def match_to_batch(self, reads):
        """Batched ``match_to``: the front adapter over the whole batch, then the back
        adapter over the remainders of the reads that had a front match."""
        fronts = self.front_adapter.match_to_batch(reads)
        idx = [i for i, fm in enumerate(fronts) if fm is not None]
        rest = [reads[i][fronts[i].rstop:] for i in idx]
        backs = self.back_adapter.match_to_batch(rest) if rest else []
        out = [None] * len(reads)
        for i, bm in zip(idx, backs):
            out[i] = LinkedMatch(fronts[i], bm, self)
        return out

    def match_records(self, ascii_upper, lens=None):
        """Device-resident twin of ``match_to``: returns (front_records, back_records), both
        int16 [n, 8] on the device; the back records refer to the read with the front match
        removed (coordinates relative to ``read[front.rstop:]``) and are -1 where the front
        adapter did not match."""
        _which, _count, front, back = _linked_set([self]).match_source(AsciiSource(ascii_upper, lens))
        return front, back

    def trimmed(self, match):
        front_trimmed = self.front_adapter.trimmed(match.front_match)
        if match.back_match:
            return self.back_adapter.trimmed(match.back_match)
        return front_trimmed

    def summarize(self):
        fa, ba = self.front_adapter, self.back_adapter
        total_front = sum(fa.lengths_front.values())
        total_back = sum(ba.lengths_back.values())
        return dict(total_front=total_front, total_back=total_back, total=total_front + total_back,
                    where=where_int_to_dict(self.where), front_sequence=fa.sequence, back_sequence=ba.sequence,
                    front_match_probabilities=fa.random_match_probabilities(),
                    back_match_probabilities=ba.random_match_probabilities(),
                    front_max_error_rate=fa.max_error_rate, back_max_error_rate=ba.max_error_rate,
                    front_lengths_front=dict(fa.lengths_front), front_lengths_back=dict(fa.lengths_back),
                    back_lengths_front=dict(ba.lengths_front), back_lengths_back=dict(ba.lengths_back))


class AdapterParser(object):
    """Factory for Adapter objects that all share the same parameters (error rate, indels
    ...); ``**kwargs`` go to the Adapter constructors.  Supports the command-line
    notation: ``name=SEQ``, ``^SEQ`` (anchored 5'), ``SEQ$`` (anchored 3'),
    ``SEQ1...SEQ2`` (linked).  (``file:`` specs, colorspace and the adapter cache are out
    of scope.)"""

    def __init__(self, **kwargs):
        self.constructor_args = kwargs

    def parse(self, spec, cmdline_type='back'):
        if spec.startswith('file:'):
            raise NotImplementedError("file: adapter specs need the FASTA reader, which is out of scope")
        yield self.parse_from_spec(spec, cmdline_type)

    def parse_from_spec(self, spec, cmdline_type='back', name=None):
        if cmdline_type not in ADAPTER_TYPES:
            raise ValueError('cmdline_type cannot be {0!r}'.format(cmdline_type))
        if spec is None:
            raise ValueError('Either name or spec must be given')
        orig_spec = spec
        where = ADAPTER_TYPES[cmdline_type].flags
        if name is None:
            name, spec = _extract_name_from_spec(spec)
        front_anchored = back_anchored = False
        if spec.startswith('^'):
            spec = spec[1:]
            front_anchored = True
        if spec.endswith('$'):
            spec = spec[:-1]
            back_anchored = True
        sequence1, middle, sequence2 = spec.partition('...')
        if where == ANYWHERE:
            if front_anchored or back_anchored:
                raise ValueError("'anywhere' (-b) adapters may not be anchored")
            if middle == '...':
                raise ValueError("'anywhere' (-b) adapters may not be linked")
            return Adapter(sequence=spec, where=where, name=name, **self.constructor_args)
        assert where == FRONT or where == BACK
        if middle == '...':
            if not sequence1:
                if where == BACK:           # -a ...ADAPTER
                    spec = sequence2
                else:                       # -g ...ADAPTER
                    raise ValueError('Invalid adapter specification')
            elif not sequence2:
                if where == BACK:           # -a ADAPTER...
                    spec = sequence1
                    where = FRONT
                    front_anchored = True
                else:                       # -g ADAPTER...
                    spec = sequence1
            else:
                if where == BACK:           # the 5' adapter is anchored automatically with -a
                    front_anchored = True
                return LinkedAdapter(sequence1, sequence2, name=name, front_anchored=front_anchored,
                                     back_anchored=back_anchored, **self.constructor_args)
        if front_anchored and back_anchored:
            raise ValueError('Trying to use both "^" and "$" in adapter specification {!r}'.format(orig_spec))
        if front_anchored:
            if where == BACK:
                raise ValueError("Cannot anchor the 3' adapter at its 5' end")
            where = PREFIX
        elif back_anchored:
            if where == FRONT:
                raise ValueError("Cannot anchor 5' adapter at 3' end")
            where = SUFFIX
        return Adapter(sequence=spec, where=where, name=name, **self.constructor_args)

    def parse_multi(self, back=None, anywhere=None, front=None):
        adapters = []
        for specs, cmdline_type in ((back, 'back'), (anywhere, 'anywhere'), (front, 'front')):
            for spec in specs or ():
                adapters.extend(self.parse(spec, cmdline_type))
        return adapters


class LinkedSet(object):
    """Device matcher for a list of LinkedAdapters whose anchored 5' parts are mutually exclusive
    (AdapterCutter._best_match over LinkedAdapter.match_to, modifiers.py:107-122 /
    adapters/__init__.py:671-690): ONE fused kernel pipeline per batch (atr_linked_match_batch)
    -- each read is loaded once, every 5' part is tried, and the 3' part of the one that matched
    is located in ``read[front.rstop:]`` without re-packing.  ``fused`` is False when the set is
    outside that pipeline's envelope (include/atropos_hip.h, atr_linked_create); the parts are then
    matched one after the other with the per-adapter kernels."""

    def __init__(self, linked_adapters):
        self.adapters = list(linked_adapters)
        self._backend = be = _lib.get_backend()
        self._handle = None
        self._tables = []                     # device RMP tables stay alive with the set
        first = self.adapters[0].front_adapter.aligner
        self.table_kind, self.table = first.table_kind, first._table
        parts = [p for la in self.adapters for p in (la.front_adapter, la.back_adapter)]
        if len(self.adapters) > _lib.LINKED_MAX_ADAPTERS or any(not p.indels for p in parts):
            return                            # anchored parts without indels are plain prefix compares
        specs = []
        for la in self.adapters:
            spec = _lib.LinkedAdapterSpec()
            for side, part in (("front", la.front_adapter), ("back", la.back_adapter)):
                setattr(spec, side, part.aligner._handle)
                setattr(spec, side + "_exact_shortcut", int(not part.adapter_wildcards))
                if part.max_rmp is not None:
                    table = torch.from_numpy(np.ascontiguousarray(part._rmp_by_size())).to(be.device)
                    self._tables.append(table)
                    setattr(spec, "d_%s_rmp" % side, table.data_ptr())
                    setattr(spec, side + "_rmp_ld", table.shape[1])
                    setattr(spec, side + "_max_rmp", float(part.max_rmp))
            specs.append(spec)
        try:
            self._handle = be.linked_create(specs)
        except _lib.AtroposUnsupported:
            self._handle = None

    @property
    def fused(self):
        return self._handle is not None

    def __del__(self):
        try:
            if self._handle is not None:
                self._backend.linked_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def match_source(self, source, active=None):
        """(which, count, front, back) for a read source (``AsciiSource`` / ``RecordSource``):
        int32 index of the first linked adapter whose 5' part matches (-1: none), the number of
        5' parts that match, and the two int16 [n, 8] record tensors; the 3' records refer to
        ``read[front.rstop:]`` and are None-records where no 5' part matched."""
        if self._handle is None:
            return _linked_records_stepwise(self.adapters, source, active)
        batch = source.batch(self.table_kind, self.table)
        wc, front, back = self._backend.linked_match_batch(self._handle, batch.packed, batch.lens, batch.nreads,
                                                           batch.max_len)
        which, count = wc[:, 0].to(torch.int32), wc[:, 1].to(torch.int32)
        if active is not None:                # reads that already left the adapter rounds
            off = active == 0
            which = which.masked_fill(off, -1)
            count = count.masked_fill(off, 0)
            front[off, 1] = -1
            back[off, 1] = -1
        return which, count, front, back


def _linked_set(linked_adapters):
    """The LinkedSet of a list of linked adapters, built once and kept on its first member."""
    key = tuple(id(la) for la in linked_adapters)
    cached = getattr(linked_adapters[0], "_linked_set", None)
    if cached is None or cached[0] != key or cached[1]._backend is not _lib.get_backend():
        cached = (key, LinkedSet(linked_adapters))
        linked_adapters[0]._linked_set = cached
    return cached[1]


def linked_best_records(linked_adapters, ascii_upper, lens=None):
    """Linked adapters whose anchored 5' parts are mutually exclusive (BASELINE config C4) on a
    uint8 [n, width] matrix of upper-case reads.  Returns (which, front, back): the index of the
    matching linked adapter per read (-1: none, -2: more than one -- the reference's AdapterCutter
    raises AttributeError in that case, modifiers.py:120) and the two int16 [n, 8] record tensors."""
    which, count, front, back = _linked_set(linked_adapters).match_source(AsciiSource(ascii_upper, lens))
    which = torch.where(count > 1, torch.full_like(which, -2), which)
    return which, front, back


def upper_ascii(ascii_2d):
    """Device twin of ``read.sequence.upper()`` (match_to upper-cases every read,
    adapters/__init__.py:349): ASCII a-z -> A-Z on a uint8 tensor."""
    lower = (ascii_2d >= 97) & (ascii_2d <= 122)
    return torch.where(lower, ascii_2d - 32, ascii_2d)


def best_adapter_records(adapters, ascii_upper, lens=None):
    """Device twin of ``AdapterCutter._best_match`` (commands/trim/modifiers.py:107-122) for
    plain (non-linked) adapters: the records of the adapter with the most matches per read
    (the first one wins ties) and its index (-1: no adapter matched)."""
    best = None
    which = None
    for idx, adapter in enumerate(adapters):
        rec = adapter.match_records(ascii_upper, lens)
        if best is None:
            best = rec
            which = torch.where(rec[:, 1] >= 0, torch.zeros_like(rec[:, 1], dtype=torch.int32),
                                torch.full_like(rec[:, 1], -1, dtype=torch.int32))
            continue
        better = (rec[:, 1] >= 0) & ((best[:, 1] < 0) | (rec[:, 4] > best[:, 4]))
        best = torch.where(better[:, None], rec, best)
        which = torch.where(better, torch.full_like(which, idx), which)
    return best, which


def linked_records_from_source(linked_adapters, batch, begin, end, active=None):
    """``LinkedSet.match_source`` on the kept intervals of a FastqBatch (``atropos_amd.fastq``)."""
    from .fastq import RecordSource
    return _linked_set(linked_adapters).match_source(RecordSource(batch, begin, end), active)


def _linked_records_stepwise(linked_adapters, source, active=None):
    """Sets outside the fused pipeline's envelope: every 5' adapter against all reads, then -- per
    adapter -- its 3' part against the slices ``read[front.rstop:]`` of the reads it claimed
    (``source.sliced``).  Same return value as ``LinkedSet.match_source``."""
    n = source.n
    front = back = None
    which = count = None
    for k, la in enumerate(linked_adapters):
        f = la.front_adapter.match_source(source)
        if front is None:
            dev = f.device
            none = torch.zeros((n, 8), dtype=torch.int16, device=dev)
            none[:, 1] = -1
            front, back = none.clone(), none.clone()
            which = torch.full((n,), -1, dtype=torch.int32, device=dev)
            count = torch.zeros((n,), dtype=torch.int32, device=dev)
        has = f[:, 1] >= 0
        if active is not None:
            has &= active != 0
        count += has.to(torch.int32)
        first = has & (which < 0)                                      # the first matching adapter claims the read
        front = torch.where(first[:, None], f, front)
        which = torch.where(first, torch.full_like(which, k), which)
    starts = torch.where(which >= 0, front[:, 3].to(torch.int32), torch.zeros_like(which))
    rest = source.sliced(starts)
    for k, la in enumerate(linked_adapters):
        mine = which == k
        if bool(mine.any()):
            b = la.back_adapter.match_source(rest)
            back = torch.where(mine[:, None], b, back)
    return which, count, front, back
