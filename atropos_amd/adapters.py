# coding: utf-8
"""Adapters: the boundary objects that call the aligner.

Same public names, constructor arguments, attributes and results as the reference's
``atropos/adapters/__init__.py`` (ADAPTER_TYPES :41-73, AdapterParser :80-229, Adapter :231-505,
LinkedMatch / LinkedAdapter :612-745, parse_braces :933-970), so that callers written against it
keep working -- but organised around BATCHES: an adapter's real entry points are
``match_to_batch`` (read objects in, Match objects out) and ``match_source`` (device-resident
reads in, result records out); ``match_to(read)`` is the batch of one.  ColorspaceAdapter and
AdapterCache are out of scope.
"""
import itertools
import operator
import re
from collections import Counter, defaultdict, namedtuple

import numpy as np
import torch

from . import _lib
from . import align
from .batch import ReadBatch
from .align import Match
from .util import IUPAC_BASES, GC_BASES, rmp_table

START_WITHIN_SEQ1, START_WITHIN_SEQ2 = align.START_WITHIN_SEQ1, align.START_WITHIN_SEQ2
STOP_WITHIN_SEQ1, STOP_WITHIN_SEQ2 = align.STOP_WITHIN_SEQ1, align.STOP_WITHIN_SEQ2

# where an adapter may sit = which ends of the alignment are free (reference :41-73)
BACK = START_WITHIN_SEQ2 | STOP_WITHIN_SEQ2 | STOP_WITHIN_SEQ1          # 14  regular 3'
FRONT = START_WITHIN_SEQ2 | STOP_WITHIN_SEQ2 | START_WITHIN_SEQ1        # 11  regular 5'
PREFIX = STOP_WITHIN_SEQ2                                               # 8   anchored 5'
SUFFIX = START_WITHIN_SEQ2                                              # 2   anchored 3'
ANYWHERE = align.SEMIGLOBAL                                             # 15  variable 5'/3'
LINKED = 'linked'

_AdapterTypeBase = namedtuple("AdapterType", "name desc flags")


class AdapterType(_AdapterTypeBase):
    """(name, description, alignment flags) of one adapter placement."""
    __slots__ = ()

    def asdict(self):
        return dict(self._asdict())


ADAPTER_TYPES = {t.name: t for t in (
    AdapterType('back', "regular 3'", BACK), AdapterType('front', "regular 5'", FRONT),
    AdapterType('prefix', "anchored 5'", PREFIX), AdapterType('suffix', "anchored 3'", SUFFIX),
    AdapterType('anywhere', "variable 5'/3'", ANYWHERE), AdapterType('linked', 'linked', LINKED))}


def where_int_to_dict(where):
    found = [t for t in ADAPTER_TYPES.values() if t.flags == where]
    if not found:
        raise ValueError("Invalid WHERE value: {}".format(where))
    return found[0].asdict()


_adapter_numbers = itertools.count(1)


_BRACE_TOKEN = re.compile(r'(?P<rep>[^{}])\{(?P<count>[^{}]*)\}|(?P<plain>[^{}]+?)(?=[^{}]\{|[{}]|$)|(?P<stray>[{}])')


def parse_braces(sequence):
    """Expand the repeat notation of adapter specifications: ``x{n}`` stands for n copies of the single
    character x (``TGA{5}CT`` -> ``TGAAAAACT``, ``A{0}`` -> nothing).  n is whatever ``int()`` accepts,
    0 <= n <= 10000.  A brace that does not close such a group right behind a character -- a leading
    ``{``, ``{}`` chained behind another group, a missing ``}`` -- is a ``ValueError``
    (behaviour of the reference's parser, adapters/__init__.py:933-970)."""
    pieces = []
    for tok in _BRACE_TOKEN.finditer(sequence):
        if tok.group('stray') is not None:
            raise ValueError('unbalanced or misplaced "{}" at offset {} of {!r}'.format(
                tok.group('stray'), tok.start(), sequence))
        if tok.group('plain') is not None:
            pieces.append(tok.group('plain'))
            continue
        count = int(tok.group('count'))            # ValueError for anything that is not a number
        if count < 0 or count > 10000:
            raise ValueError('repeat count {} outside 0 .. 10000'.format(count))
        pieces.append(tok.group('rep') * count)
    return ''.join(pieces)


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
        if not sequence:
            raise ValueError("Empty adapter sequence")
        if where not in (BACK, FRONT, PREFIX, SUFFIX, ANYWHERE):
            raise ValueError("Invalid WHERE value: {}".format(where))
        sequence = parse_braces(sequence.upper().replace('U', 'T'))
        letters = set(sequence)
        # wildcards only make sense when the adapter has some (reference :268-270)
        adapter_wildcards = bool(adapter_wildcards) and not letters <= set('ACGT')
        if adapter_wildcards and letters - IUPAC_BASES:
            raise ValueError("Invalid character(s) in adapter sequence: {}".format(','.join(letters - IUPAC_BASES)))
        if alphabet is not None:
            for character in sequence:
                if character not in alphabet:
                    raise ValueError("Character {!r} is not in the alphabet".format(character))
        self.sequence, self.where = sequence, where
        self.name = str(next(_adapter_numbers)) if name is None else name
        self.max_error_rate = max_error_rate
        self.min_overlap = min(min_overlap, len(sequence))
        self.read_wildcards, self.adapter_wildcards = read_wildcards, adapter_wildcards
        self.indels = indels
        self.match_probability, self.max_rmp, self.gc_content = match_probability, max_rmp, gc_content
        self.debug = False
        # which end of the read a match removes: True 5', False 3', None decided per match (ANYWHERE)
        self._front_flag = {BACK: False, SUFFIX: False, ANYWHERE: None}.get(where, True)
        # statistics about the removed sequences: length -> count, length -> errors -> count
        self.lengths_front, self.lengths_back = Counter(), Counter()
        self.errors_front, self.errors_back = defaultdict(Counter), defaultdict(Counter)
        self.adjacent_bases = dict.fromkeys(('A', 'C', 'G', 'T', ''), 0)
        self.aligner = align.Aligner(sequence, max_error_rate, flags=where, wildcard_ref=adapter_wildcards,
                                     wildcard_query=read_wildcards, min_overlap=self.min_overlap)
        # an adapter without indels simply prices them out of every alignment (reference :316-322)
        self._indel_cost = indel_cost if indels else 100000
        self.aligner.indel_cost = self._indel_cost
        # an anchored adapter without indels is a plain prefix / suffix comparison (reference :370-380)
        self._plain_compare = not indels and where in (PREFIX, SUFFIX)
        self._exact_aligner = None
        self._rmp_cache = None
        self._rmp_device = None

    def __repr__(self):
        shown = ("name", "sequence", "where", "max_error_rate", "min_overlap", "read_wildcards", "adapter_wildcards", "indels")
        return "<Adapter(%s)>" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in shown)

    def __len__(self):
        return len(self.sequence)

    def enable_debug(self):
        self.aligner.enable_debug()
        self.debug = True

    # ------------------------------------------------------------------ matching
    def match_to(self, read):
        """Attempt to match this adapter to the given read; returns a Match, or None if the
        criteria (minimum overlap, maximum error rate, random-match probability) are not met.
        The rules of ``match_to_batch`` on one alignment (``Aligner.locate``: one library call); the rarer modes -- plain
        prefix / suffix compare, read wildcards with a literal adapter -- are a batch of one."""
        if self._plain_compare or (not self.adapter_wildcards and self.read_wildcards):
            return self.match_to_batch([read])[0]
        hit = self.aligner.locate(_seq_of(read).upper())
        if hit is None:
            return None
        astart, astop, rstart, rstop, matches, errors = hit
        m, size = len(self.sequence), astop - astart
        ok = size >= self.min_overlap and size > 0 and errors / size <= self.max_error_rate       # :386-398
        if ok and self.max_rmp is not None:
            ok = bool(self._rmp_by_size()[min(max(size, 0), m), min(max(matches, 0), m)] <= self.max_rmp)
        if not ok and not self.adapter_wildcards and matches == m and errors == 0:
            ok = True                                 # the literal shortcut: a full-length exact occurrence bypasses the filters (:351-367)
        if not ok:
            return None
        return Match(astart, astop, rstart, rstop, matches, errors, self._front_flag, self, None if isinstance(read, str) else read)

    def _rmp_by_size(self):
        if self._rmp_cache is None:
            self._rmp_cache = rmp_table(self.match_probability, len(self.sequence))
        return self._rmp_cache

    def locate_records(self, reads_upper):
        """Raw device step of the batched match: the alignment records (int16 [n, 8]
        numpy) for already upper-cased reads (list of str or uint8 [n, width] tensor)
        plus, when the literal exact-match shortcut cannot be read off those records, the
        records of the literal search."""
        if self._plain_compare:
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
                if self._plain_compare:
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
        table)`` (the reads 4-bit packed with that table)."""
        be = _lib.get_backend()
        m = len(self.sequence)
        split = getattr(source, "split_long", None)
        if split is not None and not self._plain_compare and source.max_len() > _lib.MAX_READ_LEN:
            # records beyond the batch pipelines' read length: matched as a batch of their own (long-read sweep)
            short, long_idx, long_source = split()
            rec = self.match_source(short)
            rec[long_idx] = self.match_source(long_source)
            return rec
        if self._plain_compare:
            rec = self.aligner.compare_batch(source.batch(self.aligner.table_kind, self.aligner._table),
                                             suffix=(self.where == SUFFIX))
        else:
            batch_for = getattr(source, "batch_for", None)      # (a source that can pack for the aligner's fastest path)
            rec = self.aligner.locate_batch(batch_for(self.aligner) if batch_for else
                                            source.batch(self.aligner.table_kind, self.aligner._table)).records
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
    def trimmed(self, match):
        """The read of ``match`` with the adapter (and what lies beyond it) removed; updates the
        length / error statistics of this adapter."""
        read, front = match.read, (match.front if self._front_flag is None else self._front_flag)
        if front:
            removed = match.rstop
            self.lengths_front[removed] += 1
            self.errors_front[removed][match.errors] += 1
            return read[removed:]
        removed = len(read) - match.rstart
        self.lengths_back[removed] += 1
        self.errors_back[removed][match.errors] += 1
        before = read.sequence[match.rstart - 1:match.rstart]
        self.adjacent_bases[before if before in ('A', 'C', 'G', 'T') else ''] += 1
        return read[:match.rstart]

    # ------------------------------------------------------------------ reporting
    def random_match_probabilities(self):
        """probabilities[i]: chance that the first i adapter bases (the last i for a 5' adapter)
        match a random sequence with this GC content, indels not taken into account."""
        gc, at = self.gc_content / 2.0, (1 - self.gc_content) / 2.0
        strong = frozenset(GC_BASES if self.adapter_wildcards else 'GC')
        order = reversed(self.sequence) if self._front_flag else self.sequence
        return list(itertools.accumulate([1.0] + [gc if base in strong else at for base in order], operator.mul))

    def summarize(self):
        where, front_side, back_side = self.where, (ANYWHERE, FRONT, PREFIX), (ANYWHERE, BACK, SUFFIX)
        n_front, n_back = sum(self.lengths_front.values()), sum(self.lengths_back.values())
        assert (n_front == 0 or where in front_side) and (n_back == 0 or where in back_side)
        stats = {"adapter_class": type(self).__name__, "where": where_int_to_dict(where), "sequence": self.sequence,
                 "max_error_rate": self.max_error_rate, "match_probabilities": self.random_match_probabilities(),
                 "total_front": n_front, "total_back": n_back, "total": n_front + n_back}
        for side, sides, lengths, errors in (("front", front_side, self.lengths_front, self.errors_front),
                                            ("back", back_side, self.lengths_back, self.errors_back)):
            if where in sides:
                stats["lengths_" + side] = dict(lengths)
                stats["errors_" + side] = {length: dict(by_errors) for length, by_errors in errors.items()}
        if where in (BACK, SUFFIX):
            stats["adjacent_bases"] = dict(self.adjacent_bases)
        return stats


class LinkedMatch(object):
    """Result of ``LinkedAdapter.match_to``: the 5' match and the 3' match (or None) found in
    what the 5' match left of the read."""
    __slots__ = ("front_match", "back_match", "adapter")

    def __init__(self, front_match, back_match, adapter):
        if front_match is None:
            raise ValueError("a LinkedMatch needs a 5' match")
        self.front_match, self.back_match, self.adapter = front_match, back_match, adapter

    def get_info_record(self):
        return (self.back_match or self.front_match).get_info_record()


class LinkedAdapter(object):
    """``^FRONT...BACK``: an anchored 5' adapter and a regular 3' adapter; the 3' adapter is only
    looked for in reads that start with the 5' adapter, in ``read[front_match.rstop:]``."""

    def __init__(self, front_sequence, back_sequence, front_anchored=True, back_anchored=False, name=None, **kwargs):
        if not front_anchored or back_anchored:
            raise AssertionError("linked adapters are an anchored 5' part plus a regular 3' part")
        self.front_anchored, self.back_anchored = front_anchored, back_anchored
        self.where = LINKED
        self.name = str(next(_adapter_numbers)) if name is None else name
        self.front_adapter = Adapter(front_sequence, where=PREFIX, name=None, **kwargs)
        self.back_adapter = Adapter(back_sequence, where=BACK, name=None, **kwargs)

    def enable_debug(self):
        for part in (self.front_adapter, self.back_adapter):
            part.enable_debug()

    def match_to(self, read):
        return self.match_to_batch([read])[0]

    def match_to_batch(self, reads):
        """LinkedMatch / None per read: the 5' adapter over the whole batch, then the 3' adapter over
        the remainders of the reads that had a 5' match."""
        fronts = self.front_adapter.match_to_batch(reads)
        claimed = [i for i, fm in enumerate(fronts) if fm is not None]
        backs = self.back_adapter.match_to_batch([reads[i][fronts[i].rstop:] for i in claimed]) if claimed else []
        out = [None] * len(reads)
        for i, bm in zip(claimed, backs):
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
        rest = self.front_adapter.trimmed(match.front_match)
        return self.back_adapter.trimmed(match.back_match) if match.back_match else rest

    def summarize(self):
        out = {"where": where_int_to_dict(self.where)}
        for side, part in (("front", self.front_adapter), ("back", self.back_adapter)):
            out[side + "_sequence"] = part.sequence
            out[side + "_max_error_rate"] = part.max_error_rate
            out[side + "_match_probabilities"] = part.random_match_probabilities()
            out[side + "_lengths_front"] = dict(part.lengths_front)
            out[side + "_lengths_back"] = dict(part.lengths_back)
        out["total_front"] = sum(self.front_adapter.lengths_front.values())
        out["total_back"] = sum(self.back_adapter.lengths_back.values())
        out["total"] = out["total_front"] + out["total_back"]
        return out


# name=  ^anchored5'  body  anchored3'$ ; the body of a linked adapter is FRONT...BACK
_SPEC = re.compile(r"^(?:(?P<name>[^=]*)=)?\s*(?P<hat>\^)?(?P<body>.*?)(?P<dollar>\$)?\s*$", re.S)


def fasta_records(path):
    """(header, sequence) of every record of a FASTA file as the adapter parser needs them: lines are
    stripped (DOS line ends too), blank lines and ``#`` comment lines skipped, wrapped sequences joined; text
    before the first ``>`` is an error (reference: FastaReader.__iter__, io/seqio.py:251-280).  Plain text or
    .gz / .bz2 / .xz (what the reference's xopen dispatches on)."""
    name = str(path)
    if name.endswith(".gz"):
        import gzip
        opener = gzip.open
    elif name.endswith(".bz2"):
        import bz2
        opener = bz2.open
    elif name.endswith(".xz"):
        import lzma
        opener = lzma.open
    else:
        opener = open
    header, parts = None, []
    with opener(path, "rt") as handle:
        for number, raw in enumerate(handle, 1):
            line = raw.strip()
            if not line or (line[0] == '#'):
                continue
            if line[0] == '>':
                if header is not None:
                    yield header, "".join(parts)
                header, parts = line[1:], []
            elif header is None:
                from .fastq import FormatError
                shown = line if len(line) <= 100 else line[:97] + "..."       # util.truncate_string
                raise FormatError("At line {0}: Expected '>' at beginning of FASTA record, but got {1!r}.".format(number, shown))
            else:
                parts.append(line)
    if header is not None:
        yield header, "".join(parts)


class AdapterParser(object):
    """Factory for Adapter objects that all share the same parameters (error rate, indels
    ...); ``**kwargs`` go to the Adapter constructors.  Supports the command-line
    notation: ``name=SEQ``, ``^SEQ`` (anchored 5'), ``SEQ$`` (anchored 3'),
    ``SEQ1...SEQ2`` (linked), and ``file:PATH`` -- one adapter per record of a FASTA file, named by the
    first word of its header.  (Colorspace and the adapter cache are out of scope.)"""

    def __init__(self, **kwargs):
        self.constructor_args = kwargs

    def parse(self, spec, cmdline_type='back'):
        if spec.startswith('file:'):                  # adapters/__init__.py:113-119
            return (self.parse_from_spec(sequence, cmdline_type, header.split(None, 1)[0])
                    for header, sequence in fasta_records(spec[5:]))
        return iter([self.parse_from_spec(spec, cmdline_type)])

    def parse_from_spec(self, spec, cmdline_type='back', name=None):
        """One adapter from its command-line form; cmdline_type: the option it came with
        ('back' -a, 'front' -g, 'anywhere' -b)."""
        if cmdline_type not in ADAPTER_TYPES:
            raise ValueError('cmdline_type cannot be {0!r}'.format(cmdline_type))
        if spec is None:
            raise ValueError('Either name or spec must be given')
        parts = _SPEC.match(spec)
        if name is None and parts.group("name") is not None:
            name = parts.group("name").strip()
        elif name is not None:
            parts = _SPEC.match("=" + spec)                    # a given name: '=' is not a separator
        anchor5, anchor3, body = bool(parts.group("hat")), bool(parts.group("dollar")), parts.group("body")
        make = lambda seq, where: Adapter(sequence=seq, where=where, name=name, **self.constructor_args)
        if cmdline_type == 'anywhere':
            if anchor5 or anchor3:
                raise ValueError("'anywhere' (-b) adapters may not be anchored")
            if '...' in body:
                raise ValueError("'anywhere' (-b) adapters may not be linked")
            return make(body, ANYWHERE)
        three_prime = cmdline_type == 'back'
        first, dots, second = body.partition('...')
        if dots:
            if first and second:
                # -a FRONT...BACK anchors the 5' part by itself
                return LinkedAdapter(first, second, name=name, front_anchored=anchor5 or three_prime,
                                     back_anchored=anchor3, **self.constructor_args)
            if three_prime and second:          # -a ...ADAPTER: a plain 3' adapter
                body = second
            elif three_prime:                   # -a ADAPTER...: an anchored 5' adapter
                body, three_prime, anchor5 = first, False, True
            elif first:                         # -g ADAPTER...: a plain 5' adapter
                body = first
            else:                               # -g ...ADAPTER
                raise ValueError('Invalid adapter specification')
        if anchor5 and anchor3:
            raise ValueError('Trying to use both "^" and "$" in adapter specification {!r}'.format(spec))
        if anchor5 and three_prime:
            raise ValueError("Cannot anchor the 3' adapter at its 5' end")
        if anchor3 and not three_prime:
            raise ValueError("Cannot anchor 5' adapter at 3' end")
        where = (SUFFIX if anchor3 else BACK) if three_prime else (PREFIX if anchor5 else FRONT)
        return make(body, where)

    def parse_multi(self, back=None, anywhere=None, front=None):
        """All adapters of a command line, in the order -a, -b, -g."""
        by_option = (('back', back), ('anywhere', anywhere), ('front', front))
        return [adapter for option, specs in by_option for spec in (specs or ()) for adapter in self.parse(spec, option)]


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

    def group_applies(self, max_len):
        """Can ``pack_groups`` / ``match_groups`` take this set for reads of at most max_len bases?  (Every 3' aligner
        inside the two-pass pre-pass's envelope; the HIP backend.)"""
        be = self._backend
        return self._handle is not None and hasattr(be, "linked_group_applies") and be.linked_group_applies(self._handle, max_len)

    def pack_groups(self, ascii_upper, lens=None, max_len=None):
        """The pack-time half of LinkedAdapter.match_to for a batch (atr_linked_group_pack): every read's 5' part
        decided from its ASCII row (adapters/__init__.py:671-676), ``read[front.rstop:]`` packed as bit planes into the
        sub-batch of the adapter that matched.  Returns a ``_lib.LinkedGroups``."""
        max_len = ascii_upper.shape[1] if max_len is None else max_len
        return self._backend.linked_group_pack(self._handle, ascii_upper, lens, max_len, self.table)

    def match_groups(self, groups, ordered=True):
        """The 3' parts of a packed batch (atr_linked_group_match): (which int32, count int32, front, back) as
        ``match_source`` returns them when ``ordered``; else (slab, groups) -- raw 3' records in slot order."""
        slab, back = self._backend.linked_group_match(self._handle, groups, ordered)
        if not ordered:
            return slab, groups
        return groups.which[:, 0].to(torch.int32), groups.which[:, 1].to(torch.int32), groups.front, back

    def match_source(self, source, active=None):
        """(which, count, front, back) for a read source (``AsciiSource`` / ``RecordSource``):
        int32 index of the first linked adapter whose 5' part matches (-1: none), the number of
        5' parts that match, and the two int16 [n, 8] record tensors; the 3' records refer to
        ``read[front.rstop:]`` and are None-records where no 5' part matched."""
        if self._handle is None:
            return _linked_records_stepwise(self.adapters, source, active)
        batch = source.batch(self.table_kind, self.table)
        if batch.max_len == 0 and batch.nreads:
            # every read is empty (trimmed away by an earlier stage): nothing matches (LinkedAdapter.match_to needs the
            # 5' part, adapters/__init__.py:671-690); the device call has no packed chunk to read
            be = self._backend
            none = be.empty((batch.nreads, 8), torch.int16)
            none.zero_()
            none[:, 1] = -1
            which = torch.full((batch.nreads,), -1, dtype=torch.int32, device=none.device)
            return which, torch.zeros_like(which), none, none.clone()
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
