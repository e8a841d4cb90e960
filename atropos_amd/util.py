"""Host-side helpers on the alignment path: nucleotide complements and the random-match
probability tables the device kernels look thresholds up in.

Counterparts in the reference: ``atropos/util/__init__.py`` -- the complement table (:67-88),
``reverse_complement`` (:479-482) and ``RandomMatchProbability`` (:104-174).  The probabilities
are compared against thresholds in double precision, so the one thing taken over from the
reference is the ORDER OF FLOATING-POINT OPERATIONS of its binomial tail (:141-152): bigint
factorials, ``n! / i! / j!`` as two true divisions (two floor divisions once a factorial no
longer fits a double), terms ``q**j * p**i * coefficient`` summed left to right from
``i = matches``.  Everything around that expression is this repo's own: whole rows of the table
are evaluated at once and the callable is a view on those rows.
"""
import functools
import math

import numpy as np

# complement pairs of the IUPAC nucleotide codes; S, W and N are their own complements
_PAIRS = ("AT", "CG", "RY", "KM", "BV", "DH", "SS", "WW", "NN")
BASE_COMPLEMENTS = {}
for _a, _b in _PAIRS:
    for _x, _y in ((_a, _b), (_b, _a)):
        BASE_COMPLEMENTS[_x] = _y
        BASE_COMPLEMENTS[_x.lower()] = _y.lower()
del _a, _b, _x, _y
IUPAC_BASES = frozenset(BASE_COMPLEMENTS) | {"X"}
GC_BASES = frozenset("CGRYSKMBDHVN")

_COMPLEMENT_MAP = str.maketrans(BASE_COMPLEMENTS)


def _require_bases(seq):
    """``KeyError`` (as a dict lookup would raise) for the first character without a complement."""
    for ch in seq:
        if ch not in BASE_COMPLEMENTS:
            raise KeyError(ch)


def complement(seq):
    _require_bases(seq)
    return seq.translate(_COMPLEMENT_MAP)


def reverse_complement(seq):
    """Reverse complement; ``KeyError`` for a character without a complement."""
    _require_bases(seq)
    return seq[::-1].translate(_COMPLEMENT_MAP)


@functools.lru_cache(maxsize=None)
def _factorial(n):
    return math.factorial(n)


def _coefficient(size, i):
    """size! / i! / (size - i)! with the reference's arithmetic (see the module docstring)."""
    top, left, right = _factorial(size), _factorial(i), _factorial(size - i)
    try:
        return top / left / right
    except OverflowError:
        return top // left // right


def tail_row(size, match_prob=0.25, mismatch_prob=0.75):
    """float64 array ``row[k]`` = probability of at least ``k`` matches among ``size`` random
    bases, ``k = 0 .. size``, every entry bit-identical to the reference's per-call value: the terms
    are summed left to right starting at ``k`` (all ``k`` at once: after step ``d`` entry ``k``
    holds ``term[k] + ... + term[k + d]``), and ``k == size`` is the plain power."""
    terms = np.array([(mismatch_prob ** (size - i)) * (match_prob ** i) * _coefficient(size, i) for i in range(size + 1)],
                     dtype=np.float64)
    row = 0.0 + terms
    for d in range(1, size + 1):
        row[:size + 1 - d] += terms[d:]
    row[size] = match_prob ** size
    return row


class RandomMatchProbability(object):
    """``rmp(matches, size, match_prob=0.25, mismatch_prob=0.75)``: probability that ``matches`` or
    more of ``size`` bases match a random sequence.  A view on ``tail_row`` rows, which are kept
    per (size, probabilities)."""

    def __init__(self, init_size=150):
        del init_size                      # accepted for compatibility; rows are built on demand
        self._rows = {}

    def row(self, size, match_prob=0.25, mismatch_prob=0.75):
        key = (size, match_prob, mismatch_prob)
        if key not in self._rows:
            self._rows[key] = tail_row(size, match_prob, mismatch_prob)
        return self._rows[key]

    def __call__(self, matches, size, match_prob=0.25, mismatch_prob=0.75):
        if matches > size:
            return 0.0                     # an empty sum
        if matches < 0:
            matches = 0
        return float(self.row(size, match_prob, mismatch_prob)[matches])


def rmp_table(match_probability, max_size, **probs):
    """``table[size, matches] = match_probability(matches, size, **probs)`` for ``0 <= matches <=
    size <= max_size`` as a float64 array (0 above the diagonal) -- the layout the device kernels
    index.  Rows of a ``RandomMatchProbability`` are taken as they are; any other callable is
    evaluated entry by entry."""
    ld = max_size + 1
    table = np.zeros((ld, ld), dtype=np.float64)
    for size in range(ld):
        if isinstance(match_probability, RandomMatchProbability):
            table[size, :size + 1] = match_probability.row(size, **probs)
        else:
            for k in range(size + 1):
                table[size, k] = match_probability(k, size, **probs)
    return table
