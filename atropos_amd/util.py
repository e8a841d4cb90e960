"""Host-side helpers on the alignment path (reference atropos/util/__init__.py):
``RandomMatchProbability`` (:104-174) and ``reverse_complement`` (:67-88, :479-482).

These run on the host by design: the probabilities are table entries the host
precomputes -- with the reference's exact expression order, because they are
compared against thresholds in double precision -- and hands to the device."""


def build_iso_nucleotide_table():
    """ISO nucleotide -> complement, upper and lower case."""
    nuc = {'A': 'T', 'C': 'G', 'R': 'Y', 'S': 'S', 'W': 'W', 'K': 'M', 'B': 'V', 'D': 'H', 'N': 'N'}
    for base, comp in tuple(nuc.items()):
        nuc[comp] = base
        nuc[base.lower()] = comp.lower()
        nuc[comp.lower()] = base.lower()
    return nuc


BASE_COMPLEMENTS = build_iso_nucleotide_table()
IUPAC_BASES = frozenset(('X',) + tuple(BASE_COMPLEMENTS.keys()))
GC_BASES = frozenset('CGRYSKMBDHVN')


def complement(seq):
    return "".join(BASE_COMPLEMENTS[base] for base in seq)


def reverse_complement(seq):
    """Reverse complement; ``KeyError`` for a character without a complement."""
    return "".join(BASE_COMPLEMENTS[base] for base in reversed(seq))


class RandomMatchProbability(object):
    """Random-match probability of ``matches`` out of ``size`` bases by the binomial
    tail, with a cache of big-integer factorials (reference util/__init__.py:104-174).
    The summation order, the true-division/floor-division switch and the caching
    rule (zero probabilities are recomputed) follow the reference so that every
    value is bit-identical."""

    def __init__(self, init_size=150):
        self.cache = {}
        self.factorials = [1] * init_size
        self.max_n = 1
        self.cur_array_size = init_size

    def __call__(self, matches, size, match_prob=0.25, mismatch_prob=0.75):
        key = (matches, size, match_prob)
        prob = self.cache.get(key, None)
        if prob:
            return prob
        if matches == size:
            prob = match_prob ** matches
        else:
            nfac = self.factorial(size)
            prob = 0.0
            for i in range(matches, size + 1):
                j = size - i
                try:
                    div = nfac / self.factorial(i) / self.factorial(j)
                except OverflowError:
                    div = nfac // self.factorial(i) // self.factorial(j)
                prob += (mismatch_prob ** j) * (match_prob ** i) * div
        self.cache[key] = prob
        return prob

    def factorial(self, num):
        if num > self.max_n:
            self._fill_upto(num)
        return self.factorials[num]

    def _fill_upto(self, num):
        if num >= self.cur_array_size:
            extension_size = num - self.cur_array_size + 1
            self.factorials += [1] * extension_size
        idx = self.max_n
        next_i = idx + 1
        while idx < num:
            self.factorials[next_i] = next_i * self.factorials[idx]
            idx = next_i
            next_i += 1
        self.max_n = idx


def rmp_table(match_probability, max_size, **probs):
    """``table[size, matches] = match_probability(matches, size, **probs)`` for
    ``0 <= matches <= size <= max_size`` as a float64 array (entries above the diagonal
    are 0) -- the table the device insert aligner looks probabilities up in.

    For a ``RandomMatchProbability`` the sums are evaluated for all ``matches`` of one
    ``size`` at once with numpy, in the same left-to-right order as ``__call__`` (term_k
    + term_k+1 + ...), so every entry is bit-identical to the per-call value; any other
    callable is simply called entry by entry."""
    import numpy as np
    ld = max_size + 1
    table = np.zeros((ld, ld), dtype=np.float64)
    if not isinstance(match_probability, RandomMatchProbability):
        for size in range(ld):
            for k in range(size + 1):
                table[size, k] = match_probability(k, size, **probs)
        return table
    match_prob = probs.get("match_prob", 0.25)
    mismatch_prob = probs.get("mismatch_prob", 0.75)
    fac = match_probability.factorial
    for size in range(ld):
        nfac = fac(size)
        terms = np.empty(size + 1, dtype=np.float64)
        for i in range(size + 1):
            j = size - i
            try:
                div = nfac / fac(i) / fac(j)
            except OverflowError:
                div = nfac // fac(i) // fac(j)
            terms[i] = (mismatch_prob ** j) * (match_prob ** i) * div
        acc = 0.0 + terms
        for d in range(1, size + 1):
            acc[:size + 1 - d] += terms[d:]
        acc[size] = match_prob ** size
        table[size, :size + 1] = acc
    return table
