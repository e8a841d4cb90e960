"""A minimal read record for the per-read (batch-of-one) entry points and the tests.

The batched pipelines never build one of these (they work on device-resident FASTQ chunks,
``atropos_amd.fastq``); the boundary objects accept ANY object with ``name``, ``sequence``,
``qualities`` and slicing -- the reference's own ``Sequence`` (atropos/io/_seqio.pyx:7-161)
included.  FASTQ/FASTA parsing into such objects is out of scope.
"""
# bookkeeping the trimming modifiers attach to a read, with the value a fresh read starts with
_STATE = (("match", None), ("match_info", None), ("insert_overlap", False), ("merged", False), ("corrected", 0))


class Read(object):
    """name, sequence, qualities (``None`` when there are none) + the modifiers' bookkeeping.
    ``read[a:b]`` is the same read with sequence and qualities sliced."""

    __slots__ = ("name", "sequence", "qualities", "name2", "original_length", "clipped") + tuple(k for k, _ in _STATE)

    def __init__(self, name, sequence, qualities=None, name2="", original_length=None, clipped=None, **state):
        if qualities is not None and len(qualities) != len(sequence):
            raise ValueError("read %r: %d bases but %d quality values" % (name, len(sequence), len(qualities)))
        self.name, self.name2 = name, name2
        self.sequence, self.qualities = sequence, qualities
        self.original_length = len(sequence) if not original_length else original_length
        self.clipped = list(clipped) if clipped else [0, 0, 0, 0]
        for key, fresh in _STATE:
            setattr(self, key, state.pop(key, fresh))
        if state:
            raise TypeError("unexpected arguments: %s" % ", ".join(sorted(state)))

    def __len__(self):
        bases = self.sequence
        return len(bases)

    def __getitem__(self, key):
        part = Read.__new__(type(self))
        for slot in Read.__slots__:
            setattr(part, slot, getattr(self, slot))
        part.clipped = list(self.clipped)
        part.sequence = self.sequence[key]
        if self.qualities is not None:
            part.qualities = self.qualities[key]
        return part

    def _key(self):
        return tuple(getattr(self, slot) for slot in ("name", "sequence", "qualities"))

    def __eq__(self, other):
        return self._key() == Read._key(other)

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __repr__(self):
        shown = [repr(self.name), repr(self.sequence)] + ([] if self.qualities is None else [repr(self.qualities)])
        return "Read(%s)" % ", ".join(shown)

    def __reduce__(self):
        return (Read, self._key() + (self.name2,))


Sequence = Read          # the reference's name for it
