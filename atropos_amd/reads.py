"""The read record the boundary objects operate on: same attributes and slicing
behaviour as the reference's ``Sequence`` (atropos/io/_seqio.pyx:7-161).  FASTQ/FASTA
parsing is out of scope; any object with these attributes
(including the reference's own ``Sequence``) can be handed to the adapters and cutters."""


class Sequence(object):
    """A FASTQ/FASTA record: name, sequence, qualities (``None`` for FASTA) plus the
    bookkeeping slots the trimming modifiers fill in (match, match_info, clipped,
    insert_overlap, merged, corrected)."""

    __slots__ = ("name", "sequence", "qualities", "name2", "original_length", "match", "match_info", "clipped",
                 "insert_overlap", "merged", "corrected")

    def __init__(self, name, sequence, qualities=None, name2='', original_length=None, match=None, match_info=None,
                 clipped=None, insert_overlap=False, merged=False, corrected=0):
        if qualities is not None and len(qualities) != len(sequence):
            raise ValueError(
                "In read named {0!r}: length of quality sequence ({1}) and length of read ({2}) do not "
                "match".format(name, len(qualities), len(sequence)))
        self.name = name
        self.sequence = sequence
        self.qualities = qualities
        self.name2 = name2
        self.original_length = original_length or len(sequence)
        self.match = match
        self.match_info = match_info
        self.clipped = clipped or [0, 0, 0, 0]
        self.insert_overlap = insert_overlap
        self.merged = merged
        self.corrected = corrected

    def __getitem__(self, key):
        return self.__class__(
            self.name, self.sequence[key], self.qualities[key] if self.qualities is not None else None, self.name2,
            self.original_length, self.match, self.match_info, list(self.clipped), self.insert_overlap, self.merged,
            self.corrected)

    def __len__(self):
        return len(self.sequence)

    def __eq__(self, other):
        return (self.name == other.name and self.sequence == other.sequence and self.qualities == other.qualities)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        q = '' if self.qualities is None else ', qualities={0!r}'.format(self.qualities)
        return '<Sequence(name={0!r}, sequence={1!r}{2})>'.format(self.name, self.sequence, q)

    def __reduce__(self):
        return (Sequence, (self.name, self.sequence, self.qualities, self.name2))
