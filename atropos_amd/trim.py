"""Device-resident single-end trimming pipeline: FASTQ bytes in, trimmed FASTQ bytes out.

This is the batched, stage-wise replacement for the reference's per-read loop
``RecordHandler.handle_record`` -> ``Modifiers.modify`` -> ``Filters.filter`` ->
``Formatters.format`` (atropos/commands/trim/__init__.py:122-127, :422-601;
commands/base.py:65-78): every stage is ONE kernel launch over the whole batch.

Stages, in the order the reference's ``op_order`` (default "CGQAW") and the fixed tail of
``trim/__init__.py`` apply them:

  C  UnconditionalCutter            -> atr_clip_batch
  G  NextseqQualityTrimmer          -> atr_quality_trim_batch(nextseq=1)
  Q  QualityTrimmer                 -> atr_quality_trim_batch
  A  AdapterCutter(times, action)   -> atr_pack_records + atr_locate_batch + atr_adapter_postfilter
                                       + atr_match_trim_batch per round
  -  NEndTrimmer (--trim-n)         -> atr_nend_trim_batch
  -  filters (-m, -M, --max-n, --discard-trimmed/-untrimmed) -> atr_read_filter_batch
  -  FastqFormat                    -> atr_fastq_emit

What the pipeline does not do (it raises instead of silently differing): paired-end input,
anchored adapters without indels, modifiers that rewrite names or bases (length tags,
suffix removal, double encoding, zero cap, bisulfite trimmers, merging), info/rest files.
Those stay on the per-read object path (``atropos_amd.modifiers``).
"""
import torch

from . import _lib
from .adapters import LinkedAdapter
from .fastq import FastqBatch, RecordSource, open_by_extension

DEST_MERGED = 6              # MergedReadFilter (filters.py:109-113): installed first, so a merged pair goes nowhere else
DEST_NAMES = {_lib.DEST_KEEP: "keep", _lib.DEST_TOO_SHORT: "too_short", _lib.DEST_TOO_LONG: "too_long",
              _lib.DEST_TOO_MANY_N: "too_many_n", _lib.DEST_TRIMMED: "trimmed", _lib.DEST_UNTRIMMED: "untrimmed"}


def write_mask(batch, begin, end, ubegin, uend):
    """'N' over the masked parts [begin, ubegin) and [uend, end) of every sequence line of the chunk: after it the
    read IS what AdapterCutter's 'mask' action returns (modifiers.py:155-172), for the stages that look at bases."""
    off = batch.records[:, 2].to(torch.int64) & 0xFFFFFFFF
    for lo, hi in ((begin, torch.minimum(ubegin, end)), (torch.maximum(uend, begin), end)):
        lens = (hi - lo).clamp(min=0).to(torch.int64)
        total = int(lens.sum().item())
        if total == 0:
            continue
        first = torch.cumsum(lens, 0) - lens
        idx = torch.repeat_interleave(off + lo.to(torch.int64) - first, lens) + torch.arange(total, device=off.device)
        batch.data[idx] = ord("N")


def mask_before_later_stages(op_order, action):
    """A 'mask' adapter stage followed by a stage that reads bases (NextSeq trimming looks for G's): the N's must be in
    the chunk before that stage runs."""
    return action == "mask" and "A" in op_order and any(op in "GQ" for op in op_order[op_order.index("A") + 1:])


class TrimResult(object):
    """State of a batch after the pipeline: kept interval per read, the destination filter
    and the adapter flag; all device tensors."""

    def __init__(self, batch, begin, end, ubegin, uend, matched, dest, rounds=None, adapters=None, read_batch=None):
        self.batch, self.begin, self.end, self.ubegin, self.uend = batch, begin, end, ubegin, uend
        self.matched, self.dest = matched, dest
        self.rounds, self.adapters = rounds, adapters           # adapter rounds kept for the info / rest / wildcard files
        self.read_batch = read_batch or batch                   # the records as read (batch: after the read-name modifiers)

    def aux_text(self, kinds=("info", "rest", "wildcard")):
        """The lines the reference's InfoFormatter / RestFormatter / WildcardFormatter write for this batch
        (commands/trim/writers.py:193-222; Match.get_info_record / rest / wildcards, align/__init__.py:117-170),
        every read in input order whatever its destination: {kind: bytes}.  Assembled on the host from the result
        arrays of the adapter rounds (the pipeline must have been built with ``aux``)."""
        lines = self.aux_lines(kinds)
        return {k: "".join(line + "\n" for per_read in v for line in per_read).encode("ascii", "replace")
                for k, v in lines.items()}

    def aux_lines(self, kinds=("info", "rest", "wildcard")):
        """{kind: one list of lines per read} (see aux_text; the paired-end result interleaves the two reads' lists)."""
        if self.rounds is None:
            raise ValueError("the pipeline was built without aux=(...): the adapter rounds were not kept")
        n = len(self.batch)
        raw = bytes(self.read_batch.data[:self.read_batch.nbytes].cpu().numpy().tobytes())
        recs = self.read_batch.records.cpu().numpy().astype("int64")
        recs[:, [0, 2, 4]] &= 0xFFFFFFFF                       # (offsets are unsigned 32-bit)
        final = self.batch.records.cpu().numpy().astype("int64")           # (an unmatched read's line carries its final name)
        final[:, [0, 4]] &= 0xFFFFFFFF
        raw_final = raw if self.batch is self.read_batch else bytes(self.batch.data.cpu().numpy().tobytes())
        fb, fe = self.begin.cpu().numpy(), self.end.cpu().numpy()
        rounds = [tuple(t.cpu().numpy() for t in r) for r in self.rounds]
        out = {k: [[] for _ in range(n)] for k in kinds}
        for i in range(n):
            name = raw[recs[i, 0]:recs[i, 0] + recs[i, 1]].decode("ascii", "replace")
            so, qo, has_q = int(recs[i, 2]), int(recs[i, 4]), recs[i, 5] > 0 or recs[i, 3] == 0
            last = None
            lines = []
            for took, rec, which, b0, e0 in rounds:
                if not took[i]:
                    continue
                ad = self.adapters[int(which[i])]
                astart, astop, rstart, rstop, _matches, errors = (int(v) for v in rec[i, :6])
                seq = raw[so + b0[i]:so + e0[i]].decode("ascii", "replace")
                qual = raw[qo + b0[i]:qo + e0[i]].decode("ascii", "replace") if has_q else ""
                last = (ad, astart, astop, rstart, rstop, seq)
                lines.append("\t".join(str(f) for f in (name, errors, rstart, rstop, seq[:rstart], seq[rstart:rstop], seq[rstop:],
                                                        ad.name, qual[:rstart], qual[rstart:rstop], qual[rstop:])))
            # (the name the read has when the files are written, i.e. after the read-name modifiers: the rest and wildcard
            # lines and an unmatched read's info line carry it; a match's info record was made before them)
            fname = raw_final[final[i, 0]:final[i, 0] + final[i, 1]].decode("ascii", "replace")
            if "info" in out:
                if last is None:
                    seq = raw[so + fb[i]:so + max(fb[i], fe[i])].decode("ascii", "replace")
                    # (an unmatched read's line is written from the read as it leaves the pipeline: zero-capped qualities)
                    fqo = int(final[i, 4])
                    qual = raw_final[fqo + fb[i]:fqo + max(fb[i], fe[i])].decode("ascii", "replace") if has_q else ""
                    lines = ["\t".join((fname, "-1", seq, qual))]
                out["info"][i] = lines
            if last is None:
                continue
            ad, astart, astop, rstart, rstop, seq = last
            front = ad._front_flag if ad._front_flag is not None else rstart == 0        # Match._guess_is_front
            if "rest" in out:
                rest = seq[:rstart] if front else seq[rstop:]
                if rest:
                    out["rest"][i].append(rest + " " + fname)
            if "wildcard" in out:
                chars = [seq[rstart + k] for k in range(astop - astart)
                         if ad.sequence[astart + k] == "N" and rstart + k < len(seq)]
                out["wildcard"][i].append("".join(chars) + " " + fname)
        return out

    def counts(self):
        c = torch.bincount(self.dest.to(torch.int64), minlength=6).cpu().tolist()
        return {DEST_NAMES[i]: int(c[i]) for i in range(6)}

    def text(self, which=_lib.DEST_KEEP):
        """Formatted FASTQ text (bytes) of the reads sent to destination ``which``."""
        be = self.batch.backend
        out = be.fastq_emit(self.batch.data, self.batch.records, self.begin, self.end, self.ubegin, self.uend,
                            self.dest, which)
        return bytes(out.cpu().numpy().tobytes())


class TrimPipeline(object):
    """The single-end trimming steps of ``atropos trim`` as whole-batch device stages.

    Args mirror the command line (trim/cli.py): ``adapters`` -- list of
    ``atropos_amd.adapters.Adapter`` / ``LinkedAdapter`` (from ``AdapterParser``);
    ``times`` (-n), ``action`` ('trim' | 'mask' | None); ``cut`` (-u, list of ints);
    ``nextseq_trim``; ``quality_cutoff`` ((front, back), -q); ``quality_base``; ``trim_n``;
    ``minimum_length`` (-m), ``maximum_length`` (-M), ``max_n``; ``discard_trimmed``,
    ``discard_untrimmed``; ``op_order``.
    """

    def __init__(self, adapters=(), times=1, action="trim", cut=(), nextseq_trim=None, quality_cutoff=None,
                 quality_base=33, trim_n=False, minimum_length=None, maximum_length=None, max_n=None,
                 discard_trimmed=False, discard_untrimmed=False, op_order="CGQAW", aux=None, length_tag=None,
                 strip_suffix=(), prefix="", suffix="", zero_cap=False, outputs=None, cut_min=(), bisulfite=None):
        self.adapters = list(adapters)
        # --bisulfite: a list of cutters applied after the op-order stages and before --trim-n (trim/__init__.py:497-516):
        # ("min", front, back, count_trimmed, only_trimmed) = MinCutter, ("nondir", rrbs) = NonDirectionalBisulfiteTrimmer
        self.bisulfite = list(bisulfite or ())
        self._sides = None                                                # per read: [any 5' match, any 3' match] (bisulfite)
        # --cut-min: MinCutter with its defaults (modifiers.py:587-650; after --trim-n, trim/__init__.py:520-524)
        cut_min = list(cut_min or ())
        self.min_front = sum(c for c in cut_min if c > 0)
        self.min_back = -sum(c for c in cut_min if c < 0)
        if (self.min_front or self.min_back) and action != "trim":
            raise NotImplementedError("--cut-min with --mask-adapter / --no-trim (a match counts as trimmed bases that "
                                      "are still there)")
        # --too-short-output / --too-long-output / --untrimmed-output: {"too_short" | "too_long" | "untrimmed": path};
        # the untrimmed file switches the UntrimmedFilter on like --discard-untrimmed (trim/__init__.py:617-630)
        self.outputs = dict(outputs) if outputs else {}
        if "untrimmed" in self.outputs:
            discard_untrimmed = True
        # read-name modifiers and the quality cap, applied after every trimming step (trim/__init__.py:526-541)
        self.length_tag, self.strip_suffix = length_tag or None, list(strip_suffix or ())
        self.prefix, self.suffix, self.zero_cap = prefix or "", suffix or "", bool(zero_cap)
        self._name_mods = bool(self.length_tag or self.strip_suffix or self.prefix or self.suffix)
        self._last_which = None
        self.aux = dict(aux) if aux else None            # {"info" | "rest" | "wildcard": path}: --info-file, --rest-file, --wildcard-file
        self._rounds = None
        self.times, self.action = int(times), action
        if action not in ("trim", "mask", None):
            raise ValueError("action must be 'trim', 'mask' or None")
        cut = list(cut or ())
        self.cut_front = sum(c for c in cut if c > 0)                    # modifiers.py:577-582
        self.cut_back = sum(c for c in cut if c < 0)
        self.nextseq_trim = nextseq_trim
        if quality_cutoff is not None and not isinstance(quality_cutoff, (tuple, list)):
            quality_cutoff = (0, quality_cutoff)                          # "-q 10" = 3' cutoff only (trim/cli.py)
        self.quality_cutoff = quality_cutoff
        self.quality_base = quality_base
        self.trim_n = trim_n
        self.minimum_length = minimum_length
        self.maximum_length = maximum_length
        self.max_n = max_n
        self.discard_trimmed, self.discard_untrimmed = discard_trimmed, discard_untrimmed
        self.op_order = op_order
        linked = [a for a in self.adapters if isinstance(a, LinkedAdapter)]
        if linked and len(linked) != len(self.adapters):
            raise NotImplementedError("mixing linked and plain adapters (the reference's AdapterCutter raises "
                                      "AttributeError as soon as two of them match a read)")
        self._linked = bool(linked)
        if self.aux and self._linked:
            raise NotImplementedError("--info-file / --rest-file / --wildcard-file with linked adapters")
        if self.bisulfite:
            if self._linked:
                raise NotImplementedError("--bisulfite with linked adapters")
            for spec in self.bisulfite:
                if action != "trim" and (spec[0] == "nondir" or spec[3]):
                    raise NotImplementedError("--bisulfite cutters that count the adapters' bases or look at the read's "
                                              "first bases, with --mask-adapter / --no-trim")
        if self._linked and (self.min_front or self.min_back):
            raise NotImplementedError("--cut-min with linked adapters (what a LinkedMatch counts as trimmed is not the interval)")
        if self._linked and ("{name}" in self.prefix or "{name}" in self.suffix):
            raise NotImplementedError("{name} in --prefix / --suffix with linked adapters")

    # ------------------------------------------------------------------ adapter rounds
    @staticmethod
    def _front_code(adapter):
        flag = adapter._front_flag
        return 2 if flag is None else (1 if flag else 0)

    def _round_plain(self, batch, begin, end, active, matched):
        """One ``_best_match`` + ``trimmed`` round (modifiers.py:107-122, :133-139)."""
        be = batch.backend
        source = RecordSource(batch, begin, end)
        best = which = None
        for idx, adapter in enumerate(self.adapters):
            rec = adapter.match_source(source)
            if best is None:
                best = rec
                which = torch.zeros((len(batch),), dtype=torch.int64, device=rec.device)
                continue
            better = (rec[:, 1] >= 0) & ((best[:, 1] < 0) | (rec[:, 4] > best[:, 4]))     # strict >: first wins
            best = torch.where(better[:, None], rec, best)
            which = torch.where(better, torch.full_like(which, idx), which)
        codes = torch.tensor([self._front_code(a) for a in self.adapters], dtype=torch.uint8, device=best.device)
        front = codes[which].contiguous() if len(self.adapters) > 1 else None
        took = (active != 0) & (best[:, 1] >= 0)                          # reads this round's match applies to
        if self._rounds is not None:                                      # (... and the read the match saw)
            self._rounds.append((took, best.clone(), which.clone(), begin.clone(), end.clone()))
        if self._last_which is not None:
            self._last_which = torch.where(took, which, self._last_which)
        if self._sides is not None:                                       # Match.front / _guess_is_front of this round's match
            code = codes[which]
            is_front = torch.where(code == 2, best[:, 2] == 0, code == 1)
            self._sides[0] |= took & is_front
            self._sides[1] |= took & ~is_front
        be.match_trim_batch(best.contiguous(), front, int(codes[0].item()), begin, end, active, matched)

    def _round_linked(self, batch, begin, end, active, matched):
        """LinkedAdapter.match_to + trimmed (adapters/__init__.py:648-706) for linked adapters
        whose 5' parts are mutually exclusive: the first one whose 5' part matches is used."""
        from .adapters import linked_records_from_source
        be = batch.backend
        which, _count, front, back = linked_records_from_source(self.adapters, batch, begin, end, active)
        claimed = (which >= 0).to(torch.uint8)
        be.match_trim_batch(front.contiguous(), None, 1, begin, end, claimed.clone(), None)    # read[front.rstop:]
        be.match_trim_batch(back.contiguous(), None, 0, begin, end, claimed.clone(), None)     # then read[:back.rstart]
        matched |= claimed
        active &= claimed                                                 # no 5' match: the loop over `times` stops

    def _adapter_stage(self, batch, begin, end):
        n = len(batch)
        dev = begin.device
        matched = torch.zeros((n,), dtype=torch.uint8, device=dev)
        if not self.adapters or n == 0:
            return matched, None, None
        before_b, before_e = begin.clone(), end.clone()
        if self.bisulfite:
            self._sides = [torch.zeros((n,), dtype=torch.bool, device=dev), torch.zeros((n,), dtype=torch.bool, device=dev)]
        active = (end > begin).to(torch.uint8)                            # if len(read) == 0: return read
        for _ in range(self.times):
            if self._linked:
                self._round_linked(batch, begin, end, active, matched)
            else:
                self._round_plain(batch, begin, end, active, matched)
        ubegin = uend = None
        if self.action == "mask":                                         # modifiers.py:155-172
            ubegin, uend = begin.clone(), end.clone()
            begin.copy_(before_b)
            end.copy_(before_e)
        elif self.action is None:                                         # :173-174
            begin.copy_(before_b)
            end.copy_(before_e)
        if self.bisulfite:                                                # the interval right after the adapter stage
            self._after = (begin.clone(), end.clone())
        return matched, ubegin, uend

    def _bisulfite_stage(self, batch, begin, end, matched):
        """MinCutter / RRBSTrimmer / NonDirectionalBisulfiteTrimmer (modifiers.py:587-650, :786-831) as interval
        arithmetic.  What a cutter counts as already trimmed: with count_trimmed everything that is gone from that end
        (Sequence.clipped + the matches' rsize_total = the interval); without, what the trimmers removed AFTER the
        adapter stage for a read with a match (clipped[2], clipped[3]) and everything for a read without one."""
        has = matched != 0
        total = batch.seq_lens
        after = getattr(self, "_after", None)
        ab, ae = after if after is not None else (begin, end)
        sides = self._sides if self._sides is not None else [torch.zeros_like(has), torch.zeros_like(has)]
        for spec in self.bisulfite:
            if spec[0] == "nondir":
                # ^C[AG]A on the read as it is now -> two bases off the 5' end (counting only what was cut after a
                # match); else, with rrbs, the RRBS cutter
                off = (batch.records[:, 2].to(torch.int64) & 0xFFFFFFFF) + begin.to(torch.int64)
                last = batch.data.numel() - 1
                c0, c1, c2 = (batch.data[(off + k).clamp(max=last)] for k in range(3))
                hit = (end - begin >= 3) & (c0 == 67) & ((c1 == 65) | (c1 == 71)) & (c2 == 65)
                self._min_cut(begin, end, total, has, sides, ab, ae, 2, 0, False, False, hit)
                if spec[1]:
                    # (NonDirectionalBisulfiteTrimmer builds its RRBSTrimmer as RRBSTrimmer(trim_3p): the 2 lands in
                    # trim_5p and trim_3p keeps its default -- two bases from EITHER trimmed end, modifiers.py:808-813)
                    self._min_cut(begin, end, total, has, sides, ab, ae, 2, 2, False, True, ~hit & (end > begin))
            else:
                _kind, front, back, count_trimmed, only_trimmed = spec
                self._min_cut(begin, end, total, has, sides, ab, ae, front, back, count_trimmed, only_trimmed, None)

    @staticmethod
    def _min_cut(begin, end, total, has, sides, ab, ae, front, back, count_trimmed, only_trimmed, where):
        live = end > begin                                                # Trimmer.clip leaves an empty read alone
        if where is not None:
            live = live & where
        trim_front = trim_back = live
        if only_trimmed:                                                  # :612-621
            trim_front = live & has & sides[0]
            trim_back = live & has & sides[1]
        if count_trimmed:
            gone_f, gone_b = begin, total - end
        else:
            gone_f = torch.where(has, begin - ab, begin)
            gone_b = torch.where(has, ae - end, total - end)
        tf = torch.where(trim_front, (front - gone_f).clamp(min=0), torch.zeros_like(begin))
        tb = torch.where(trim_back, (back - gone_b).clamp(min=0), torch.zeros_like(begin))
        nb = torch.minimum(begin + tf, end)
        ne = torch.maximum(end - tb, nb)
        begin.copy_(nb)
        end.copy_(ne)

    # ------------------------------------------------------------------ whole pipeline
    def _simple_stage(self, op, batch, begin, end):
        """The C, G, Q stages (interval updates without alignment)."""
        be = batch.backend
        if op == "C" and (self.cut_front or self.cut_back):
            be.clip_batch(batch.records, begin, end, self.cut_front, self.cut_back)
        elif op == "G" and self.nextseq_trim is not None:
            be.quality_trim_batch(batch.data, batch.records, begin, end, 0, int(self.nextseq_trim),
                                  self.quality_base, True)
        elif op == "Q" and self.quality_cutoff:
            be.quality_trim_batch(batch.data, batch.records, begin, end, int(self.quality_cutoff[0]),
                                  int(self.quality_cutoff[1]), self.quality_base, False)

    def _filter_stage(self, batch, begin, end, ubegin, uend, matched, masks=False):
        """--trim-n, then the read filters: destination byte per read (or the fail masks)."""
        be = batch.backend
        if self.bisulfite:
            self._bisulfite_stage(batch, begin, end, matched)
        if self.trim_n:
            be.nend_trim_batch(batch.data, batch.records, begin, end, ubegin, uend)
        if self.min_front or self.min_back:
            # at least min_front / min_back bases gone from the two ends, whatever removed them so far: everything that
            # was cut, quality-trimmed or adapter-trimmed is in the interval (clipped[] + the matches' rsize_total)
            live = end > begin                                            # Trimmer.clip leaves an empty read alone
            total = batch.seq_lens
            nb = torch.minimum(torch.maximum(begin, torch.full_like(begin, self.min_front)), end)
            ne = torch.maximum(torch.minimum(end, total - self.min_back), nb)
            begin.copy_(torch.where(live, nb, begin))
            end.copy_(torch.where(live, ne, end))
        min_len = self.minimum_length if self.minimum_length is not None and self.minimum_length > 0 else 0
        max_len = self.maximum_length if self.maximum_length is not None else -1
        max_n = float(self.max_n) if self.max_n is not None else -1.0
        return be.read_filter_batch(batch.data, batch.records, begin, end, ubegin, uend, matched, min_len, max_len,
                                    max_n, self.discard_trimmed, self.discard_untrimmed, masks=masks)

    def run(self, batch):
        """All stages over one FastqBatch; returns a TrimResult."""
        n = len(batch)
        begin = torch.zeros((n,), dtype=torch.int32, device=batch.records.device)
        end = batch.seq_lens.clone()
        matched = torch.zeros((n,), dtype=torch.uint8, device=begin.device)
        ubegin = uend = None
        self._rounds = [] if self.aux else None
        self._sides = self._after = None
        unmasked = None
        self._last_which = (torch.zeros((n,), dtype=torch.int64, device=begin.device)
                            if ("{name}" in self.prefix or "{name}" in self.suffix) else None)
        for op in self.op_order:
            if op == "A":
                if self.adapters:
                    matched, ubegin, uend = self._adapter_stage(batch, begin, end)
                    if ubegin is not None and mask_before_later_stages(self.op_order, self.action):
                        if self._rounds is not None:          # (the info / rest / wildcard lines show the read a match saw)
                            unmasked = batch.data.clone()
                        write_mask(batch, begin, end, ubegin, uend)
                        ubegin = uend = None
            else:
                self._simple_stage(op, batch, begin, end)
        dest = self._filter_stage(batch, begin, end, ubegin, uend, matched)
        rounds, self._rounds = self._rounds, None
        if self.zero_cap and n:
            if rounds is not None and unmasked is None:            # a match's info record shows the read as the match saw it:
                unmasked = batch.data.clone()                      # before ZeroCapper (align/__init__.py:145-170)
            self._zero_cap(batch)
        read_batch = batch if unmasked is None else FastqBatch(unmasked, batch.nbytes, batch.records, batch.backend,
                                                               batch.line_ends)
        if self._name_mods and n:
            batch = self._rewrite_names(batch, begin, end, matched)
        self._last_which = None
        return TrimResult(batch, begin, end, ubegin, uend, matched, dest, rounds, self.adapters, read_batch)

    def _zero_cap(self, batch):
        """ZeroCapper (modifiers.py:709-720): quality characters below the base become the base, in the chunk."""
        rec = batch.records
        lens = rec[:, 5].to(torch.int64)
        start = rec[:, 4].to(torch.int64) & 0xFFFFFFFF
        first = torch.cumsum(lens, 0) - lens
        idx = torch.repeat_interleave(start - first, lens) + torch.arange(int(lens.sum().item()), device=rec.device)
        batch.data[idx] = batch.data[idx].clamp(min=self.quality_base)

    def _rewrite_names(self, batch, begin, end, matched):
        """LengthTagModifier, SuffixRemover, PrefixSuffixAdder (modifiers.py:652-695) in the reference's order: the
        new names are made on the host (string work per read, as in the reference), appended to the chunk in device
        memory, and the records point at them -- the formatter copies whatever name a record names."""
        import re
        import numpy as np
        raw = bytes(batch.data[:batch.nbytes].cpu().numpy().tobytes())
        recs = batch.records.cpu().numpy().astype("int64")
        lens = (end - begin).clamp(min=0).cpu().tolist()
        took = matched.cpu().tolist()
        which = self._last_which.cpu().tolist() if self._last_which is not None else None
        tag = self.length_tag
        regex = re.compile(r"\b" + tag + r"[0-9]*\b") if tag else None
        names, offs, cur = [], [], 0
        for i in range(recs.shape[0]):
            off = recs[i, 0] & 0xFFFFFFFF
            name = raw[off:off + recs[i, 1]].decode("latin-1")
            if tag and name.find(tag) >= 0:
                name = regex.sub(tag + str(lens[i]), name)
            for sfx in self.strip_suffix:
                if name.endswith(sfx):
                    name = name[:-len(sfx)]
            if self.prefix or self.suffix:
                ad = self.adapters[which[i]].name if (which is not None and took[i]) else "no_adapter"
                name = self.prefix.replace("{name}", ad) + name + self.suffix.replace("{name}", ad)
            blob = name.encode("latin-1")
            names.append(blob)
            offs.append(cur)
            cur += len(blob)
        base = (batch.data.numel() + 15) // 16 * 16
        if base + cur >= (1 << 32) - 16:
            raise ValueError("the chunk and its rewritten names must stay below 4 GiB")
        extra = torch.from_numpy(np.frombuffer(b"".join(names) + b"\0" * 16, dtype=np.uint8).copy()).to(batch.data.device)
        data = torch.cat([batch.data, torch.zeros((base - batch.data.numel(),), dtype=torch.uint8, device=extra.device), extra])
        noff = np.asarray(offs, dtype=np.int64) + base
        # the '+' line keeps the text the file had (FastqFormat prints name2, io/seqio.py:690-699): flag bit 1,
        # the old name's offset in `reserved`, its length above the flag bits (atr_fastq_record)
        rep = (recs[:, 6] & 1) != 0
        recs[:, 7] = np.where(rep, recs[:, 0], recs[:, 7])
        flags = np.where(rep, (recs[:, 6] & 1) | 2 | (recs[:, 1] << 8), recs[:, 6])
        recs[:, 6] = np.where(flags >= (1 << 31), flags - (1 << 32), flags)
        recs[:, 0] = np.where(noff >= (1 << 31), noff - (1 << 32), noff)
        recs[:, 1] = [len(b) for b in names]
        records = torch.from_numpy(recs.astype(np.int32)).to(batch.records.device)
        return FastqBatch(data, batch.nbytes, records, batch.backend, batch.line_ends)

    def trim_bytes(self, data, which=_lib.DEST_KEEP):
        """FASTQ text in, trimmed FASTQ text out (one batch)."""
        batch, _ = FastqBatch.from_bytes(data, final=True)
        return self.run(batch).text(which)

    def trim_file(self, path_in, path_out, chunk_bytes=256 << 20, keep_output=False, output_parts=1):
        """Stream a FASTQ file through the GPU in chunks of whole records; returns the
        destination counts.  (Plain files; compressed input is the caller's business.)
        Host side: ``ChunkedFastqReader`` / ``FastqSink`` (page-locked staging buffers, threaded
        reads, read-ahead; device -> host copies on their own stream and write-behind, so the GPU
        works on chunk i + 1 while chunk i travels back and into the file).  ``keep_output``: overwrite
        an existing output file in place instead of truncating it first.  ``output_parts`` > 1: the output
        as that many part files ``<path_out>.part<i>`` with a writer each (``fastq.PartSink``: chunk k goes to
        part k mod N; what the reference's ``--no-writer-process`` does with its worker processes) -- for hosts
        that serialise the writers of one file.  The seconds the loop spent waiting per stage are left in
        ``self.stage_seconds``."""
        import time
        from .fastq import ChunkedFastqReader, StageClock, make_sink
        be = _lib.get_backend()
        totals = {name: 0 for name in DEST_NAMES.values()}
        clock = StageClock()
        reader = ChunkedFastqReader(path_in, chunk_bytes, be, clock)
        sink = make_sink(path_out, output_parts, chunk_bytes + (64 << 20) + 32, be, clock, keep=keep_output)
        aux_files = {kind: open_by_extension(path) for kind, path in (self.aux or {}).items()}
        dest_codes = {name: code for code, name in DEST_NAMES.items()}
        dest_files = {dest_codes[kind]: open_by_extension(path) for kind, path in self.outputs.items()}
        try:
            while True:
                batch = reader.next_batch()
                done = reader.advance()                       # starts reading the next chunk
                t0 = time.perf_counter()
                res = self.run(batch)
                text = be.fastq_emit(res.batch.data, res.batch.records, res.begin, res.end, res.ubegin, res.uend, res.dest,
                                     _lib.DEST_KEEP)
                counts = res.counts()
                clock.add("trim_and_format", t0)
                sink.write(text)
                if aux_files:                                 # host-assembled lines (debugging outputs, not a throughput path)
                    for kind, blob in res.aux_text(tuple(aux_files)).items():
                        aux_files[kind].write(blob)
                for code, fh in dest_files.items():           # the filtered reads that have a file of their own
                    fh.write(res.text(code))
                for name, v in counts.items():
                    totals[name] += v
                if done:
                    break
        finally:
            reader.close()
            sink.close()
            for fh in list(aux_files.values()) + list(dest_files.values()):
                fh.close()
            self.stage_seconds = dict(clock.seconds)
        return totals


class PairedTrimResult(object):
    """Result of the paired-end pipeline: one TrimResult per read, sharing the destination."""

    def __init__(self, res1, res2, merged=None):
        self.read1, self.read2, self.dest = res1, res2, res1.dest
        self.merged = merged                     # uint8 device tensor: the FASTQ text of the merged reads, or None

    def counts(self):
        c = torch.bincount(self.dest.to(torch.int64), minlength=7).cpu().tolist()
        out = {DEST_NAMES[i]: int(c[i]) for i in range(6)}
        if self.merged is not None:
            out["merged"] = int(c[DEST_MERGED])
        return out

    def aux_text(self, kinds=("info", "rest", "wildcard")):
        """The info / rest / wildcard lines of the pairs: read 1's, then read 2's, pair after pair
        (Formatters.format calls every info formatter on read 1 and on read 2, writers.py:156-159)."""
        l1, l2 = self.read1.aux_lines(kinds), self.read2.aux_lines(kinds)
        # (`if read2:` in Formatters.format: a read 2 that was trimmed to nothing is false and gets no lines)
        has2 = (self.read2.end > self.read2.begin).cpu().tolist()
        return {k: "".join(line + "\n" for a, b, two in zip(l1[k], l2[k], has2) for line in (a + b if two else a)).encode("ascii", "replace")
                for k in kinds}

    def merged_text(self):
        """FASTQ text of the merged reads (the --merged-output file), in input order."""
        return b"" if self.merged is None else bytes(self.merged.cpu().numpy().tobytes())

    def text(self, which=_lib.DEST_KEEP):
        return self.read1.text(which), self.read2.text(which)


class PairedTrimPipeline(object):
    """Paired-end trimming in "both" mode (both reads are modified; trim/cli.py:633-648) as
    whole-batch device stages.  ``aligner``: 'adapter' -- one AdapterCutter per read
    (``adapters1`` / ``adapters2``), or 'insert' -- InsertAdapterCutter with exactly one 3'
    adapter per read (trim/__init__.py:406-456).  ``pair_filter``: 'any' or 'both'.
    ``correct_mismatches`` ('liberal' | 'conservative' | 'N', insert aligner only): error
    correction of the overlaps, written IN PLACE into the two FASTQ chunks in device memory
    (a batch is consumed by ``run``).  ``merge_overlapping``: MergeOverlapping as the last modifier
    (commands/trim/modifiers.py:864-931, trim/__init__.py:546-552): pairs whose reads overlap by
    ``merge_min_overlap`` (a fraction of the shorter read up to 1, else bases) at ``merge_error_rate``
    become ONE read -- destination ``DEST_MERGED``, text in ``PairedTrimResult.merged`` -- corrected
    first with ``correct_mismatches`` unless the insert aligner already saw the pair."""

    def __init__(self, adapters1=(), adapters2=(), aligner="adapter", times=1, action="trim", cut=(), cut2=(),
                 nextseq_trim=None, quality_cutoff=None, quality_base=33, trim_n=False, minimum_length=None,
                 maximum_length=None, max_n=None, discard_trimmed=False, discard_untrimmed=False, pair_filter="any",
                 op_order="CGQAW", insert_args=None, correct_mismatches=None, merge_overlapping=False,
                 merge_min_overlap=0.9, merge_error_rate=0.2, aux=None, length_tag=None, strip_suffix=(), prefix="",
                 suffix="", zero_cap=False, outputs=None, cut_min=(), cut_min2=(), bisulfite=None, bisulfite2=None):
        # {"too_short" | "too_long" | "untrimmed": (path for read 1, path for read 2)}: the filtered pairs' own files
        self.outputs = dict(outputs) if outputs else {}
        if "untrimmed" in self.outputs:
            discard_untrimmed = True                                      # (the UntrimmedFilter is on, trim/__init__.py:617)
        common = dict(times=times, action=action, nextseq_trim=nextseq_trim, quality_cutoff=quality_cutoff,
                      quality_base=quality_base, trim_n=trim_n, minimum_length=minimum_length,
                      maximum_length=maximum_length, max_n=max_n, discard_trimmed=discard_trimmed,
                      discard_untrimmed=discard_untrimmed, op_order=op_order, aux=aux, length_tag=length_tag,
                      strip_suffix=strip_suffix, prefix=prefix, suffix=suffix, zero_cap=zero_cap)
        self.aux = dict(aux) if aux else None
        if (aux or length_tag or strip_suffix or prefix or suffix or zero_cap) and (aligner != "adapter" or merge_overlapping):
            raise NotImplementedError("--info-file / --rest-file / --wildcard-file, read-name modifiers and --zero-cap with "
                                      "the insert aligner or with merging")
        self.p1 = TrimPipeline(adapters=adapters1, cut=cut, cut_min=cut_min, bisulfite=bisulfite, **common)
        self.p2 = TrimPipeline(adapters=adapters2, cut=cut2, cut_min=cut_min2, bisulfite=bisulfite2, **common)
        if (bisulfite or bisulfite2) and aligner != "adapter":
            raise NotImplementedError("--bisulfite with the insert aligner")
        self.aligner, self.action, self.op_order = aligner, action, op_order
        if pair_filter not in ("any", "both"):
            raise ValueError("pair_filter must be 'any' or 'both'")
        self.min_affected = 2 if pair_filter == "both" else 1            # trim/__init__.py:549
        if correct_mismatches not in (None, "liberal", "conservative", "N"):
            raise ValueError("correct_mismatches must be 'liberal', 'conservative' or 'N'")
        if correct_mismatches and aligner != "insert" and not merge_overlapping:
            raise ValueError("error correction needs the insert aligner or --merge-overlapping")
        self.merge_overlapping = bool(merge_overlapping)
        # above 1: a number of bases; up to 1: a fraction of the shorter read (modifiers.py:867)
        self.merge_min_overlap = merge_min_overlap if merge_min_overlap <= 1 else int(merge_min_overlap)
        self.merge_error_rate = float(merge_error_rate)
        self.merged_pairs = 0
        self.correct_mismatches = correct_mismatches
        self.corrected_pairs, self.corrected_bp = 0, [0, 0]               # ErrorCorrectorMixin counters
        self.insert = None
        if aligner == "insert":
            from .adapters import BACK
            from .align import InsertAligner
            if len(self.p1.adapters) != 1 or len(self.p2.adapters) != 1 or any(
                    isinstance(a, LinkedAdapter) or a.where != BACK for a in self.p1.adapters + self.p2.adapters):
                raise ValueError("Insert aligner requires a single 3' adapter for each read")   # trim/__init__.py:406-411
            self.insert = InsertAligner(self.p1.adapters[0].sequence, self.p2.adapters[0].sequence, **(insert_args or {}))
        elif aligner != "adapter":
            raise ValueError("aligner must be 'adapter' or 'insert'")

    @staticmethod
    def _fallback(adapter, batch, st, miss, n):
        """Adapter.match_to records for the reads listed in ``miss`` (-1 records elsewhere)."""
        none = torch.zeros((n, 8), dtype=torch.int16, device=st[0].device)
        none[:, 1] = -1
        if miss.numel() == 0:
            return none
        if miss.numel() == n:
            return adapter.match_source(RecordSource(batch, st[0], st[1]))
        sub = FastqBatch(batch.data, batch.nbytes, batch.records.index_select(0, miss).contiguous(), batch.backend)
        rec = adapter.match_source(RecordSource(sub, st[0].index_select(0, miss).contiguous(),
                                                st[1].index_select(0, miss).contiguous()))
        none[miss] = rec
        return none

    def _insert_stage(self, b1, b2, st1, st2):
        """InsertAdapterCutter over the batch (commands/trim/modifiers.py:391-496, no error correction)."""
        be = b1.backend
        n = len(b1)
        src1, src2 = RecordSource(b1, st1[0], st1[1]), RecordSource(b2, st2[0], st2[1])
        max_len = 0
        if n:
            max_len = int(torch.maximum((st1[1] - st1[0]).max(), (st2[1] - st2[0]).max()).clamp_(min=0).item())
        if max_len > _lib.INSERT_MAX_READ:
            raise _lib.AtroposHipError("InsertAligner: reads longer than %d bases are outside the device envelope"
                                       % _lib.INSERT_MAX_READ)
        table = be.translate_table(_lib.TABLE_DNA15)
        pb1 = src1.planes(max_len, _lib.TABLE_DNA15, table, count=True)
        pb2 = src2.planes(max_len, _lib.TABLE_DNA15, table, count=True)
        if pb1.uncoded_reads or pb2.uncoded_reads:
            # soft-masked reads: match_insert tells the cases apart in the insert compare and folds them in the adapter
            # compares (atr_insert_match_batch_coded) -- both reads again, with the case-sensitive codes
            table = be.case_sensitive_table()
            pb1 = src1.planes(max_len, _lib.TABLE_CUSTOM, table)
            pb2 = src2.planes(max_len, _lib.TABLE_CUSTOM, table)
        # match_insert complements read 2 only as far as read 1 reaches (align/__init__.py:259-267)
        bad = be.planes_count_uncoded(pb2.packed, pb2.lens, pb1.lens, n, max_len)
        if bad:
            raise ValueError("%d second read(s) contain bases without an upper-case IUPAC code where they face the first "
                             "read; the device insert aligner cannot reverse-complement them" % bad)
        ins = self.insert.match_insert_batch(pb1, pb2).records
        self._insert_matched = (ins[:, 0, 1] >= 0).to(torch.uint8)          # read.insert_overlap, modifiers.py:397
        # semi-global fallback (modifiers.py:405-407): only the pairs without an insert match need it
        miss = torch.nonzero(ins[:, 0, 1] < 0).squeeze(1)
        fb1 = self._fallback(self.p1.adapters[0], b1, st1, miss, n)
        fb2 = self._fallback(self.p2.adapters[0], b2, st2, miss, n)
        action = {None: 0, "trim": 1, "mask": 2}[self.action]
        uend1 = uend2 = None
        if action == 2:
            uend1, uend2 = st1[1].clone(), st2[1].clone()
        from .modifiers import COMP_TABLE, _ACTIONS
        correct = _ACTIONS[self.correct_mismatches] if self.correct_mismatches else -1
        # NB: with error correction the bases / qualities of both FASTQ chunks are rewritten in place
        m1, m2, corrected, err = be.insert_plan_batch(
            ins.contiguous(), fb1.contiguous(), fb2.contiguous(), b1, b2, st1[0], st1[1], st2[0], st2[1], uend1, uend2,
            self.insert.min_insert_overlap, True, action, correct, 1, COMP_TABLE)
        if err != _lib.INT64_MAX:                                          # the exception the reference raises
            exc = {1: KeyError, 2: IndexError, 3: ValueError}[err % 8]
            raise exc("error correction of pair %d: %s" % (err // 8, {
                1: "base without a complement", 2: "overlap outside a read",
                3: "Cannot determine the mode of an empty sequence"}[err % 8]))
        self._already_corrected = None
        if correct >= 0:
            # read.corrected of both reads: a later correct_errors call (MergeOverlapping) returns at once for
            # these pairs (modifiers.py:232-233) -- also for the ones corrected from complementary fallback
            # matches, which have no insert match
            self._already_corrected = (corrected.sum(dim=1) > 0).to(torch.uint8)
            self.corrected_pairs += int(self._already_corrected.sum().item())
            tot = corrected.sum(dim=0).cpu().tolist()
            self.corrected_bp[0] += int(tot[0])
            self.corrected_bp[1] += int(tot[1])
        if action == 2:
            return (m1, st1[0].clone(), uend1), (m2, st2[0].clone(), uend2)
        return (m1, None, None), (m2, None, None)

    def _merge_stage(self, b1, b2, st1, st2, insert_matched, already_corrected=None):
        """MergeOverlapping over the batch: the alignments ``Aligner(reverse_complement(read2), error_rate,
        flags).locate(read1)`` of the pairs long enough to be tried (atr_locate_pairs_batch, one call per
        flag set), then the merge itself.  Returns (merged flags uint8 [n], merged FASTQ text)."""
        from .align import SEMIGLOBAL, START_WITHIN_SEQ1, STOP_WITHIN_SEQ2
        from .modifiers import COMP_TABLE, _ACTIONS
        be = b1.backend
        n = len(b1)
        dev = st1[0].device
        len1, len2 = (st1[1] - st1[0]).clamp(min=0), (st2[1] - st2[0]).clamp(min=0)
        shorter = torch.minimum(len1, len2)
        if self.merge_min_overlap > 1:
            need = torch.full_like(shorter, int(self.merge_min_overlap))
        else:                                                              # max(2, round(frac * shorter)), :870-874
            need = torch.round(shorter.to(torch.float64) * float(self.merge_min_overlap)).to(torch.int32).clamp(min=2)
        tried = shorter >= need                                            # :876-877
        if insert_matched is None:
            insert_matched = torch.zeros((n,), dtype=torch.uint8, device=dev)
        align = torch.zeros((n, 8), dtype=torch.int16, device=dev)
        align[:, 1] = -1
        from .align import case_sensitive_pair_table
        tables = [be.translate_table(_lib.TABLE_DNA15)]
        tables.append(case_sensitive_pair_table(tables[0]))     # soft-masked reads: second try, both reads with it
        for flags, group in ((START_WITHIN_SEQ1 | STOP_WITHIN_SEQ2, tried & (insert_matched != 0)),
                             (SEMIGLOBAL, tried & (insert_matched == 0))):      # :879-886
            idx = torch.nonzero(group).squeeze(1)
            if idx.numel() == 0:
                continue
            sides = []
            for b, st in ((b2, st2), (b1, st1)):                            # reference = read 2, query = read 1
                sub = b if idx.numel() == n else FastqBatch(b.data, b.nbytes, b.records.index_select(0, idx).contiguous(),
                                                            b.backend)
                begin, end = (st[0], st[1]) if idx.numel() == n else (st[0].index_select(0, idx).contiguous(),
                                                                      st[1].index_select(0, idx).contiguous())
                max_len = int((end - begin).max().clamp_(min=0).item())
                if max_len > _lib.MAX_READ_LEN:
                    raise _lib.AtroposHipError("MergeOverlapping: reads longer than %d bases are outside the device "
                                               "envelope" % _lib.MAX_READ_LEN)
                sides.append((sub, begin, end, max_len))
            for table in tables:
                # the aligner compares characters (_align.pyx:390-391): upper-case IUPAC codes first; if a read is
                # soft-masked, both reads again with the table that tells the cases apart
                packs, bad = [], 0
                for sub, begin, end, max_len in sides:
                    packed, lens, nbad = be.pack_records(sub.data, sub.records, begin, end, max_len, table, count_invalid=True)
                    packs.append((packed, lens, max_len))
                    bad += nbad
                if not bad:
                    break
            if bad:
                raise ValueError("%d read(s) contain characters the device pair aligner has no 4-bit code for (upper-case "
                                 "IUPAC letters, or A C G T N W B D H V in either case)" % bad)
            (rp, rl, rmax), (qp, ql, qmax) = packs
            # the merge below only looks at alignments with matches >= need (modifiers.py:896-897)
            need_g = need.to(torch.int32) if idx.numel() == n else need.to(torch.int32).index_select(0, idx)
            if max(rmax, qmax) > _lib.PAIRS_MAX_LEN:        # reads of 321 .. 736 bases: the per-pair aligner's long path
                from .align import long_pairs_records
                rec = long_pairs_records(be, rp, rl, rmax, True, qp, ql, qmax, int(idx.numel()), self.merge_error_rate, flags,
                                         False, False, 1, 1)
            else:
                rec = be.locate_pairs_batch(rp, rl, rmax, True, qp, ql, qmax, int(idx.numel()), self.merge_error_rate, flags,
                                            False, False, 1, 1, need=need_g.contiguous())
            if idx.numel() == n:
                align = rec
            else:
                align[idx] = rec
        correct = _ACTIONS[self.correct_mismatches] if self.correct_mismatches else -1
        # pairs the merge must not correct: bit 0 the insert aligner saw them (modifiers.py:900), bit 1 a read of
        # the pair was corrected before (correct_errors' own guard, :232-233)
        no_correct = insert_matched if already_corrected is None else insert_matched | (already_corrected << 1)
        kind, text, corrected, err = be.merge_batch(align.contiguous(), need.to(torch.int32).contiguous(),
                                                    no_correct.contiguous(), b1, b2, st1[0], st1[1], st2[0], st2[1],
                                                    correct, 1, COMP_TABLE)
        if err != _lib.INT64_MAX:
            if err % 8 == 4:
                raise ValueError("Invalid alignment while trying to merge pair %d" % (err // 8))
            exc = {1: KeyError, 2: IndexError, 3: ValueError}[err % 8]
            raise exc("error correction of pair %d: %s" % (err // 8, {
                1: "base without a complement", 2: "overlap outside a read",
                3: "Cannot determine the mode of an empty sequence"}[err % 8]))
        merged = kind != 0
        if correct >= 0:
            self.corrected_pairs += int((corrected.sum(dim=1) > 0).sum().item())
            tot = corrected.sum(dim=0).cpu().tolist()
            self.corrected_bp[0] += int(tot[0])
            self.corrected_bp[1] += int(tot[1])
        self.merged_pairs += int(merged.sum().item())
        return merged, text

    def run(self, batch1, batch2):
        if len(batch1) != len(batch2):
            raise ValueError("the two FASTQ batches hold different numbers of records")
        be = batch1.backend
        n = len(batch1)
        dev = batch1.records.device
        st = []
        for b in (batch1, batch2):
            st.append([torch.zeros((n,), dtype=torch.int32, device=dev), b.seq_lens.clone()])
        extra = [(torch.zeros((n,), dtype=torch.uint8, device=dev), None, None) for _ in range(2)]
        pipes, batches = (self.p1, self.p2), [batch1, batch2]
        insert_matched = already_corrected = None
        unmasked = [None, None]
        for pipe in pipes:
            pipe._sides = pipe._after = None
            pipe._rounds = [] if self.aux else None
            pipe._last_which = (torch.zeros((n,), dtype=torch.int64, device=dev)
                                if ("{name}" in pipe.prefix or "{name}" in pipe.suffix) else None)
        for op in self.op_order:
            if op == "A":
                if self.aligner == "insert":
                    extra = list(self._insert_stage(batch1, batch2, st[0], st[1]))
                    insert_matched, already_corrected = self._insert_matched, self._already_corrected
                else:
                    for k in range(2):
                        if pipes[k].adapters:
                            extra[k] = pipes[k]._adapter_stage(batches[k], st[k][0], st[k][1])
                for k in range(2):
                    if extra[k][1] is not None and mask_before_later_stages(self.op_order, self.action):
                        if pipes[k]._rounds is not None:
                            unmasked[k] = batches[k].data.clone()
                        write_mask(batches[k], st[k][0], st[k][1], extra[k][1], extra[k][2])
                        extra[k] = (extra[k][0], None, None)
            else:
                for k in range(2):
                    pipes[k]._simple_stage(op, batches[k], st[k][0], st[k][1])
        masks = []
        for k in range(2):
            matched, ub, ue = extra[k]
            masks.append(pipes[k]._filter_stage(batches[k], st[k][0], st[k][1], ub, ue, matched, masks=True))
        dest = be.pair_filter_batch(masks[0], masks[1], self.min_affected)
        merged_text = None
        if self.merge_overlapping:                                  # the last modifier, the first filter
            if self.action == "mask":
                # masked adapters: MergeOverlapping sees the reads with their N's (modifiers.py:155-172 made them part of
                # the sequence): written into the chunk here, the intervals then need no mask any more
                for k in range(2):
                    matched, ub, ue = extra[k]
                    if ub is not None:
                        self._write_mask(batches[k], st[k][0], st[k][1], ub, ue)
                        extra[k] = (matched, None, None)
            merged, merged_text = self._merge_stage(batch1, batch2, st[0], st[1], insert_matched, already_corrected)
            dest = torch.where(merged, torch.full_like(dest, DEST_MERGED), dest)
        res = []
        for k in range(2):
            pipe, read_batch = pipes[k], batches[k]
            rounds, pipe._rounds = pipe._rounds, None
            if pipe.zero_cap and n and rounds is not None and unmasked[k] is None:
                unmasked[k] = batches[k].data.clone()              # (the info records of matched reads: before ZeroCapper)
            if unmasked[k] is not None:
                read_batch = FastqBatch(unmasked[k], read_batch.nbytes, read_batch.records, read_batch.backend, read_batch.line_ends)
            if pipe.zero_cap and n:
                pipe._zero_cap(batches[k])
            if pipe._name_mods and n:
                batches[k] = pipe._rewrite_names(batches[k], st[k][0], st[k][1], extra[k][0])
            pipe._last_which = None
            res.append(TrimResult(batches[k], st[k][0], st[k][1], extra[k][1], extra[k][2], extra[k][0], dest, rounds,
                                  pipe.adapters, read_batch))
        return PairedTrimResult(res[0], res[1], merged_text)

    _write_mask = staticmethod(lambda batch, begin, end, ubegin, uend: write_mask(batch, begin, end, ubegin, uend))

    def trim_files(self, in1, in2, out1, out2, chunk_bytes=128 << 20, merged_out=None, keep_output=False, output_parts=1):
        """Stream two FASTQ files through the GPU in lock step (chunks of whole records, the
        same number from each file); returns the destination counts.  ``merged_out``: the
        --merged-output file (without it merged reads are dropped, as by the reference).
        ``output_parts`` > 1: every output as that many part files (``TrimPipeline.trim_file``); part i of
        ``out1`` and part i of ``out2`` hold the same pairs in the same order."""
        import time
        from .fastq import ChunkedFastqReader, StageClock, make_sink
        be = _lib.get_backend()
        totals = {name: 0 for name in DEST_NAMES.values()}
        clock = StageClock()
        readers = [ChunkedFastqReader(p, chunk_bytes, be, clock) for p in (in1, in2)]
        sinks = [make_sink(p, output_parts, chunk_bytes + (64 << 20) + 32, be, clock, keep=keep_output) for p in (out1, out2)]
        aux_files = {kind: open_by_extension(path) for kind, path in (self.aux or {}).items()}
        dest_codes = {name: code for code, name in DEST_NAMES.items()}
        dest_files = {dest_codes[kind]: [open_by_extension(p) for p in paths] for kind, paths in self.outputs.items()}
        if self.merge_overlapping:
            totals["merged"] = 0
            if merged_out is not None:
                # a merged record is at most its two input records, and each input chunk may carry up to 64 MB over
                sinks.append(make_sink(merged_out, output_parts, 2 * (chunk_bytes + (64 << 20)) + 32, be, clock, keep=keep_output))
        try:
            while True:
                batches = [r.next_batch() for r in readers]
                nrec = min(len(batches[0]), len(batches[1]))
                heads = [batches[k].head(nrec) for k in range(2)]
                done = [readers[k].advance(heads[k][1]) for k in range(2)]
                if all(r.final for r in readers) and len(batches[0]) != len(batches[1]):
                    raise ValueError("the two input files hold different numbers of records")
                t0 = time.perf_counter()
                res = self.run(heads[0][0], heads[1][0])
                texts = [be.fastq_emit(r.batch.data, r.batch.records, r.begin, r.end, r.ubegin, r.uend, r.dest,
                                       _lib.DEST_KEEP) for r in (res.read1, res.read2)]
                counts = res.counts()
                clock.add("trim_and_format", t0)
                for k in range(2):
                    sinks[k].write(texts[k])
                if len(sinks) == 3:
                    sinks[2].write(res.merged if res.merged is not None else texts[0][:0])
                if aux_files:
                    for kind, blob in res.aux_text(tuple(aux_files)).items():
                        aux_files[kind].write(blob)
                for code, fhs in dest_files.items():
                    for fh, text in zip(fhs, res.text(code)):
                        fh.write(text)
                for name, v in counts.items():
                    totals[name] += v
                if all(done):
                    break
                if any(done):
                    raise ValueError("the two input files hold different numbers of records")
        finally:
            for obj in readers + sinks + list(aux_files.values()) + [fh for fhs in dest_files.values() for fh in fhs]:
                obj.close()
            self.stage_seconds = dict(clock.seconds)
        return totals

    def trim_bytes(self, data1, data2, which=_lib.DEST_KEEP):
        """Two FASTQ texts in (same number of records), the two trimmed texts out."""
        b1, _ = FastqBatch.from_bytes(data1, final=True)
        b2, _ = FastqBatch.from_bytes(data2, final=True)
        return self.run(b1, b2).text(which)


class LegacyPairedPipeline(PairedTrimPipeline):
    """Paired-end input without any option that switches full paired-end trimming on (-A / -G / -B / -U, -q,
    --trim-n, --pair-filter ...): the reference's backwards-compatible 'legacy mode' (trim/cli.py:630-648) -- the
    single-end pipeline on read 1, its filters on read 1 alone (FilterFactory wraps them in SingleWrapper,
    filters.py:100-107), read 2 written as it came for every pair that is kept."""

    def __init__(self, first):
        self.first = first
        self.merge_overlapping, self.aux, self.outputs = False, None, {}
        self.p1 = first

    def run(self, batch1, batch2):
        if len(batch1) != len(batch2):
            raise ValueError("the two FASTQ batches hold different numbers of records")
        res1 = self.first.run(batch1)
        n = len(batch2)
        dev = batch2.records.device
        res2 = TrimResult(batch2, torch.zeros((n,), dtype=torch.int32, device=dev), batch2.seq_lens.clone(), None, None,
                          torch.zeros((n,), dtype=torch.uint8, device=dev), res1.dest)
        return PairedTrimResult(res1, res2)


def pipeline_from_args(argv, paired_input=False):
    """Build a TrimPipeline (or, when paired-end options are present, a PairedTrimPipeline) from
    the subset of ``atropos trim`` command-line options the device pipeline covers (same
    spellings and defaults as trim/cli.py:57-335, :455-530, :655-803).  Anything else raises --
    the caller then uses the per-read object path.  ``paired_input``: the reads come as pairs (-pe1 / -pe2); without
    an option that asks for full paired-end trimming that is the reference's legacy mode (LegacyPairedPipeline)."""
    import argparse
    from .adapters import AdapterParser
    if isinstance(argv, str):
        argv = argv.split()
    ap = argparse.ArgumentParser(prog="atropos_amd.trim", add_help=False)
    ap.add_argument("-a", "--adapter", action="append", default=[], dest="adapters")
    ap.add_argument("-g", "--front", action="append", default=[])
    ap.add_argument("-b", "--anywhere", action="append", default=[])
    ap.add_argument("-A", action="append", default=[], dest="adapters2")
    ap.add_argument("-G", action="append", default=[], dest="front2")
    ap.add_argument("-B", action="append", default=[], dest="anywhere2")
    ap.add_argument("--aligner", choices=("adapter", "insert"), default="adapter")
    ap.add_argument("-e", "--error-rate", type=float, default=None)
    ap.add_argument("-O", "--overlap", type=int, default=None)
    ap.add_argument("-n", "--times", type=int, default=1)
    ap.add_argument("-N", "--no-match-adapter-wildcards", action="store_false", dest="match_adapter_wildcards",
                    default=True)
    ap.add_argument("--match-read-wildcards", action="store_true", default=False)
    ap.add_argument("--no-indels", action="store_false", dest="indels", default=True)
    ap.add_argument("--indel-cost", type=int, default=None)
    ap.add_argument("--adapter-max-rmp", type=float, default=None)
    ap.add_argument("--insert-max-rmp", type=float, default=1E-6)
    ap.add_argument("--insert-match-error-rate", type=float, default=None)
    ap.add_argument("--insert-match-adapter-error-rate", type=float, default=None)
    ap.add_argument("--no-trim", action="store_true", default=False)
    ap.add_argument("--mask-adapter", action="store_true", default=False)
    ap.add_argument("--op-order", default="CGQAW")
    ap.add_argument("-u", "--cut", type=int, action="append", default=[])
    ap.add_argument("-U", type=int, action="append", default=[], dest="cut2")
    ap.add_argument("--bisulfite", default=None)
    ap.add_argument("--cut-min", type=int, action="append", default=[])
    ap.add_argument("--cut-min2", type=int, action="append", default=[])
    ap.add_argument("-q", "--quality-cutoff", default=None)
    ap.add_argument("--quality-base", type=int, default=33)
    ap.add_argument("--nextseq-trim", type=int, default=None)
    ap.add_argument("--trim-n", action="store_true", default=False)
    ap.add_argument("-m", "--minimum-length", type=int, default=None)
    ap.add_argument("-M", "--maximum-length", type=int, default=None)
    ap.add_argument("--max-n", type=float, default=None)
    ap.add_argument("--discard-trimmed", "--discard", action="store_true", default=False)
    ap.add_argument("--discard-untrimmed", "--trimmed-only", action="store_true", default=False)
    ap.add_argument("--pair-filter", choices=("any", "both"), default=None)
    ap.add_argument("--correct-mismatches", choices=("liberal", "conservative", "N"), default=None)
    ap.add_argument("-R", "--merge-overlapping", action="store_true", default=False)
    ap.add_argument("--merge-min-overlap", type=float, default=0.9)
    ap.add_argument("--merge-error-rate", type=float, default=None)
    ap.add_argument("-x", "--prefix", default="")
    ap.add_argument("-y", "--suffix", default="")
    ap.add_argument("--strip-suffix", action="append", default=[])
    ap.add_argument("--length-tag", default=None)
    ap.add_argument("-z", "--zero-cap", action="store_true", default=False)
    ap.add_argument("--too-short-output", default=None)
    ap.add_argument("--too-long-output", default=None)
    ap.add_argument("--untrimmed-output", default=None)
    ap.add_argument("--too-short-paired-output", default=None)
    ap.add_argument("--too-long-paired-output", default=None)
    ap.add_argument("--untrimmed-paired-output", default=None)
    ap.add_argument("--info-file", default=None)
    ap.add_argument("--rest-file", "-r", default=None)
    ap.add_argument("--wildcard-file", default=None)
    o = ap.parse_args(argv)
    paired = bool(o.adapters2 or o.front2 or o.anywhere2 or o.cut2 or o.cut_min2 or o.pair_filter or o.aligner == "insert" or
                  o.merge_overlapping)
    for cm in (o.cut_min, o.cut_min2):                                     # cli.py:810-830
        if len(cm) > 2:
            raise ValueError("You cannot remove bases from more than two ends.")
        if len(cm) == 2 and cm[0] * cm[1] > 0:
            raise ValueError("You cannot remove bases from the same end twice.")
    if paired_input:
        # cli.py:630-641, the reference's own list: any of these asks for full paired-end trimming, none of them is its
        # legacy mode.  The list there also names interleaved_input and overwrite_low_quality: this parser has neither
        # option (argparse refuses them), so they cannot be set here -- whoever adds them must add them to `full` too, or
        # such a run would silently take the legacy pipeline.  (--aligner insert and -R are NOT on it: the reference then modifies read 1 alone, which the insert
        # aligner and the merge stage have no meaning for -- refused here rather than run as something else)
        full = bool(o.adapters2 or o.front2 or o.anywhere2 or o.cut2 or o.cut_min2 or o.quality_cutoff or o.trim_n or
                    o.pair_filter or o.too_short_paired_output or o.too_long_paired_output)
        if not full and (o.aligner == "insert" or o.merge_overlapping):
            raise NotImplementedError("--aligner insert / --merge-overlapping with paired-end input and none of the options "
                                      "that switch the reference's legacy mode off (trim/cli.py:630-641)")
        legacy = not full
        paired = full
    else:
        legacy = False
    if o.merge_min_overlap <= 0:
        raise ValueError("--merge-min-overlap must be positive")          # positive(float, True), cli.py:206-207
    if o.merge_overlapping and o.merge_error_rate is None:
        o.merge_error_rate = o.error_rate or 0.2                          # cli.py:690-691 (before -e gets its default)
    insert_args = None
    if o.aligner == "adapter":                                            # cli.py:659-666
        if o.indels and o.indel_cost is None:
            o.indel_cost = 1
        if o.overlap is None:
            o.overlap = 3 if o.adapter_max_rmp is None else 1
    else:                                                                 # cli.py:667-684
        if o.indels and o.indel_cost is None:
            o.indel_cost = 3
        if o.overlap is None:
            o.overlap = 1
            if o.adapter_max_rmp is None:
                o.adapter_max_rmp = 1E-6
        if o.insert_match_error_rate is None:
            o.insert_match_error_rate = o.error_rate or 0.2
        if o.insert_match_adapter_error_rate is None:
            o.insert_match_adapter_error_rate = o.insert_match_error_rate
        insert_args = dict(insert_max_rmp=o.insert_max_rmp, max_insert_mismatch_frac=o.insert_match_error_rate,
                           max_adapter_mismatch_frac=o.insert_match_adapter_error_rate,
                           read_wildcards=o.match_read_wildcards, adapter_wildcards=o.match_adapter_wildcards)
    if o.error_rate is None:
        o.error_rate = 0.1                                                # cli.py:801-802
    for cut in (o.cut, o.cut2):
        if len(cut) > 2 or (len(cut) == 2 and cut[0] * cut[1] > 0):
            raise ValueError("You cannot remove bases from the same end twice.")
    qc = None
    if o.quality_cutoff is not None:
        qc = [int(x) for x in str(o.quality_cutoff).split(",")]
        if all(c <= 0 for c in qc):
            qc = None                                                     # :750-754
        elif len(qc) == 1:
            qc = [0] + qc
    from .util import RandomMatchProbability
    kwargs = dict(max_error_rate=o.error_rate, min_overlap=o.overlap, read_wildcards=o.match_read_wildcards,
                  adapter_wildcards=o.match_adapter_wildcards, indels=o.indels,
                  match_probability=RandomMatchProbability())             # trim/__init__.py:345, :364
    if o.indel_cost is not None:
        kwargs["indel_cost"] = o.indel_cost
    if o.adapter_max_rmp:
        kwargs["max_rmp"] = o.adapter_max_rmp                             # trim/__init__.py:367-368
    parser = AdapterParser(**kwargs)
    adapters = parser.parse_multi(o.adapters, o.anywhere, o.front)
    action = None if o.no_trim else ("mask" if o.mask_adapter else "trim")    # trim/cli.py:103-111
    adapters2 = parser.parse_multi(o.adapters2, o.anywhere2, o.front2) if paired else []
    if (not adapters and not adapters2 and not qc and o.nextseq_trim is None and not o.cut and not o.cut2 and
            (o.minimum_length is None or o.minimum_length <= 0) and o.maximum_length is None and not o.trim_n and
            o.max_n is None):
        raise ValueError("You need to provide at least one adapter sequence.")        # trim/__init__.py:386-404
    if action == "mask" and any(isinstance(a, LinkedAdapter) for a in adapters + adapters2):
        raise NotImplementedError("--mask-adapter with a linked adapter: the reference's AdapterCutter fails on it "
                                  "(LinkedMatch has no astart, modifiers.py:158)")
    common = dict(times=o.times, action=action, nextseq_trim=o.nextseq_trim, quality_cutoff=qc,
                  quality_base=o.quality_base, trim_n=o.trim_n, minimum_length=o.minimum_length,
                  maximum_length=o.maximum_length, max_n=o.max_n, discard_trimmed=o.discard_trimmed,
                  discard_untrimmed=o.discard_untrimmed, op_order=o.op_order)
    bis1 = bis2 = None
    if o.bisulfite:                                                       # cli.py:702-739, trim/__init__.py:497-516
        kind = o.bisulfite
        if kind == "swift":
            if not paired:
                raise ValueError("Swift trimming is only compatible with paired-end reads")
            bis1, bis2 = [("min", 0, 10, False, False)], [("min", 10, 0, False, False)]
        elif kind == "non-directional":
            # (the reference's run fails at the end: NonDirectionalBisulfiteTrimmer.summarize reads the RRBS cutter it
            # only has with rrbs, modifiers.py:824-831)
            raise NotImplementedError("--bisulfite non-directional: the reference command fails on it")
        elif kind == "non-directional-rrbs":
            bis1 = bis2 = [("nondir", True)]
        elif kind == "rrbs":
            bis1 = bis2 = [("min", 0, 2, False, True)]
        elif kind in ("truseq", "epignome"):
            bis1 = bis2 = None                                            # (the reference adds no modifier for these)
        else:
            specs = []
            for arg in kind.split(";"):
                parts = [int(x) for x in arg.split(",")]
                if len(parts) != 4:
                    raise ValueError("Invalidate format for bisulfite parameters")
                # (cli.py:714-729: the flags index the tuple (False, True) -- 0 / 1, and -2 / -1 by Python's indexing; anything
                #  else is the parser's format error.  One negative length is accepted there and handed to MinCutter as
                #  it is: refused here rather than guessed at.)
                if any(f not in (0, 1, -1, -2) for f in parts[2:]):
                    raise ValueError("Invalidate format for bisulfite parameters")
                if parts[0] <= 0 and parts[1] <= 0:
                    specs.append(None)
                    continue
                if parts[0] < 0 or parts[1] < 0:
                    raise NotImplementedError("--bisulfite with a negative length next to a positive one")
                specs.append([("min", parts[0], parts[1], (False, True)[parts[2]], (False, True)[parts[3]])])
            if paired and len(specs) == 1:
                specs = [specs[0], specs[0]]
            elif not paired and len(specs) > 1:
                raise ValueError("Too many bisulfite parameters for single-end reads")
            bis1, bis2 = specs[0], (specs[1] if len(specs) > 1 else None)
    if legacy and (bis1 or bis2):
        raise NotImplementedError("--bisulfite with paired-end input in legacy mode")
    aux = {kind: path for kind, path in (("info", o.info_file), ("rest", o.rest_file), ("wildcard", o.wildcard_file)) if path}
    outputs = {kind: path for kind, path in (("too_short", o.too_short_output), ("too_long", o.too_long_output),
                                             ("untrimmed", o.untrimmed_output)) if path}
    if paired and outputs:
        second = dict(too_short=o.too_short_paired_output, too_long=o.too_long_paired_output, untrimmed=o.untrimmed_paired_output)
        if any(second[kind] is None for kind in outputs):
            raise NotImplementedError("paired-end input: --too-short-output / --too-long-output / --untrimmed-output need "
                                      "their --*-paired-output as well")
        outputs = {kind: (path, second[kind]) for kind, path in outputs.items()}
    if legacy and (aux or outputs or o.length_tag or o.strip_suffix or o.prefix or o.suffix or o.zero_cap):
        raise NotImplementedError("side files and read-name modifiers with paired-end input in legacy mode")
    if not paired:
        first = TrimPipeline(adapters=adapters, cut=o.cut, cut_min=o.cut_min, bisulfite=bis1, aux=aux or None, outputs=outputs or None, length_tag=o.length_tag,
                             strip_suffix=o.strip_suffix, prefix=o.prefix, suffix=o.suffix, zero_cap=o.zero_cap, **common)
        return LegacyPairedPipeline(first) if legacy else first
    return PairedTrimPipeline(outputs=outputs or None, aux=aux or None, length_tag=o.length_tag, strip_suffix=o.strip_suffix, prefix=o.prefix,
                              suffix=o.suffix, zero_cap=o.zero_cap, adapters1=adapters, adapters2=adapters2, aligner=o.aligner, cut=o.cut, cut2=o.cut2,
                              cut_min=o.cut_min, cut_min2=o.cut_min2, bisulfite=bis1, bisulfite2=bis2,
                              pair_filter=o.pair_filter or "any", insert_args=insert_args,
                              correct_mismatches=o.correct_mismatches, merge_overlapping=o.merge_overlapping,
                              merge_min_overlap=o.merge_min_overlap,
                              merge_error_rate=0.2 if o.merge_error_rate is None else o.merge_error_rate, **common)
