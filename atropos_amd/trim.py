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
from .fastq import FastqBatch, RecordSource

DEST_NAMES = {_lib.DEST_KEEP: "keep", _lib.DEST_TOO_SHORT: "too_short", _lib.DEST_TOO_LONG: "too_long",
              _lib.DEST_TOO_MANY_N: "too_many_n", _lib.DEST_TRIMMED: "trimmed", _lib.DEST_UNTRIMMED: "untrimmed"}


class TrimResult(object):
    """State of a batch after the pipeline: kept interval per read, the destination filter
    and the adapter flag; all device tensors."""

    def __init__(self, batch, begin, end, ubegin, uend, matched, dest):
        self.batch, self.begin, self.end, self.ubegin, self.uend = batch, begin, end, ubegin, uend
        self.matched, self.dest = matched, dest

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
                 discard_trimmed=False, discard_untrimmed=False, op_order="CGQAW"):
        self.adapters = list(adapters)
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
        be.match_trim_batch(best.contiguous(), front, int(codes[0].item()), begin, end, active, matched)

    def _round_linked(self, batch, begin, end, active, matched):
        """LinkedAdapter.match_to + trimmed (adapters/__init__.py:648-706) for linked adapters
        whose 5' parts are mutually exclusive: the first one whose 5' part matches is used."""
        be = batch.backend
        n = len(batch)
        claimed = torch.zeros((n,), dtype=torch.uint8, device=begin.device)
        for la in self.adapters:
            fsrc = RecordSource(batch, begin, end)
            frec = la.front_adapter.match_source(fsrc)
            has = ((frec[:, 1] >= 0) & (claimed == 0) & (active != 0)).to(torch.uint8)
            claimed |= has
            fb, fe = begin.clone(), end.clone()
            be.match_trim_batch(frec.contiguous(), None, 1, fb, fe, has.clone(), None)      # read[front.rstop:]
            brec = la.back_adapter.match_source(RecordSource(batch, fb, fe))
            be.match_trim_batch(brec.contiguous(), None, 0, fb, fe, has.clone(), None)      # then read[:back.rstart]
            sel = has != 0
            begin.copy_(torch.where(sel, fb, begin))
            end.copy_(torch.where(sel, fe, end))
        matched |= claimed
        active &= claimed                                                 # no 5' match: the loop over `times` stops

    def _adapter_stage(self, batch, begin, end):
        n = len(batch)
        dev = begin.device
        matched = torch.zeros((n,), dtype=torch.uint8, device=dev)
        if not self.adapters or n == 0:
            return matched, None, None
        before_b, before_e = begin.clone(), end.clone()
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
        return matched, ubegin, uend

    # ------------------------------------------------------------------ whole pipeline
    def run(self, batch):
        """All stages over one FastqBatch; returns a TrimResult."""
        be = batch.backend
        n = len(batch)
        begin = torch.zeros((n,), dtype=torch.int32, device=batch.records.device)
        end = batch.seq_lens.clone()
        matched = torch.zeros((n,), dtype=torch.uint8, device=begin.device)
        ubegin = uend = None
        for op in self.op_order:
            if op == "C" and (self.cut_front or self.cut_back):
                be.clip_batch(batch.records, begin, end, self.cut_front, self.cut_back)
            elif op == "G" and self.nextseq_trim is not None:
                be.quality_trim_batch(batch.data, batch.records, begin, end, 0, int(self.nextseq_trim),
                                      self.quality_base, True)
            elif op == "Q" and self.quality_cutoff:
                be.quality_trim_batch(batch.data, batch.records, begin, end, int(self.quality_cutoff[0]),
                                      int(self.quality_cutoff[1]), self.quality_base, False)
            elif op == "A" and self.adapters:
                matched, ubegin, uend = self._adapter_stage(batch, begin, end)
        if self.trim_n:
            be.nend_trim_batch(batch.data, batch.records, begin, end, ubegin, uend)
        min_len = self.minimum_length if self.minimum_length is not None and self.minimum_length > 0 else 0
        max_len = self.maximum_length if self.maximum_length is not None else -1
        max_n = float(self.max_n) if self.max_n is not None else -1.0
        dest = be.read_filter_batch(batch.data, batch.records, begin, end, ubegin, uend, matched, min_len, max_len,
                                    max_n, self.discard_trimmed, self.discard_untrimmed)
        return TrimResult(batch, begin, end, ubegin, uend, matched, dest)

    def trim_bytes(self, data, which=_lib.DEST_KEEP):
        """FASTQ text in, trimmed FASTQ text out (one batch)."""
        batch, _ = FastqBatch.from_bytes(data, final=True)
        return self.run(batch).text(which)

    def trim_file(self, path_in, path_out, chunk_bytes=256 << 20):
        """Stream a FASTQ file through the GPU in chunks of whole records; returns the
        destination counts.  (Plain files; compressed input is the caller's business.)"""
        totals = {name: 0 for name in DEST_NAMES.values()}
        strip = None
        carry = b""
        with open(path_in, "rb") as fin, open(path_out, "wb") as fout:
            while True:
                block = fin.read(chunk_bytes)
                final = len(block) < chunk_bytes
                buf = carry + block
                if strip is None and buf:
                    nl = buf.find(b"\n")
                    strip = 2 if (buf[:nl + 1] if nl >= 0 else buf).endswith(b"\r\n") else 1
                batch, consumed = FastqBatch.from_bytes(buf, final=final, strip=strip)
                carry = buf[consumed:] if not final else b""
                res = self.run(batch)
                fout.write(res.text(_lib.DEST_KEEP))
                for k, v in res.counts().items():
                    totals[k] += v
                if final:
                    break
        return totals


def pipeline_from_args(argv):
    """Build a TrimPipeline from the subset of ``atropos trim`` command-line options the device
    pipeline covers (same spellings and defaults as trim/cli.py:57-335, :655-803).  Anything
    else raises -- the caller then uses the per-read object path."""
    import argparse
    from .adapters import AdapterParser
    if isinstance(argv, str):
        argv = argv.split()
    ap = argparse.ArgumentParser(prog="atropos_amd.trim", add_help=False)
    ap.add_argument("-a", "--adapter", action="append", default=[], dest="adapters")
    ap.add_argument("-g", "--front", action="append", default=[])
    ap.add_argument("-b", "--anywhere", action="append", default=[])
    ap.add_argument("-e", "--error-rate", type=float, default=None)
    ap.add_argument("-O", "--overlap", type=int, default=None)
    ap.add_argument("-n", "--times", type=int, default=1)
    ap.add_argument("-N", "--no-match-adapter-wildcards", action="store_false", dest="match_adapter_wildcards",
                    default=True)
    ap.add_argument("--match-read-wildcards", action="store_true", default=False)
    ap.add_argument("--no-indels", action="store_false", dest="indels", default=True)
    ap.add_argument("--indel-cost", type=int, default=None)
    ap.add_argument("--no-trim", action="store_true", default=False)
    ap.add_argument("--mask-adapter", action="store_true", default=False)
    ap.add_argument("--op-order", default="CGQAW")
    ap.add_argument("-u", "--cut", type=int, action="append", default=[])
    ap.add_argument("-q", "--quality-cutoff", default=None)
    ap.add_argument("--quality-base", type=int, default=33)
    ap.add_argument("--nextseq-trim", type=int, default=None)
    ap.add_argument("--trim-n", action="store_true", default=False)
    ap.add_argument("-m", "--minimum-length", type=int, default=None)
    ap.add_argument("-M", "--maximum-length", type=int, default=None)
    ap.add_argument("--max-n", type=float, default=None)
    ap.add_argument("--discard-trimmed", "--discard", action="store_true", default=False)
    ap.add_argument("--discard-untrimmed", "--trimmed-only", action="store_true", default=False)
    o = ap.parse_args(argv)
    if o.error_rate is None:
        o.error_rate = 0.1                                                # cli.py:801-802
    if o.indels and o.indel_cost is None:
        o.indel_cost = 1                                                  # :660-661
    if o.overlap is None:
        o.overlap = 3                                                     # :662-666 (no --adapter-max-rmp here)
    if len(o.cut) > 2 or (len(o.cut) == 2 and o.cut[0] * o.cut[1] > 0):
        raise ValueError("You cannot remove bases from the same end twice.")
    qc = None
    if o.quality_cutoff is not None:
        qc = [int(x) for x in str(o.quality_cutoff).split(",")]
        if all(c <= 0 for c in qc):
            qc = None                                                     # :750-754
        elif len(qc) == 1:
            qc = [0] + qc
    kwargs = dict(max_error_rate=o.error_rate, min_overlap=o.overlap, read_wildcards=o.match_read_wildcards,
                  adapter_wildcards=o.match_adapter_wildcards, indels=o.indels)
    if o.indel_cost is not None:
        kwargs["indel_cost"] = o.indel_cost
    adapters = AdapterParser(**kwargs).parse_multi(o.adapters, o.anywhere, o.front)
    action = None if o.no_trim else ("mask" if o.mask_adapter else "trim")    # trim/cli.py:103-111
    return TrimPipeline(adapters=adapters, times=o.times, action=action, cut=o.cut, nextseq_trim=o.nextseq_trim,
                        quality_cutoff=qc, quality_base=o.quality_base, trim_n=o.trim_n,
                        minimum_length=o.minimum_length, maximum_length=o.maximum_length, max_n=o.max_n,
                        discard_trimmed=o.discard_trimmed, discard_untrimmed=o.discard_untrimmed,
                        op_order=o.op_order)
