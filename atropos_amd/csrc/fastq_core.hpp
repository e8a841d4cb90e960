// fastq_core.hpp -- per-record arithmetic of the device-resident FASTQ batch: record
// indexing/validation, quality trimming, N-end trimming, read filters.
//
// The reference reads FASTQ into one Python `Sequence` object per record
// (atropos/io/_seqio.pyx:163-245), runs each modifier per read
// (commands/trim/modifiers.py) and formats each surviving read back to text
// (io/seqio.py:686-700).  Here a batch is the raw file bytes in HBM plus one 32-byte
// record descriptor per read and a kept interval [begin, end) of its sequence line;
// every trimming modifier of the single-end path only ever clips the ends, so it is an
// interval update, and the writer slices the original bytes.
//
// Compiled for gfx950 and, with -DATR_HOST_EMU, for the CPU test emulation (tests/emu).
#ifndef ATR_FASTQ_CORE_HPP
#define ATR_FASTQ_CORE_HPP

#include <stdint.h>
#include <string.h>
#include "atropos_hip.h"

#ifdef ATR_HOST_EMU
#ifndef ATR_DEV
#define ATR_DEV static inline
#endif
#else
#ifndef ATR_DEV
#define ATR_DEV __device__ __forceinline__
#endif
#endif

namespace atr {

// Descriptor of one record; same layout as atr_fastq_record.
struct FastqRecord {
    uint32_t name_off, name_len;                     // the description after '@' (line end stripped)
    uint32_t seq_off, seq_len;
    uint32_t qual_off, qual_len;
    uint32_t flags;                                  // bit 0: the '+' line repeats the description (name2 = name)
                                                     // bit 1: ... and that text sits at `reserved`, flags >> 8 bytes of it
                                                     //   (records whose name a read-name modifier rewrote: name2 keeps the
                                                     //   text the file had, io/seqio.py:690-699)
    uint32_t reserved;
};
ATR_DEV uint32_t fastq_name2_len(const FastqRecord &rec) { return (rec.flags & 1u) ? ((rec.flags & 2u) ? rec.flags >> 8 : rec.name_len) : 0u; }
ATR_DEV uint32_t fastq_name2_off(const FastqRecord &rec) { return (rec.flags & 2u) ? rec.reserved : rec.name_off; }

// Line terminators follow Python's universal-newline text mode, which is how the reference
// reads its input (xopen(..., 'r')): "\n", "\r\n" and a lone "\r" each end a line.  Line L
// (0-based) is the bytes [start, end] where end = line_ends[L] is the position of the LAST byte
// of its terminator and start = L ? line_ends[L-1] + 1 : 0.
ATR_DEV bool is_line_end(uint8_t c, uint8_t next) { return c == '\n' || (c == '\r' && next != '\n'); }

// content length of the line [start, end]: without its terminator ("\r\n" counts as one).
// A '\r' in front of the terminator's last byte can only be the first half of "\r\n" (a lone
// one would have ended the line itself).  has_cr: whether the chunk holds any '\r' at all.
ATR_DEV uint32_t line_content(const uint8_t *bytes, uint32_t start, uint32_t end, bool has_cr) {
    uint32_t len = end - start;                                      // drop the terminator's last byte
    if (has_cr && len > 0 && bytes[end - 1] == '\r') --len;
    return len;
}

// Returns 0 or an ATR_FASTQ_ERR_* code (the record is still written, for the message).
// (After newline translation every line the reference sees ends in "\n", so its `strip` is -1.)
ATR_DEV int fastq_record_one(const uint8_t *bytes, const uint32_t *line_ends, long long r, bool has_cr, FastqRecord &rec) {
    const long long L = 4 * r;
    const uint32_t s0 = L ? line_ends[L - 1] + 1 : 0u;
    const uint32_t e0 = line_ends[L], e1 = line_ends[L + 1], e2 = line_ends[L + 2], e3 = line_ends[L + 3];
    const uint32_t s1 = e0 + 1, s2 = e1 + 1, s3 = e2 + 1;
    const uint32_t c0 = line_content(bytes, s0, e0, has_cr), c1 = line_content(bytes, s1, e1, has_cr),
                   c2 = line_content(bytes, s2, e2, has_cr), c3 = line_content(bytes, s3, e3, has_cr);
    int err = 0;
    // line 1: '@' + name = line[1:-1]                               (_seqio.pyx:208-216)
    if (c0 == 0 || bytes[s0] != '@') err = ATR_FASTQ_ERR_AT;
    rec.name_off = s0 + 1;
    rec.name_len = c0 > 0 ? c0 - 1 : 0u;
    // line 2: sequence = line[:-1]                                  (:217-218)
    rec.seq_off = s1;
    rec.seq_len = c1;
    // line 3: '+' or '+' + name                                     (:219-238)
    rec.flags = 0;
    if (c2 == 0 || bytes[s2] != '+') {
        if (!err) err = ATR_FASTQ_ERR_PLUS;
    } else if (c2 > 1) {
        bool same = (c2 - 1 == rec.name_len);
        for (uint32_t i = 0; same && i < rec.name_len; ++i) same = bytes[s2 + 1 + i] == bytes[rec.name_off + i];
        if (!same) { if (!err) err = ATR_FASTQ_ERR_NAME2; }
        else rec.flags = 1;
    }
    // line 4: qualities                                             (:239-243)
    rec.qual_off = s3;
    rec.qual_len = c3;
    if (rec.qual_len != rec.seq_len && !err) err = ATR_FASTQ_ERR_LENGTH;                      // _seqio.pyx:33-43
    rec.reserved = 0;
    return err;
}

// 16 bytes at the 16-byte aligned offset blk of the chunk, as four little-endian dwords
ATR_DEV void load_block16(const uint8_t *bytes, uint32_t blk, uint32_t w[4]) {
#ifdef ATR_HOST_EMU
    memcpy(w, bytes + blk, 16);
#else
    const uint4 v = *(const uint4 *)(bytes + blk);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
#endif
}

// quality_trim_index on the qualities at bytes[qoff : qoff + len]
// (commands/trim/_qualtrim.pyx:7-48).  `bytes` is the 16-byte aligned chunk: the scans fetch
// the quality line one aligned 16-byte block at a time instead of one dependent byte load per
// base (both scans stop at the first negative running sum, typically within a block or two).
ATR_DEV void quality_trim_one(const uint8_t *bytes, uint32_t qoff, int len, int cutoff_front, int cutoff_back, int base,
                              int &start, int &stop) {
    start = 0;
    stop = len;
    int s = 0, max_qual = 0, i = 0;
    bool done = false;
    while (i < len && !done) {                                    // 5' end: i ascending
        const uint32_t addr = qoff + (uint32_t)i, blk = addr & ~15u;
        uint32_t w[4];
        load_block16(bytes, blk, w);
        const int k0 = (int)(addr - blk);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k >= k0 && i < len && !done) {
                const int q = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                s += cutoff_front - (q - base);
                if (s < 0) done = true;
                else {
                    if (s > max_qual) { max_qual = s; start = i + 1; }
                    ++i;
                }
            }
        }
    }
    max_qual = 0;
    s = 0;
    i = len - 1;
    done = false;
    while (i >= 0 && !done) {                                     // 3' end: i descending
        const uint32_t addr = qoff + (uint32_t)i, blk = addr & ~15u;
        uint32_t w[4];
        load_block16(bytes, blk, w);
        const int k0 = (int)(addr - blk);
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            if (k <= k0 && i >= 0 && !done) {
                const int q = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                s += cutoff_back - (q - base);
                if (s < 0) done = true;
                else {
                    if (s > max_qual) { max_qual = s; stop = i; }
                    --i;
                }
            }
        }
    }
    if (start >= stop) { start = 0; stop = 0; }
}

// nextseq_trim_index  (_qualtrim.pyx:51-84): as the 3' part above, but a 'G' counts as
// quality cutoff - 1.
ATR_DEV int nextseq_trim_one(const uint8_t *seq, const uint8_t *qual, int len, int cutoff, int base) {
    int s = 0, max_qual = 0, max_i = len;
    for (int i = len - 1; i >= 0; --i) {
        int q = (int)qual[i] - base;
        if (seq[i] == 'G') q = cutoff - 1;
        s += cutoff - q;
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; max_i = i; }
    }
    return max_i;
}

// Base i of the current read seq[0:len) when AdapterCutter's action is 'mask': positions
// outside [ub, ue) were replaced by 'N' (modifiers.py:155-172).  ub = 0, ue = len: no mask.
ATR_DEV uint8_t masked_base(const uint8_t *seq, int i, int ub, int ue) { return (i >= ub && i < ue) ? seq[i] : (uint8_t)'N'; }

// NEndTrimmer (modifiers.py:766-784): regexes ^N+ and N+$ on the sequence -- upper-case N only.
ATR_DEV void nend_trim_one(const uint8_t *seq, int len, int ub, int ue, int &start, int &stop) {
    int a = 0;
    while (a < len && masked_base(seq, a, ub, ue) == 'N') ++a;
    // `N+$` is searched on the whole string: for an all-N read both regexes match everything
    int b = len;
    while (b > 0 && masked_base(seq, b - 1, ub, ue) == 'N') --b;
    start = a;
    stop = b;
}

// Python's read[begin:end] on a read of length len, end may be "None" (= len) -- slice
// clamping for a non-negative begin and any end (negative counts from the end).
ATR_DEV void py_clip(int len, int begin, int end_or_neg, bool has_end, int &a, int &b) {
    a = begin > len ? len : begin;
    b = len;
    if (has_end) {
        int e = end_or_neg;
        if (e < 0) e += len;
        if (e < 0) e = 0;
        if (e > len) e = len;
        b = e;
    }
    if (b < a) b = a;
}

// Which filters fire for a read after all modifiers (commands/trim/filters.py:109-184):
// bit d of the result is set when the filter with destination code d (ATR_DEST_*) fires.
ATR_DEV uint32_t read_filter_mask(const uint8_t *seq, int len, int ub, int ue, bool matched, int min_len, int max_len,
                                  double max_n, int discard_trimmed, int discard_untrimmed) {
    uint32_t mask = 0;
    if (min_len > 0 && len < min_len) mask |= 1u << ATR_DEST_TOO_SHORT;
    if (max_len >= 0 && len > max_len) mask |= 1u << ATR_DEST_TOO_LONG;
    if (max_n >= 0.0) {
        int n_count = 0;
        for (int i = 0; i < len; ++i) {
            const uint8_t c = masked_base(seq, i, ub, ue);
            n_count += (c == 'N' || c == 'n') ? 1 : 0;
        }
        if (max_n < 1.0) {
            if (len != 0 && (double)n_count / (double)len > max_n) mask |= 1u << ATR_DEST_TOO_MANY_N;
        } else if ((double)n_count > max_n) {
            mask |= 1u << ATR_DEST_TOO_MANY_N;
        }
    }
    if (discard_trimmed && matched) mask |= 1u << ATR_DEST_TRIMMED;
    if (discard_untrimmed && !matched) mask |= 1u << ATR_DEST_UNTRIMMED;
    return mask;
}

// The destination: the first filter that fires, in the order trim/__init__.py:566-601 installs
// them (too short, too long, too many N, discard-trimmed, then discard-untrimmed).  For pairs
// (PairedWrapper, filters.py:66-90) a filter fires when it fires for at least min_affected of
// the two reads.
ATR_DEV int filter_destination(uint32_t mask1, uint32_t mask2, bool paired, int min_affected) {
    for (int d = ATR_DEST_TOO_SHORT; d <= ATR_DEST_UNTRIMMED; ++d) {
        const int f = (int)((mask1 >> d) & 1u) + (paired ? (int)((mask2 >> d) & 1u) : 0);
        if (f >= (paired ? min_affected : 1)) return d;
    }
    return ATR_DEST_KEEP;
}

ATR_DEV int read_filter_one(const uint8_t *seq, int len, int ub, int ue, bool matched, int min_len, int max_len,
                            double max_n, int discard_trimmed, int discard_untrimmed) {
    return filter_destination(read_filter_mask(seq, len, ub, ue, matched, min_len, max_len, max_n, discard_trimmed,
                                               discard_untrimmed), 0u, false, 1);
}

// InsertAdapterCutter.__call__ after the alignments (commands/trim/modifiers.py:391-496), in
// two steps around the optional error correction:
//   plan: pick the adapter matches (from the insert match, else the two adapters' own
//         semi-global matches), mirror a lone match onto the other read (`symmetric`, :419-446)
//         and decide whether -- and over which overlap -- errors are to be corrected (:397-446);
//   trim: `trim` (:455-496) with the read lengths as they are AFTER the correction.
// ins: the three records of atr_insert_match_batch for this pair; fb1/fb2: Adapter.match_to
// records of the two reads; len1/len2: current read lengths.
struct InsertPlan {
    bool active;                  // both reads at least min_insert_len long (:392-394)
    bool h1, h2;                  // adapter match present for read 1 / read 2
    int rstart1, rstart2;
    bool correct;                 // run correct_errors with corr[] as the insert match
    int16_t corr[4];
};

ATR_DEV void insert_plan_matches(const int16_t *ins, const int16_t *fb1, const int16_t *fb2, int len1, int len2,
                                 int min_insert_len, int symmetric, bool mismatch_action, InsertPlan &P) {
    P.active = !(len1 < min_insert_len || len2 < min_insert_len);
    P.h1 = P.h2 = P.correct = false;
    P.rstart1 = P.rstart2 = 0;
    P.corr[0] = P.corr[1] = P.corr[2] = P.corr[3] = 0;
    if (!P.active) return;
    const bool has_insert = ins[1] >= 0;
    bool have_tuple = false;
    const int16_t *m1 = has_insert ? ins + 8 : fb1, *m2 = has_insert ? ins + 16 : fb2;
    P.h1 = m1[1] >= 0; P.h2 = m2[1] >= 0;
    P.rstart1 = m1[2]; P.rstart2 = m2[2];
    if (has_insert) {                                                     // :397-403
        have_tuple = true;
        P.corr[0] = ins[0]; P.corr[1] = ins[1]; P.corr[2] = ins[2]; P.corr[3] = ins[3];
        P.correct = mismatch_action && ins[5] > 0;
    } else if (mismatch_action && P.h1 && P.h2 && P.rstart1 == P.rstart2) {   // :408-415: complementary adapter matches
        have_tuple = true;
        P.corr[0] = (int16_t)(len2 - P.rstart1); P.corr[1] = (int16_t)len2; P.corr[2] = 0; P.corr[3] = (int16_t)P.rstart1;
        P.correct = true;
    }
    if (symmetric && (P.h1 != P.h2)) {                                    // :419-446
        if (P.h1) { if (P.rstart1 <= len2) { P.h2 = true; P.rstart2 = P.rstart1; } }
        else if (P.rstart2 <= len1) { P.h1 = true; P.rstart1 = P.rstart2; }
        if (mismatch_action && !have_tuple && P.h1 && P.h2) {
            P.corr[0] = (int16_t)(len2 - P.rstart1); P.corr[1] = (int16_t)len2; P.corr[2] = 0; P.corr[3] = (int16_t)P.rstart1;
            P.correct = true;
        }
    }
}

// Returns the new lengths in cut1/cut2 (== len: untouched) and the `read.match is not None` flags.
ATR_DEV void insert_plan_trim(const InsertPlan &P, int len1, int len2, int trim_action, int &cut1, int &cut2,
                              bool &matched1, bool &matched2) {
    cut1 = len1; cut2 = len2; matched1 = matched2 = false;
    if (!P.active) return;
    if (P.h1) { matched1 = true; if (trim_action && P.rstart1 < len1) cut1 = P.rstart1; }   // rstart >= len: no trim
    if (P.h2) { matched2 = true; if (trim_action && P.rstart2 < len2) cut2 = P.rstart2; }
}

// Bytes of a formatted record (io/seqio.py:690-699): '@' name '\n' seq '\n+' name2 '\n' qual '\n'
ATR_DEV uint32_t fastq_record_bytes(const FastqRecord &rec, int kept) {
    return 1u + rec.name_len + 1u + (uint32_t)kept + 2u + fastq_name2_len(rec) + 1u + (uint32_t)kept + 1u;
}

// ---- MergeOverlapping (commands/trim/modifiers.py:864-931) ---------------------------------
// al = the record of Aligner(reverse_complement(read2), error_rate, flags).locate(read1):
// (r2_start, r2_stop, r1_start, r1_stop, matches, errors), refstop -1 = None (or a pair that was
// too short to be aligned at all).  "mate" = reverse_complement(read2) as it was BEFORE any
// correction (:887); the merged read is one or two segments, each of read 1 or of the mate.
enum { MERGE_NONE = 0,         // no (sufficient) alignment: the pair stays a pair
       MERGE_R2_INSIDE = 1,    // read 2 lies inside read 1: read 1 as it is               (:904-906)
       MERGE_R1_INSIDE = 2,    // read 1 lies inside read 2: the mate                      (:907-910)
       MERGE_APPEND = 3,       // read1 + mate[r2_stop:]                                   (:911-915)
       MERGE_PREPEND = 4,      // mate + read1[r1_stop:]                                   (:916-921)
       MERGE_INVALID = 5 };    // "Invalid alignment while trying to merge read"          (:922-926)

struct MergeShape {
    int kind;
    int mate_first;            // the first segment is the mate's (kinds 2 and 4)
    int from[2], len[2];       // segment s: bases [from, from + len) of its source
};

ATR_DEV MergeShape merge_shape(const int16_t *al, int len1, int len2, int need) {
    MergeShape m;
    m.kind = MERGE_NONE; m.mate_first = 0;
    m.from[0] = m.from[1] = m.len[0] = m.len[1] = 0;
    if (al[1] < 0 || (int)al[4] < need) return m;                       // :898-899
    const int r2_start = al[0], r2_stop = al[1], r1_start = al[2], r1_stop = al[3];
    if (r2_start == 0 && r2_stop == len2) { m.kind = MERGE_R2_INSIDE; m.len[0] = len1; }
    else if (r1_start == 0 && r1_stop == len1) { m.kind = MERGE_R1_INSIDE; m.mate_first = 1; m.len[0] = len2; }
    else if (r1_start > 0) { m.kind = MERGE_APPEND; m.len[0] = len1; m.from[1] = r2_stop; m.len[1] = len2 - r2_stop; }
    else if (r2_start > 0) { m.kind = MERGE_PREPEND; m.mate_first = 1; m.len[0] = len2; m.from[1] = r1_stop; m.len[1] = len1 - r1_stop; }
    else m.kind = MERGE_INVALID;
    if (m.len[1] < 0) m.len[1] = 0;
    return m;
}

// The merged record's text, written by `nl` cooperating lanes (lane = 0 .. nl-1; the emulation
// passes 0, 1).  mate_pass: only the mate's BASES (to be run before the pair is corrected: the
// reference takes the mate before correct_errors and its qualities after, :887 / :909 / :914 / :919);
// otherwise everything else -- name, read 1's bases, the '+' line, all qualities.
ATR_DEV void merge_emit_one(uint8_t *o, const MergeShape &m, const FastqRecord &rec1, const uint8_t *bytes1, int a1,
                            const FastqRecord &rec2, const uint8_t *bytes2, int a2, int len2, const uint8_t *comp,
                            bool mate_pass, int lane, int nl) {
    const uint32_t total = (uint32_t)(m.len[0] + m.len[1]);
    const uint8_t *s1 = bytes1 + rec1.seq_off + a1, *q1 = bytes1 + rec1.qual_off + a1;
    const uint8_t *s2 = bytes2 + rec2.seq_off + a2, *q2 = bytes2 + rec2.qual_off + a2;
    uint8_t *seq = o + 1 + rec1.name_len + 1;
    uint8_t *plus = seq + total;
    uint8_t *qual = plus + 2 + ((rec1.flags & 1u) ? rec1.name_len : 0u) + 1;
    for (int s = 0; s < 2; ++s) {
        const bool mate = (s == 0) == (m.mate_first != 0);
        const int base = s == 0 ? 0 : m.len[0];
        if (mate) {
            for (int k = lane; k < m.len[s]; k += nl) {
                const int src = len2 - 1 - (m.from[s] + k);
                if (mate_pass) seq[base + k] = comp[s2[src]];
                else qual[base + k] = q2[src];
            }
        } else if (!mate_pass) {
            for (int k = lane; k < m.len[s]; k += nl) {
                seq[base + k] = s1[m.from[s] + k];
                qual[base + k] = q1[m.from[s] + k];
            }
        }
    }
    if (mate_pass) return;
    for (uint32_t k = (uint32_t)lane; k < rec1.name_len; k += (uint32_t)nl) {
        o[1 + k] = bytes1[rec1.name_off + k];
        if (rec1.flags & 1u) plus[2 + k] = bytes1[rec1.name_off + k];
    }
    if (lane == 0) {
        o[0] = '@';
        o[1 + rec1.name_len] = '\n';
        plus[0] = '\n'; plus[1] = '+';
        qual[-1] = '\n';
        qual[total] = '\n';
    }
}

}  // namespace atr
#endif
