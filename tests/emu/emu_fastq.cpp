// TEST INFRASTRUCTURE: CPU twin of atropos_amd/csrc/fastq_kernels.hip, built from the same
// per-record source (fastq_core.hpp, locate_core.hpp::pack_word) with -DATR_HOST_EMU so that
// the CPU test-suite can check the FASTQ batch logic without a GPU.  One loop iteration
// here = one thread (or one wave iteration) of the kernel of the same name.
#include <stdint.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "atropos_hip.h"
#include "locate_core.hpp"
#include "fastq_core.hpp"
#include "misc_core.hpp"

using namespace atr;

extern "C" {

int emu_fastq_count_lines(const uint8_t *bytes, int64_t nbytes, int64_t *nlines) {
    if (nbytes < 0 || nbytes >= (int64_t)0xFFFFFFF0ll || !nlines) return ATR_ERR_INVALID;
    int64_t c = 0;
    for (int64_t i = 0; i < nbytes; ++i) c += is_line_end(bytes[i], i + 1 < nbytes ? bytes[i + 1] : 0) ? 1 : 0;
    *nlines = c;
    return ATR_OK;
}

int emu_fastq_index(const uint8_t *bytes, int64_t nbytes, uint32_t *line_ends, int64_t nlines,
                    atr_fastq_record *records, int64_t *error) {
    if (nbytes < 0 || nlines < 0 || !error) return ATR_ERR_INVALID;
    *error = LLONG_MAX;
    int64_t k = 0;
    for (int64_t i = 0; i < nbytes && k < nlines; ++i)
        if (is_line_end(bytes[i], i + 1 < nbytes ? bytes[i + 1] : 0)) line_ends[k++] = (uint32_t)i;
    for (int64_t r = 0; r < nlines / 4; ++r) {
        FastqRecord rec;
        const int err = fastq_record_one(bytes, line_ends, r, true, rec);
        memcpy(&records[r], &rec, sizeof(rec));
        if (err && r * 8 + err < *error) *error = r * 8 + err;
    }
    return ATR_OK;
}

int emu_pack_records(const uint8_t *bytes, const atr_fastq_record *records, const int32_t *begin, const int32_t *end,
                     int64_t nreads, int max_len, const uint8_t table[256], int planes, uint8_t *packed, int32_t *lens,
                     int32_t *invalid) {
    if (nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || !table) return ATR_ERR_INVALID;
    const int nchunks = (max_len + 31) / 32;
    uint32_t *out = (uint32_t *)packed;
    uint32_t spread[256];
    for (int c = 0; c < 256; ++c) spread[c] = spread_code((uint32_t)table[c] & 15u);
    for (int64_t r = 0; r < nreads; ++r) {
        const FastqRecord &rec = *(const FastqRecord *)&records[r];
        int a = begin ? begin[r] : 0, b = end ? end[r] : (int)rec.seq_len;
        a = a < 0 ? 0 : (a > (int)rec.seq_len ? (int)rec.seq_len : a);
        b = b > (int)rec.seq_len ? (int)rec.seq_len : b;
        b = b < a ? a : b;
        const int n = b - a < max_len ? b - a : max_len;
        if (lens) lens[r] = n;
        const uint8_t *row = bytes + rec.seq_off + a;
        const int64_t tile = r >> 6;
        const int lane = (int)(r & 63);
        bool zero_seen = false;
        for (int c = 0; c < nchunks; ++c) {
            uint32_t *dst = out + (((size_t)tile * nchunks + c) * 64 + lane) * 4;
            if (planes) pack_planes_chunk(row, c * 32, n, spread, zero_seen, dst);
            else
                for (int d = 0; d < 4; ++d) dst[d] = pack_word(row, c * 32 + d * 8, n, table, zero_seen);
        }
        if (invalid && zero_seen) *invalid += 1;
    }
    return ATR_OK;
}

int emu_clip_batch(const atr_fastq_record *, int32_t *begin, int32_t *end, int64_t n, int front, int back) {
    if (n < 0 || front < 0 || back > 0) return ATR_ERR_INVALID;
    for (int64_t r = 0; r < n; ++r) {
        const int a = begin[r], b = end[r];
        if (b <= a || (front == 0 && back == 0)) continue;
        int na, nb;
        py_clip(b - a, front, back, back < 0, na, nb);
        begin[r] = a + na;
        end[r] = a + nb;
    }
    return ATR_OK;
}

int emu_quality_trim_batch(const uint8_t *bytes, const atr_fastq_record *records, int32_t *begin, int32_t *end,
                           int64_t n, int cutoff_front, int cutoff_back, int base, int nextseq) {
    for (int64_t r = 0; r < n; ++r) {
        const FastqRecord &rec = *(const FastqRecord *)&records[r];
        const int a = begin[r], b = end[r];
        if (b <= a) continue;
        const uint8_t *qual = bytes + rec.qual_off + a;
        if (nextseq) {
            end[r] = a + nextseq_trim_one(bytes + rec.seq_off + a, qual, b - a, cutoff_back, base);
        } else {
            int s, e;
            quality_trim_one(bytes, rec.qual_off + (uint32_t)a, b - a, cutoff_front, cutoff_back, base, s, e);
            begin[r] = a + s;
            end[r] = a + e;
        }
    }
    return ATR_OK;
}

int emu_nend_trim_batch(const uint8_t *bytes, const atr_fastq_record *records, int32_t *begin, int32_t *end,
                        const int32_t *ubegin, const int32_t *uend, int64_t n) {
    for (int64_t r = 0; r < n; ++r) {
        const FastqRecord &rec = *(const FastqRecord *)&records[r];
        const int a = begin[r], b = end[r];
        if (b <= a) continue;
        const int ub = ubegin ? ubegin[r] - a : 0, ue = uend ? uend[r] - a : b - a;
        int s, e;
        nend_trim_one(bytes + rec.seq_off + a, b - a, ub, ue, s, e);
        begin[r] = a + s;
        end[r] = a + (e < s ? s : e);
    }
    return ATR_OK;
}

int emu_match_trim_batch(const int16_t *matches, const uint8_t *front, int default_front, int32_t *begin, int32_t *end,
                         uint8_t *active, uint8_t *matched, int64_t n) {
    for (int64_t r = 0; r < n; ++r) {
        if (active && !active[r]) continue;
        const int16_t *m = matches + 8 * r;
        if (m[1] < 0) { if (active) active[r] = 0; continue; }
        const int rstart = m[2], rstop = m[3];
        int f = front ? (int)front[r] : default_front;
        if (f > 1) f = rstart == 0 ? 1 : 0;
        const int a = begin[r], b = end[r];
        if (f) begin[r] = a + rstop < b ? a + rstop : b;
        else { int e = a + rstart < b ? a + rstart : b; end[r] = e > a ? e : a; }
        if (matched) matched[r] = 1;
    }
    return ATR_OK;
}

int emu_read_filter_batch(const uint8_t *bytes, const atr_fastq_record *records, const int32_t *begin,
                          const int32_t *end, const int32_t *ubegin, const int32_t *uend, const uint8_t *matched,
                          int64_t n, int min_len, int max_len, double max_n, int discard_trimmed, int discard_untrimmed,
                          uint8_t *dest, uint8_t *fail_mask) {
    for (int64_t r = 0; r < n; ++r) {
        const FastqRecord &rec = *(const FastqRecord *)&records[r];
        const int a = begin[r], b = end[r] > a ? end[r] : a;
        const int ub = ubegin ? ubegin[r] - a : 0, ue = uend ? uend[r] - a : b - a;
        const uint32_t mask = read_filter_mask(bytes + rec.seq_off + a, b - a, ub, ue, matched ? matched[r] != 0 : false,
                                               min_len, max_len, max_n, discard_trimmed, discard_untrimmed);
        if (dest) dest[r] = (uint8_t)filter_destination(mask, 0u, false, 1);
        if (fail_mask) fail_mask[r] = (uint8_t)mask;
    }
    return ATR_OK;
}

int emu_pair_filter_batch(const uint8_t *mask1, const uint8_t *mask2, int64_t n, int min_affected, uint8_t *dest) {
    if (min_affected != 1 && min_affected != 2) return ATR_ERR_INVALID;
    for (int64_t r = 0; r < n; ++r) dest[r] = (uint8_t)filter_destination(mask1[r], mask2[r], true, min_affected);
    return ATR_OK;
}

int emu_insert_plan_batch(const int16_t *ins, const int16_t *fb1, const int16_t *fb2, uint8_t *bytes1,
                          const atr_fastq_record *records1, uint8_t *bytes2, const atr_fastq_record *records2,
                          int32_t *begin1, int32_t *end1, int32_t *begin2, int32_t *end2, int32_t *uend1, int32_t *uend2,
                          int64_t n, int min_insert_len, int symmetric, int trim_action, int correct_action,
                          int min_qual_diff, const uint8_t *comp, uint8_t *matched1, uint8_t *matched2,
                          int32_t *corrected, int64_t *error) {
    if (error) *error = LLONG_MAX;
    for (int64_t r = 0; r < n; ++r) {
        const int a1 = begin1[r], a2 = begin2[r];
        int len1 = end1[r] > a1 ? end1[r] - a1 : 0, len2 = end2[r] > a2 ? end2[r] - a2 : 0;
        InsertPlan P;
        insert_plan_matches(ins + 24 * r, fb1 + 8 * r, fb2 + 8 * r, len1, len2, min_insert_len, symmetric,
                            correct_action >= 0, P);
        if (corrected) corrected[2 * r] = corrected[2 * r + 1] = 0;
        if (P.correct) {
            const FastqRecord &r1 = *(const FastqRecord *)&records1[r], &r2 = *(const FastqRecord *)&records2[r];
            int32_t changed[2], newlen[2];
            correct_errors_one(bytes1 + r1.seq_off + a1, bytes1 + r1.qual_off + a1, len1, bytes2 + r2.seq_off + a2,
                               bytes2 + r2.qual_off + a2, len2, P.corr, correct_action, min_qual_diff, true, comp, changed,
                               newlen);
            if (changed[0] < 0) {
                if (r * 8 - changed[0] < *error) *error = r * 8 - changed[0];
            } else {
                len1 = newlen[0]; len2 = newlen[1];
                end1[r] = a1 + len1; end2[r] = a2 + len2;
                if (corrected) { corrected[2 * r] = changed[0]; corrected[2 * r + 1] = changed[1]; }
            }
        }
        int cut1, cut2;
        bool m1, m2;
        insert_plan_trim(P, len1, len2, trim_action, cut1, cut2, m1, m2);
        if (trim_action == 2) { uend1[r] = a1 + cut1; uend2[r] = a2 + cut2; }
        else { end1[r] = a1 + cut1; end2[r] = a2 + cut2; }
        matched1[r] = m1 ? 1 : 0;
        matched2[r] = m2 ? 1 : 0;
    }
    return ATR_OK;
}

// atr_merge_plan_batch + atr_merge_emit_batch in one call: out == NULL sizes the output
// (offsets[n + 1], kind[n]); with out the text is written (mate bases, correction, the rest).
int emu_merge_batch(const int16_t *align, const int32_t *need, const uint8_t *insert_matched, uint8_t *bytes1,
                    const atr_fastq_record *records1, uint8_t *bytes2, const atr_fastq_record *records2,
                    const int32_t *begin1, const int32_t *end1, const int32_t *begin2, const int32_t *end2, int64_t n,
                    int correct_action, int min_qual_diff, const uint8_t *comp, uint8_t *kind, int64_t *offsets,
                    int32_t *corrected, int64_t *error, uint8_t *out) {
    if (!out) {
        *error = LLONG_MAX;
        int64_t at = 0;
        for (int64_t r = 0; r < n; ++r) {
            const int len1 = std::max(0, end1[r] - begin1[r]), len2 = std::max(0, end2[r] - begin2[r]);
            const MergeShape m = merge_shape(align + 8 * r, len1, len2, need[r]);
            kind[r] = (uint8_t)m.kind;
            offsets[r] = at;
            if (m.kind == MERGE_INVALID) *error = std::min<int64_t>(*error, r * 8 + 4);
            else if (m.kind != MERGE_NONE) at += fastq_record_bytes(*(const FastqRecord *)&records1[r], m.len[0] + m.len[1]);
        }
        offsets[n] = at;
        return ATR_OK;
    }
    for (int pass = 0; pass < 3; ++pass)
        for (int64_t r = 0; r < n; ++r) {
            if (kind[r] == MERGE_NONE || kind[r] == MERGE_INVALID) continue;
            const FastqRecord &r1 = *(const FastqRecord *)&records1[r], &r2 = *(const FastqRecord *)&records2[r];
            const int a1 = begin1[r], a2 = begin2[r];
            const int len1 = std::max(0, end1[r] - a1), len2 = std::max(0, end2[r] - a2);
            if (pass == 1) {
                if (corrected) corrected[2 * r] = corrected[2 * r + 1] = 0;
                if (correct_action < 0 || align[8 * r + 5] <= 0 || (insert_matched && insert_matched[r])) continue;
                int32_t changed[2], newlen[2];
                correct_errors_one(bytes1 + r1.seq_off + a1, bytes1 + r1.qual_off + a1, len1, bytes2 + r2.seq_off + a2,
                                   bytes2 + r2.qual_off + a2, len2, align + 8 * r, correct_action, min_qual_diff, false, comp,
                                   changed, newlen);
                if (changed[0] < 0) *error = std::min<int64_t>(*error, r * 8 - changed[0]);
                else if (corrected) { corrected[2 * r] = changed[0]; corrected[2 * r + 1] = changed[1]; }
                continue;
            }
            const MergeShape m = merge_shape(align + 8 * r, len1, len2, 0);
            merge_emit_one(out + offsets[r], m, r1, bytes1, a1, r2, bytes2, a2, len2, comp, pass == 0, 0, 1);
        }
    return ATR_OK;
}

int emu_fastq_emit(const uint8_t *bytes, const atr_fastq_record *records, const int32_t *begin, const int32_t *end,
                   const int32_t *ubegin, const int32_t *uend, const uint8_t *dest, int which, int64_t n,
                   int64_t *offsets, uint8_t *out) {
    if (!out) {
        int64_t run = 0;
        for (int64_t r = 0; r < n; ++r) {
            offsets[r] = run;
            if (!dest || dest[r] == which) {
                const int kept = end[r] > begin[r] ? end[r] - begin[r] : 0;
                run += fastq_record_bytes(*(const FastqRecord *)&records[r], kept);
            }
        }
        offsets[n] = run;
        return ATR_OK;
    }
    for (int64_t r = 0; r < n; ++r) {
        if (dest && dest[r] != which) continue;
        const FastqRecord &rec = *(const FastqRecord *)&records[r];
        const int a = begin[r], b = end[r] > a ? end[r] : a;
        const uint32_t kept = (uint32_t)(b - a);
        uint8_t *o = out + offsets[r];
        *o++ = '@';
        memcpy(o, bytes + rec.name_off, rec.name_len); o += rec.name_len;
        *o++ = '\n';
        for (uint32_t k = 0; k < kept; ++k) {
            const int pos = a + (int)k;
            o[k] = (!ubegin || (pos >= ubegin[r] && pos < uend[r])) ? bytes[rec.seq_off + pos] : (uint8_t)'N';
        }
        o += kept;
        *o++ = '\n'; *o++ = '+';
        if (rec.flags & 1u) { memcpy(o, bytes + fastq_name2_off(rec), fastq_name2_len(rec)); o += fastq_name2_len(rec); }
        *o++ = '\n';
        memcpy(o, bytes + rec.qual_off + a, kept); o += kept;
        *o++ = '\n';
    }
    return ATR_OK;
}

}  // extern "C"
