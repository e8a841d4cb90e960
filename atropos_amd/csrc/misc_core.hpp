// misc_core.hpp -- per-thread arithmetic of the two small general-purpose kernels
// behind the rest of atropos.align's API surface:
//   * MultiAligner.locate for arbitrary flags and m != n  (_align.pyx:593-783)
//   * compare_prefixes / compare_suffixes                 (_align.pyx:501-544,
//                                                          align/__init__.py:28-44)
// They work on raw ASCII bytes (any alphabet), one pair per thread, and are not
// throughput paths: the insert aligner's own overlap search is insert_core.hpp, and
// anchored no-indel adapters are the only bulk caller of compare_prefixes.
// Compiled for gfx950 and, with -DATR_HOST_EMU, for the CPU test emulation.
#ifndef ATR_MISC_CORE_HPP
#define ATR_MISC_CORE_HPP

#include <stdint.h>
#include "atropos_hip.h"
#include "locate_core.hpp"      // atr_ctz
#include "insert_core.hpp"      // atr_bfrev

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

constexpr int MULTI_OVERHANG = 100000;              // _align.pyx:546

ATR_DEV int mc_min(int a, int b) { return a < b ? a : b; }
ATR_DEV int mc_max(int a, int b) { return a > b ? a : b; }

ATR_DEV void put_record(int16_t *rec, int origin, int ref_stop, int query_stop, int matches, int cost) {
    int s1 = 0, s2 = 0;
    if (origin >= 0) s2 = origin; else s1 = -origin;
    rec[0] = (int16_t)s1; rec[1] = (int16_t)ref_stop; rec[2] = (int16_t)s2; rec[3] = (int16_t)query_stop;
    rec[4] = (int16_t)matches; rec[5] = (int16_t)mc_min(cost, 32767); rec[6] = 0; rec[7] = 0;
}

// MultiAligner.locate for one (reference, query) pair.  `col` is this thread's scratch of
// 3*(m+1) ints (cost, matches, origin per row; strided by `cs`).  Writes up to `cap`
// records and returns the number of hits the reference would return (0 == None).
// Keeps the reference's evaluation order including Ukkonen's `last`, the early exit on a
// perfect match, the max_matches cut and the for...else last-column scan.
ATR_DEV int multi_locate_one(const uint8_t *ref, int m, const uint8_t *query, int n, double e, int flags,
                             int min_overlap, int max_matches, int *col, long long cs, int16_t *out, int cap) {
    const bool sr = (flags & ATR_START_WITHIN_SEQ1) != 0, sq = (flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (flags & ATR_STOP_WITHIN_SEQ2) != 0;
    const int max_cost = m + n;
    const int k = (int)(e * m);                                        // :634
    int max_n = n, min_n = 0;
    if (!sq) max_n = mc_min(n, m + k);
    if (!eq) min_n = mc_max(0, n - m - k);
#define COST(i) col[(long long)(3 * (i)) * cs]
#define MATS(i) col[(long long)(3 * (i) + 1) * cs]
#define ORIG(i) col[(long long)(3 * (i) + 2) * cs]
    for (int i = 0; i <= m; ++i) {                                     // :646-665
        MATS(i) = 0;
        if (!sr && !sq)      { COST(i) = mc_max(i, min_n) * MULTI_OVERHANG; ORIG(i) = 0; }
        else if (sr && !sq)  { COST(i) = min_n * MULTI_OVERHANG;          ORIG(i) = mc_min(0, min_n - i); }
        else if (!sr && sq)  { COST(i) = i * MULTI_OVERHANG;              ORIG(i) = mc_max(0, min_n - i); }
        else                 { COST(i) = mc_min(i, min_n) * MULTI_OVERHANG; ORIG(i) = min_n - i; }
    }
    int last = sr ? m : mc_min(m, k + 1);
    int nh = 0, exact = -1;
    bool broke = false;
    for (int j = min_n + 1; j <= max_n; ++j) {
        int dc = COST(0), dm = MATS(0), dorg = ORIG(0);
        if (sq) ORIG(0) = j; else COST(0) = j * MULTI_OVERHANG;
        const uint8_t qc = query[j - 1];
        for (int i = 1; i <= last; ++i) {
            int nc = dc, nm = dm;
            const int no = dorg;
            if (ref[i - 1] == qc) nm += 1; else nc += 1;               // diagonal only: :690-704
            dc = COST(i); dm = MATS(i); dorg = ORIG(i);
            COST(i) = nc; MATS(i) = nm; ORIG(i) = no;
        }
        while (last >= 0 && COST(last) > k) --last;
        if (last < m) { ++last; continue; }
        if (!eq) continue;
        const int cost = COST(m);
        if (cost > max_cost) continue;
        const int length = m + mc_min(ORIG(m), 0);
        if (length >= min_overlap && (double)cost <= length * e) {
            const int matches = MATS(m);
            if (cost == 0 && matches == m) {                           // :737-741
                exact = nh;
                if (nh < cap) put_record(out + 8 * nh, ORIG(m), m, j, matches, cost);
                else if (cap > 0) put_record(out + 8 * (cap - 1), ORIG(m), m, j, matches, cost);
                ++nh; broke = true;
                break;
            }
            if (nh < cap) put_record(out + 8 * nh, ORIG(m), m, j, matches, cost);
            if (++nh >= max_matches) { broke = true; break; }
        }
    }
    if (!broke && max_n == n) {                                        // for...else: :746-763
        for (int i = er ? 0 : m; i <= m; ++i) {
            const int cost = COST(i);
            if (cost > max_cost) continue;
            const int length = i + mc_min(ORIG(i), 0);
            if (length >= min_overlap && (double)cost <= length * e) {
                if (nh < cap) put_record(out + 8 * nh, ORIG(i), i, n, MATS(i), cost);
                ++nh;
            }
        }
    }
    if (exact >= 0) {                                                  // :767-768: only the exact hit
        if (exact != 0 && exact < cap) for (int t = 0; t < 8; ++t) out[t] = out[8 * exact + t];
        else if (exact != 0 && cap > 0) for (int t = 0; t < 8; ++t) out[t] = out[8 * (cap - 1) + t];
        return 1;
    }
    return nh;
#undef COST
#undef MATS
#undef ORIG
}

// ---- the same result without a DP: MultiAligner allows no indels, so a cell only depends on its diagonal ----------
// Every cell the reference can accept lies on a diagonal whose first cell -- in row 0 or in the initial column -- has
// cost 0 (the others start with an overhang cost of 100000 per position and never get below it), and its cost is the
// number of mismatches between that first cell and itself: one Hamming distance per candidate.  Ukkonen's `last`
// only ever hides cells of cost > k, which no test accepts (cost <= length * e implies cost <= k), and the stale
// values it leaves behind in the last column are init values or costs > k.  What remains of the DP is its ORDER:
// row-m cells by column (break at a perfect full-length hit, cut at max_matches), then -- unless the loop broke --
// the last column by row (the for ... else of :746-763), and "only the exact hit" at the end (:767-768).
// The kernel evaluates 64 candidates at a time, one per lane; this file holds what a lane does.
struct MultiSetup {
    bool sr, sq, er, eq;
    int m, n, k, min_n, max_n, min_overlap;
    double e;
};
ATR_DEV MultiSetup multi_setup(int m, int n, double e, int flags, int min_overlap) {
    MultiSetup S;
    S.sr = (flags & ATR_START_WITHIN_SEQ1) != 0; S.sq = (flags & ATR_START_WITHIN_SEQ2) != 0;
    S.er = (flags & ATR_STOP_WITHIN_SEQ1) != 0;  S.eq = (flags & ATR_STOP_WITHIN_SEQ2) != 0;
    S.m = m; S.n = n; S.e = e; S.min_overlap = min_overlap;
    S.k = (int)(e * m);                                                // :634
    S.max_n = S.sq ? n : mc_min(n, m + S.k);
    S.min_n = S.eq ? 0 : mc_max(0, n - m - S.k);
    return S;
}

// First cell (i0, j0) of the diagonal through (ie, je) and its origin; false when that cell carries an overhang cost.
ATR_DEV bool multi_diag_start(const MultiSetup &S, int ie, int je, int &i0, int &j0, int &origin) {
    const int d = je - ie;
    if (d > S.min_n) {                                                 // row 0, column d (:672-676)
        if (!S.sq) return false;                                       // cost d * overhang
        i0 = 0; j0 = d; origin = d;
        return true;
    }
    i0 = S.min_n - d; j0 = S.min_n;                                    // the initial column (:646-665)
    if (!S.sr && !S.sq) { origin = 0; return i0 == 0 && S.min_n == 0; }
    if (S.sr && !S.sq)  { origin = mc_min(0, S.min_n - i0); return S.min_n == 0; }
    if (!S.sr && S.sq)  { origin = mc_max(0, S.min_n - i0); return i0 == 0; }
    origin = S.min_n - i0;
    return i0 == 0 || S.min_n == 0;
}

// The candidate test of :717-745 / :750-763 for the cell (ie, je); rec: its record when accepted.
// perfect: a full-length hit without errors (the reference breaks out of its loop there).
ATR_DEV bool multi_candidate(const MultiSetup &S, const uint8_t *ref, const uint8_t *query, int ie, int je, int16_t rec[8],
                             bool &perfect) {
    perfect = false;
    int i0, j0, origin;
    if (!multi_diag_start(S, ie, je, i0, j0, origin) || i0 > ie) return false;
    const int len = ie - i0;
    int mism = 0;
    for (int t = 0; t < len; ++t) mism += (ref[i0 + t] != query[j0 + t]) ? 1 : 0;
    const int length = ie + mc_min(origin, 0);
    if (!(length >= S.min_overlap && (double)mism <= length * S.e)) return false;
    put_record(rec, origin, ie, je, len - mism, mism);
    perfect = mism == 0 && len == S.m;
    return true;
}

// The whole call, candidate by candidate in the reference's order (the CPU twin of multi_wave_kernel; the kernel does
// the same with 64 candidates per step and ballots).
ATR_DEV int multi_locate_diag(const uint8_t *ref, int m, const uint8_t *query, int n, double e, int flags, int min_overlap,
                              int max_matches, int16_t *out, int cap) {
    const MultiSetup S = multi_setup(m, n, e, flags, min_overlap);
    int nh = 0, exact = -1;
    bool broke = false;
    int16_t rec[8];
    bool perfect;
    if (S.eq) {
        for (int j = S.min_n + 1; j <= S.max_n && !broke; ++j) {
            if (!multi_candidate(S, ref, query, m, j, rec, perfect)) continue;
            const int slot = nh < cap ? nh : (perfect && cap > 0 ? cap - 1 : -1);
            if (slot >= 0) for (int t = 0; t < 8; ++t) out[8 * slot + t] = rec[t];
            if (perfect) { exact = nh; ++nh; broke = true; }
            else if (++nh >= max_matches) broke = true;
        }
    }
    if (!broke && S.max_n == n) {
        for (int i = S.er ? 0 : m; i <= m; ++i) {
            if (!multi_candidate(S, ref, query, i, n, rec, perfect)) continue;
            if (nh < cap) for (int t = 0; t < 8; ++t) out[8 * nh + t] = rec[t];
            ++nh;
        }
    }
    if (exact >= 0) {
        if (exact != 0 && exact < cap) for (int t = 0; t < 8; ++t) out[t] = out[8 * exact + t];
        else if (exact != 0 && cap > 0) for (int t = 0; t < 8; ++t) out[t] = out[8 * (cap - 1) + t];
        return 1;
    }
    return nh;
}

// compare_prefixes / compare_suffixes of one uniform reference against one query.
// tr / tq: translate tables or NULL for byte equality (both NULL together).
ATR_DEV void compare_one(const uint8_t *ref, int m, const uint8_t *query, int n, const uint8_t *tr,
                         const uint8_t *tq, bool suffix, int16_t *rec) {
    const int len = mc_min(m, n);
    const uint8_t *r = suffix ? ref + (m - len) : ref;
    const uint8_t *q = suffix ? query + (n - len) : query;
    int matches = 0;
    if (!tr) { for (int i = 0; i < len; ++i) matches += (r[i] == q[i]); }
    else { for (int i = 0; i < len; ++i) matches += ((tr[r[i]] & tq[q[i]]) != 0); }
    if (suffix) { rec[0] = (int16_t)(m - len); rec[1] = (int16_t)m; rec[2] = (int16_t)(n - len); rec[3] = (int16_t)n; }
    else { rec[0] = 0; rec[1] = (int16_t)len; rec[2] = 0; rec[3] = (int16_t)len; }
    rec[4] = (int16_t)matches; rec[5] = (int16_t)(len - matches); rec[6] = 0; rec[7] = 0;
}

// ---- Aligner.enable_debug(): the DP matrix as the reference would print it (_align.pyx:88-119, :354-357,
// :428-431).  A debugging aid for ONE (reference, query) pair, so this is the reference's own loop with plain
// integer cells and its Ukkonen cut-off `last` -- the matrix shows exactly the cells that loop computes (an entry
// it never wrote stays "not computed"), including the stale neighbours a re-entered row is computed from, and the
// TRUE indel cost (the throughput kernels cap it at k + 1, which changes no result but would change these
// numbers).  mismatch(i, code): does reference row i (1-based) differ from the query code?
constexpr int32_t DEBUG_NOT_COMPUTED = INT32_MIN;
struct DebugCell { int cost, matches, origin; };

template <class MIS, class CODE>
ATR_DEV void locate_debug_one(int m, int n, double e, int flags, int min_overlap, int indel, MIS mismatch, CODE code_at,
                              DebugCell *col /* m + 1 */, int32_t *matrix /* (m + 1) x (n + 1), row-major */, int16_t *rec) {
    const bool sr = (flags & ATR_START_WITHIN_SEQ1) != 0, sq = (flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (flags & ATR_STOP_WITHIN_SEQ2) != 0;
    const int k = (int)(e * m);
    const int max_n = sq ? n : mc_min(n, m + k), min_n = eq ? 0 : mc_max(0, n - m - k);
    for (long long t = 0; t < (long long)(m + 1) * (n + 1); ++t) matrix[t] = DEBUG_NOT_COMPUTED;
    for (int i = 0; i <= m; ++i) {                                       // :333-352
        DebugCell c;
        c.matches = 0;
        if (!sr && !sq) { c.cost = mc_max(i, min_n) * indel; c.origin = 0; }
        else if (sr && !sq) { c.cost = min_n * indel; c.origin = mc_min(0, min_n - i); }
        else if (!sr && sq) { c.cost = i * indel; c.origin = mc_max(0, min_n - i); }
        else { c.cost = mc_min(i, min_n) * indel; c.origin = min_n - i; }
        col[i] = c;
        matrix[(long long)i * (n + 1) + min_n] = c.cost;
    }
    int b_cost = m + n, b_matches = 0, b_origin = 0, b_ref_stop = m, b_query_stop = n;
    int last = sr ? m : mc_min(m, k + 1);
    for (int j = min_n + 1; j <= max_n; ++j) {
        DebugCell diag = col[0];
        if (sq) col[0].origin = j; else col[0].cost = j * indel;
        const uint32_t qc = code_at(j);
        for (int i = 1; i <= last; ++i) {
            DebugCell nw;
            if (!mismatch(i, qc)) {
                nw.cost = diag.cost; nw.origin = diag.origin; nw.matches = diag.matches + 1;
            } else {
                const int sub = diag.cost + 1, del = col[i].cost + indel, ins = col[i - 1].cost + indel;
                if (sub <= del && sub <= ins) { nw.cost = sub; nw.origin = diag.origin; nw.matches = diag.matches; }
                else if (ins <= del) { nw.cost = ins; nw.origin = col[i - 1].origin; nw.matches = col[i - 1].matches; }
                else { nw.cost = del; nw.origin = col[i].origin; nw.matches = col[i].matches; }
            }
            diag = col[i];
            col[i] = nw;
        }
        for (int i = 0; i <= last; ++i) matrix[(long long)i * (n + 1) + j] = col[i].cost;       // :428-431
        while (last >= 0 && col[last].cost > k) --last;
        if (last < m) {
            ++last;
        } else if (eq) {
            const int length = m + mc_min(col[m].origin, 0), cost = col[m].cost, matches = col[m].matches;
            if (length >= min_overlap && (double)cost <= (double)length * e &&
                (matches > b_matches || (matches == b_matches && cost < b_cost))) {
                b_matches = matches; b_cost = cost; b_origin = col[m].origin; b_ref_stop = m; b_query_stop = j;
                if (cost == 0 && matches == m) break;
            }
        }
    }
    if (max_n == n) {
        for (int i = er ? 0 : m; i <= m; ++i) {
            const int length = i + mc_min(col[i].origin, 0), cost = col[i].cost, matches = col[i].matches;
            if (length >= min_overlap && (double)cost <= (double)length * e &&
                (matches > b_matches || (matches == b_matches && cost < b_cost))) {
                b_matches = matches; b_cost = cost; b_origin = col[i].origin; b_ref_stop = i; b_query_stop = n;
            }
        }
    }
    rec[0] = 0; rec[1] = -1; rec[2] = rec[3] = rec[4] = rec[5] = rec[6] = rec[7] = 0;
    if (b_cost != m + n) {
        rec[0] = (int16_t)(b_origin >= 0 ? 0 : -b_origin); rec[1] = (int16_t)b_ref_stop;
        rec[2] = (int16_t)(b_origin >= 0 ? b_origin : 0); rec[3] = (int16_t)b_query_stop;
        rec[4] = (int16_t)b_matches; rec[5] = (int16_t)b_cost;
    }
}

// compare_prefixes / compare_suffixes of an aligner's reference against one tile64-packed read
// (the codes the read was packed with are the aligner's query table, so "does row i match this
// base" is the aligner's own nmask bit: byte equality or 4-bit AND as its wildcard flags say,
// _align.pyx:521-539).  words: the read's packed dwords, word w of the read at words[wstride(w)]
// with wstride(w) = (w >> 2) * 256 + (w & 3) for the lane's slot of a tile (chunk stride 64 uint4).
template <class WORD>
ATR_DEV void compare_packed_one(const uint32_t (*nmask)[4], int m, WORD word_at, int n, bool suffix, int16_t *rec) {
    const int len = mc_min(m, n);
    // nmask is indexed by register position: reference row i (0-based) owns bit round_up_rows_dev(m) - m + i
    const int r0 = round_up_rows_dev(m) - m + (suffix ? m - len : 0), q0 = suffix ? n - len : 0;
    int matches = 0;
    uint32_t cur = 0;
    for (int i = 0; i < len; ++i) {
        const int pos = q0 + i;
        if (i == 0 || (pos & 7) == 0) cur = word_at(pos >> 3);
        const uint32_t code = (cur >> (4 * (pos & 7))) & 15u;
        const int row = r0 + i;
        matches += 1 - (int)((nmask[code][row >> 5] >> (row & 31)) & 1u);
    }
    if (suffix) { rec[0] = (int16_t)(m - len); rec[1] = (int16_t)m; rec[2] = (int16_t)(n - len); rec[3] = (int16_t)n; }
    else { rec[0] = 0; rec[1] = (int16_t)len; rec[2] = 0; rec[3] = (int16_t)len; }
    rec[4] = (int16_t)matches; rec[5] = (int16_t)(len - matches); rec[6] = 0; rec[7] = 0;
}

// Post-filter of Adapter.match_to (atropos/adapters/__init__.py:386-398) on one result
// record, in place: keep the alignment iff size >= min_overlap and errors/size <=
// max_error_rate (a DIVISION in double, unlike the DP's product) and, when a table is given,
// rmp[size][matches] <= max_rmp.  accept_full: the exact-match shortcut (:351-367) -- a
// full-length zero-error occurrence of an adapter without wildcards is returned without
// passing the filters.  Rejected records get refstop = -1.
ATR_DEV void adapter_postfilter_one(int16_t *rec, int m, int min_overlap, double max_error_rate,
                                    const double *rmp, int rmp_ld, double max_rmp, bool accept_full) {
    if (rec[1] < 0) return;
    const int size = (int)rec[1] - (int)rec[0], matches = rec[4], errors = rec[5];
    bool ok = size >= min_overlap && (double)errors / (double)size <= max_error_rate;
    if (ok && rmp) ok = rmp[(size_t)mc_min(size, rmp_ld - 1) * rmp_ld + mc_min(mc_max(matches, 0), rmp_ld - 1)] <= max_rmp;
    if (accept_full && matches == m && errors == 0 && size == m) ok = true;
    if (!ok) { rec[0] = 0; rec[1] = -1; rec[2] = rec[3] = rec[4] = rec[5] = 0; }
}

// Python list indexing / slicing semantics (negative indices wrap once).
ATR_DEV int py_index(int idx, int n) {            // returns -1 when Python raises IndexError
    if (idx < 0) idx += n;
    return (idx < 0 || idx >= n) ? -1 : idx;
}
ATR_DEV void py_slice(int start, int stop, int n, int &a, int &b) {
    if (start < 0) start = mc_max(start + n, 0); else start = mc_min(start, n);
    if (stop < 0) stop = mc_max(stop + n, 0); else stop = mc_min(stop, n);
    a = start; b = mc_max(start, stop);
}

// ErrorCorrectorMixin.correct_errors for one pair (atropos/commands/trim/modifiers.py:
// 219-350), on raw ASCII sequences and qualities, in place.
//   im[4]   = insert_match[0..3] (ref start/stop in rc(read2), query start/stop in read1)
//   action  : 0 = 'N', 1 = 'conservative', 2 = 'liberal'
//   comp    : 256-entry complement table (BASE_COMPLEMENTS, 0 = no complement -> KeyError)
//   q1/q2   : may be NULL (no qualities; only action 'N' and the N-fill rules apply)
// Writes changed[0..1] = bases changed in read1 / read2 and newlen[0..1] = the sequence
// lengths after the call.  changed[0] < 0 reports the exception the reference raises for
// this pair: -1 KeyError (base without complement), -2 IndexError (overlap outside a
// read), -3 ValueError (mean of an empty quality slice).  Index arithmetic follows
// Python's list semantics (negative indices wrap), because the reference feeds
// synthesized overlaps of unequal-length reads through them.  Reference quirk kept: with
// truncate_seqs a corrected read1 that is longer than read2 comes back truncated to len2
// (update_read is handed the un-truncated len1, modifiers.py:343-345); read2 keeps its tail.
ATR_DEV void correct_errors_one(uint8_t *s1, uint8_t *q1, int len1, uint8_t *s2, uint8_t *q2, int len2,
                                const int16_t *im, int action, int min_qual_diff, bool truncate,
                                const uint8_t *comp, int32_t *changed, int32_t *newlen) {
    const bool has_quals = q1 != nullptr && q2 != nullptr;
    const int orig_len2 = len2;
    int n1 = len1, n2 = len2;                      // lengths of the working lists r1_seq / r2_seq
    if (truncate) {                                // :250-259
        if (len1 > len2) n1 = len2;
        else if (len2 > len1) { n2 = len1; len2 = len1; }
    }
    const int r1_start = im[2], r1_end = im[3];
    const int r2_start = len2 - im[1], r2_end = len2 - im[0];
    int c1 = 0, c2 = 0, err = 0, npend = 0;
    // zip(range(r1_start, r1_end), range(r2_end - 1, r2_start - 1, -1))
    const int steps = mc_max(0, mc_min(r1_end - r1_start, r2_end - r2_start));
    for (int t = 0; t < steps && !err; ++t) {
        const int i = py_index(r1_start + t, n1), j = py_index(r2_end - 1 - t, n2);
        if (i < 0 || j < 0) { err = -2; break; }
        const uint8_t base1 = s1[i];
        const uint8_t base2 = comp[s2[j]];
        if (base2 == 0) { err = -1; break; }
        if (base1 == base2) continue;
        if (action == 0) {                         // 'N'
            s1[i] = 'N'; s2[j] = 'N'; ++c1; ++c2;
        } else if (base1 == 'N') {
            s1[i] = base2;
            if (has_quals) q1[i] = q2[j];
            ++c1;
        } else if (base2 == 'N') {
            const uint8_t cb = comp[base1];
            if (cb == 0) { err = -1; break; }
            s2[j] = cb;
            if (has_quals) q2[j] = q1[i];
            ++c2;
        } else if (has_quals) {
            const int diff = (int)q1[i] - (int)q2[j];
            if (diff >= min_qual_diff) {
                const uint8_t cb = comp[base1];
                if (cb == 0) { err = -1; break; }
                s2[j] = cb; q2[j] = q1[i]; ++c2;
            } else if (diff <= -min_qual_diff) {
                s1[i] = base2; q1[i] = q2[j]; ++c1;
            } else if (action == 2) {
                ++npend;                           // quals_equal.append(...)
            }
        }
    }
    if (!err && npend > 0) {                       // :301-322
        int a1, b1, a2, b2;
        py_slice(r1_start, r1_end, n1, a1, b1);
        py_slice(r2_start, r2_end, n2, a2, b2);
        if (b1 <= a1 || b2 <= a2) {
            err = -3;                              // mean([]) raises ValueError
        } else {
            long long sum1 = 0, sum2 = 0;
            for (int i = a1; i < b1; ++i) sum1 += q1[i];
            for (int j = a2; j < b2; ++j) sum2 += q2[j];
            const double diff = (double)sum1 / (double)(b1 - a1) - (double)sum2 / (double)(b2 - a2);
            if (diff > 1.0 || diff < -1.0) {
                // Re-walk: after the first pass every handled position compares equal, so
                // the positions still unequal are exactly the recorded quals_equal entries
                // (neither base N, quality difference inside the dead band); their bases and
                // qualities are untouched since they were recorded.
                for (int t = 0; t < steps; ++t) {
                    const int i = py_index(r1_start + t, n1), j = py_index(r2_end - 1 - t, n2);
                    const uint8_t base1 = s1[i], base2 = comp[s2[j]];
                    if (base1 == base2 || base1 == 'N' || base2 == 'N') continue;
                    const int qd = (int)q1[i] - (int)q2[j];
                    if (qd >= min_qual_diff || qd <= -min_qual_diff) continue;
                    if (diff > 1.0) {
                        const uint8_t cb = comp[base1];
                        if (cb == 0) { err = -1; break; }
                        s2[j] = cb; q2[j] = q1[i]; ++c2;
                    } else {
                        s1[i] = base2; q1[i] = q2[j]; ++c1;
                    }
                }
            }
        }
    }
    changed[0] = err ? err : c1;
    changed[1] = err ? 0 : c2;
    newlen[0] = (c1 > 0 && !err) ? n1 : len1;      // the truncation quirk (see above)
    newlen[1] = orig_len2;                         // read2 keeps its tail (partial update, :336-339)
}

// ---- the same correction, guided by the bit planes of the two reads -------------------------------
// For an insert match of the insert aligner (flags START_WITHIN_SEQ1 | STOP_WITHIN_SEQ2, no indels) the
// tuple is (L - j, L, 0, j, ...): read1[t] faces the complement of read2[j - 1 - t], t = 0 .. j - 1
// (correct_errors_one walks exactly these pairs: r1 = [0, j), r2 = [0, j) backwards).  With both reads
// packed as DNA15 bit planes (plane64, atr_pack_planes) the positions where they DISAGREE come out 32
// at a time -- complement = plane p <-> 3 - p, and the 32 bases of read 2 that face word w of read 1
// are the bit-reversed window of read 2 at bit j - 32 (w + 1) -- and only those few positions are
// visited in the ASCII matrices; everywhere else the reference's loop does nothing.  Equal codes are
// equal characters here: read 2 holds upper-case IUPAC letters only (atr_insert_match_batch's packing
// check), and a character of read 1 without a code differs from all of them.

// b2 word source: plane p, word idx of read 2 (0 outside the read)
template <class W2>
ATR_DEV uint32_t facing_mismatches(const uint32_t a[4], W2 b2word, int nwords, int j, int w) {
    // bits of read 2 at [u, u + 32), u = j - 32 (w + 1), reversed: bit b faces read1[32 w + b]
    const int u = j - 32 * (w + 1);
    const int q = u >> 5;                                            // floor (arithmetic shift), u may be negative
    const uint32_t sh = (uint32_t)(u & 31);
    uint32_t diff = 0u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t lo = (q >= 0 && q < nwords) ? b2word(3 - p, q) : 0u;
        const uint32_t hi = (q + 1 >= 0 && q + 1 < nwords) ? b2word(3 - p, q + 1) : 0u;
        const uint32_t window = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        diff |= a[p] ^ atr_bfrev(window);
    }
    const int live = j - 32 * w;                                     // positions of this word inside the overlap
    return live >= 32 ? diff : live <= 0 ? 0u : diff & ((1u << live) - 1u);
}

// One disagreeing position (i in read 1, jx in read 2): the body of the reference's loop
// (modifiers.py:272-300).  Returns 0, or the error code (-1: base without complement).
// known1 / known2: the characters of read1[i] / read2[jx] when the caller already has them (decoded from
// the bit planes), else 0: read from the matrices.
// The decision once the position's bytes are on hand: base1 = read1[i], raw2 = read2[jx] (not yet
// complemented), qa / qb their qualities (ignored without quality rows).
// (the three counters as ONE packed word: with three `int &` the compiler indexed them in scratch memory)
constexpr uint32_t CORRECT_C1 = 1u, CORRECT_C2 = 1u << 10, CORRECT_NP = 1u << 20;
// (i / jx unsigned: the wave-level caller passes a tile's base pointers and 32-bit offsets, which the device addresses as
// scalar base + vector offset)
ATR_DEV int correct_apply_delta(uint8_t *s1, uint8_t *q1, uint8_t *s2, uint8_t *q2, uint32_t i, uint32_t jx, uint8_t base1, uint8_t raw2,
                          int qa, int qb, int action, int min_qual_diff, const uint8_t *comp, uint32_t &delta) {
    const bool has_quals = q1 != nullptr && q2 != nullptr;
    const uint8_t base2 = comp[raw2];
    if (base2 == 0) return -1;
    if (base1 == base2) return 0;
    if (action == 0) {                             // 'N'
        s1[i] = 'N'; s2[jx] = 'N'; delta += CORRECT_C1 + CORRECT_C2;
    } else if (base1 == 'N') {
        s1[i] = base2;
        if (has_quals) q1[i] = (uint8_t)qb;
        delta += CORRECT_C1;
    } else if (base2 == 'N') {
        const uint8_t cb = comp[base1];
        if (cb == 0) return -1;
        s2[jx] = cb;
        if (has_quals) q2[jx] = (uint8_t)qa;
        delta += CORRECT_C2;
    } else if (has_quals) {
        const int diff = qa - qb;
        if (diff >= min_qual_diff) {
            const uint8_t cb = comp[base1];
            if (cb == 0) return -1;
            s2[jx] = cb; q2[jx] = (uint8_t)qa; delta += CORRECT_C2;
        } else if (diff <= -min_qual_diff) {
            s1[i] = base2; q1[i] = (uint8_t)qb; delta += CORRECT_C1;
        } else if (action == 2) {
            delta += CORRECT_NP;                   // quals_equal.append(...)
        }
    }
    return 0;
}

ATR_DEV int correct_apply(uint8_t *s1, uint8_t *q1, uint8_t *s2, uint8_t *q2, int i, int jx, uint8_t base1, uint8_t raw2,
                          int qa, int qb, int action, int min_qual_diff, const uint8_t *comp, int &c1, int &c2, int &npend) {
    uint32_t delta = 0u;
    const int e = correct_apply_delta(s1, q1, s2, q2, (uint32_t)i, (uint32_t)jx, base1, raw2, qa, qb, action, min_qual_diff, comp, delta);
    c1 += (int)(delta & 1023u); c2 += (int)((delta >> 10) & 1023u); npend += (int)(delta >> 20);
    return e;
}

ATR_DEV int correct_position(uint8_t *s1, uint8_t *q1, uint8_t *s2, uint8_t *q2, int i, int jx, int action,
                             int min_qual_diff, const uint8_t *comp, int &c1, int &c2, int &npend,
                             uint8_t known1 = 0, uint8_t known2 = 0) {
    const bool has_quals = q1 != nullptr && q2 != nullptr;
    return correct_apply(s1, q1, s2, q2, i, jx, known1 ? known1 : s1[i], known2 ? known2 : s2[jx], has_quals ? (int)q1[i] : 0,
                         has_quals ? (int)q2[jx] : 0, action, min_qual_diff, comp, c1, c2, npend);
}

// sum of the first n bytes of a row (rows of the ASCII matrices start at any byte address: four bytes per
// load through an alignment-1 type, summed with one v_sad_u8 on the device)
typedef uint32_t __attribute__((aligned(1))) atr_u32_unaligned;
struct __attribute__((packed, aligned(1))) atr_u128_unaligned { uint32_t x, y, z, w; };
ATR_DEV long long byte_sum(const uint8_t *row, int n) {
    // sixteen bytes per load, four loads in flight per round (a lane walks its own row: every round trip counts)
    uint32_t acc0 = 0, acc1 = 0;
    int i = 0;
#ifndef ATR_HOST_EMU
    for (; i + 64 <= n; i += 64) {
        const atr_u128_unaligned v0 = *(const atr_u128_unaligned *)(row + i), v1 = *(const atr_u128_unaligned *)(row + i + 16);
        const atr_u128_unaligned v2 = *(const atr_u128_unaligned *)(row + i + 32), v3 = *(const atr_u128_unaligned *)(row + i + 48);
        acc0 = __builtin_amdgcn_sad_u8(v0.x, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v0.y, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v0.z, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v0.w, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v1.x, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v1.y, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v1.z, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v1.w, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v2.x, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v2.y, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v2.z, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v2.w, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v3.x, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v3.y, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v3.z, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v3.w, 0u, acc1);
    }
    for (; i + 16 <= n; i += 16) {
        const atr_u128_unaligned v = *(const atr_u128_unaligned *)(row + i);
        acc0 = __builtin_amdgcn_sad_u8(v.x, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v.y, 0u, acc1);
        acc0 = __builtin_amdgcn_sad_u8(v.z, 0u, acc0); acc1 = __builtin_amdgcn_sad_u8(v.w, 0u, acc1);
    }
#endif
    for (; i + 4 <= n; i += 4) {
        const uint32_t v = *(const atr_u32_unaligned *)(row + i);
#ifdef ATR_HOST_EMU
        acc0 += (v & 0xFFu) + ((v >> 8) & 0xFFu) + ((v >> 16) & 0xFFu) + (v >> 24);
#else
        acc0 = __builtin_amdgcn_sad_u8(v, 0u, acc0);
#endif
    }
    for (; i < n; ++i) acc0 += row[i];
    return (long long)acc0 + (long long)acc1;
}

// 'liberal' with positions of equal quality left over (:301-322): the read with the better mean quality over the
// overlap wins all of them.  mism: the disagreeing positions as facing_mismatches left them (positions settled in
// the first pass have equal bases by now and are skipped).
ATR_DEV void correct_ties(uint8_t *s1, uint8_t *q1, uint8_t *s2, uint8_t *q2, int j, const uint32_t *mism, int nwords,
                          int min_qual_diff, const uint8_t *comp, int &c1, int &c2, int &err) {
    {
        if (j <= 0) {
            err = -3;
        } else {
            const long long sum1 = byte_sum(q1, j), sum2 = byte_sum(q2, j);
            const double diff = (double)sum1 / (double)j - (double)sum2 / (double)j;
            if (diff > 1.0 || diff < -1.0) {
                for (int w = 0; w < nwords && !err; ++w) {
                    uint32_t m = mism[w];
                    while (m && !err) {
                        const int b = atr_ctz(m);
                        m &= m - 1u;
                        const int i = 32 * w + b, jx = j - 1 - i;
                        const uint8_t base1 = s1[i], base2 = comp[s2[jx]];
                        if (base1 == base2 || base1 == 'N' || base2 == 'N') continue;
                        const int qd = (int)q1[i] - (int)q2[jx];
                        if (qd >= min_qual_diff || qd <= -min_qual_diff) continue;
                        if (diff > 1.0) {
                            const uint8_t cb = comp[base1];
                            if (cb == 0) { err = -1; break; }
                            s2[jx] = cb; q2[jx] = q1[i]; ++c2;
                        } else {
                            s1[i] = base2; q1[i] = q2[jx]; ++c1;
                        }
                    }
                }
            }
        }
    }
}

// correct_errors(read1, read2, insert_match, truncate_seqs=True) for a pair with an insert match of
// j bases; mism[w] = facing_mismatches(...) of word w.  Same outputs as correct_errors_one.
// code1(i) / code2(jx): the DNA15 codes of read1[i] / read2[jx] out of the planes (0: a character without
// a code -- it is then read from the matrix), which spares the two scattered byte loads per position.
// (codes 3, 5, 6, 10, 12 are M R S Y K in the DNA15 table and the lower-case a c n g t of the case-sensitive one:
// those are left to the matrix)
ATR_DEV uint8_t dna15_letter(uint32_t code) { return (uint8_t)"\0AC\0G\0\0VTW\0H\0DBN"[code & 15u]; }

// NW: compile-time bound of the word loops (fully unrolled, so that code1 may index registers by w)
template <int NW, class C1, class C2>
ATR_DEV void correct_errors_planes_one(uint8_t *s1, uint8_t *q1, int len1, uint8_t *s2, uint8_t *q2, int len2, int j,
                                       const uint32_t *mism, int nwords, int action, int min_qual_diff,
                                       const uint8_t *comp, int32_t *changed, int32_t *newlen, C1 code1, C2 code2) {
    const int n1 = mc_min(len1, len2);             // both reads are cut to the common length (:250-259)
    int c1 = 0, c2 = 0, err = 0, npend = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t m = (w < nwords && !err) ? mism[w] : 0u;
        while (m && !err) {
            const int b = atr_ctz(m);
            m &= m - 1u;
            const int i = 32 * w + b;
            err = correct_position(s1, q1, s2, q2, i, j - 1 - i, action, min_qual_diff, comp, c1, c2, npend,
                                   dna15_letter(code1(w, b)), dna15_letter(code2(j - 1 - i)));
        }
    }
    if (!err && npend > 0) correct_ties(s1, q1, s2, q2, j, mism, nwords, min_qual_diff, comp, c1, c2, err);
    changed[0] = err ? err : c1;
    changed[1] = err ? 0 : c2;
    newlen[0] = (c1 > 0 && !err) ? n1 : len1;      // the truncation quirk of correct_errors_one
    newlen[1] = len2;
}

}  // namespace atr
#endif
