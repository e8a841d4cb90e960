// filter_core.hpp -- bit-parallel (Myers/Hyyro) pre-pass of the batched Aligner.locate.
//
// The reference's DP (atropos/align/_align.pyx:378-459) costs ~7 VALU ops per cell in
// our packed-word form (locate_core.hpp).  Its COSTS, however, are the plain unit-cost
// edit-distance matrix with a free start in the read (START_WITHIN_SEQ2), and that matrix
// can be swept 64 rows at a time with Myers' bit-vector recurrence (~30 ops per column
// for a <= 64-base adapter).  What the bit-vectors cannot give is the reference's
// tie-broken (matches, origin) payload, so the pre-pass only decides, exactly:
//
//   * reads with no cell that could be accepted      -> result None, done;
//   * reads whose first zero-cost full-length hit is at column j (D[m][j] == 0, j >= m)
//     -> the reference stops there (:456-458) with (0, m, j-m, j, m, 0), done;
//   * everything else -> a column window [j_lo, j_hi] that provably contains the
//     traceback of every cell the reference could accept; the packed-word DP is then run
//     on that window only (locate_core.hpp, window mode), started from a fresh column:
//       - a cell (m, j) of cost c <= k starts in row 0 at a column >= j - m - k (at most k
//         of its steps are deletions), or in column 0 (START_WITHIN_SEQ1) only if j <= m+k;
//       - so with j_first = the first column holding any candidate (row m with D <= k, or
//         the last column if a last-column cell passes its threshold), starting the DP at
//         j_lo = max(0, j_first - m - k) from the column "row i reached from (0, j_lo) by i
//         insertions" reproduces every candidate cell bit for bit: along a candidate's
//         traceback the chosen predecessor is the same in both matrices, and every
//         non-chosen neighbour is >= its true value, so no comparison flips;
//       - after j_hi = the last candidate column nothing can be accepted any more.
//
// With an indel cost above 1 the unit-cost matrix is a lower bound of the reference's
// costs, so the same pre-pass is a (slightly looser) exact filter there too.
// Requires START_WITHIN_SEQ2 and STOP_WITHIN_SEQ2 (the 3'/5'/anywhere adapter types) and
// m <= 64; other aligners use the full sweep.
#ifndef ATR_FILTER_CORE_HPP
#define ATR_FILTER_CORE_HPP

#include "locate_core.hpp"

namespace atr {

constexpr int FILTER_MAX_M = 64;
constexpr int FILTER_BINS = 96;                    // window-start bins of 8 columns (n <= 736)

struct FilterParams {
    uint64_t peq[16];                               // peq[c] bit i: reference row i+1 matches query code c
};

// window word written per read by the pre-pass
//   [9:0] j_lo   [19:10] j_hi   [20] take last-column candidates   [31] valid (needs the DP)
ATR_DEV uint32_t window_word(int j_lo, int j_hi, bool scan) {
    return 0x80000000u | (uint32_t)j_lo | ((uint32_t)j_hi << 10) | (scan ? (1u << 20) : 0u);
}
ATR_DEV int window_lo(uint32_t w) { return (int)(w & 0x3FFu); }
ATR_DEV int window_hi(uint32_t w) { return (int)((w >> 10) & 0x3FFu); }
ATR_DEV bool window_scan(uint32_t w) { return ((w >> 20) & 1u) != 0; }
ATR_DEV bool window_valid(uint32_t w) { return (w >> 31) != 0; }

struct FilterState {
    uint64_t pv, mv;                                // vertical +1 / -1 deltas of the current column
    int score;                                      // D[m][j]
    int j_first, j_last;                            // first / last column with D[m][j] <= k (0: none)
    int j_exact;                                    // first column with D[m][j] == 0 and j >= m (0: none)
};

ATR_DEV void filter_init(FilterState &F, const Uniform &u) {
    // column 0: cost i per row (not START_WITHIN_SEQ1) or 0 everywhere (_align.pyx:333-352)
    F.pv = u.sr ? 0ull : ~0ull;
    F.mv = 0ull;
    F.score = u.sr ? 0 : u.m;
    F.j_first = F.j_last = F.j_exact = 0;
}

// One column of Myers' recurrence (Hyyro's formulation), row-0 delta 0 (free start in the read).
ATR_DEV void filter_step(FilterState &F, const Uniform &u, uint64_t eq, int j) {
    const uint64_t xv = eq | F.mv;
    const uint64_t xh = (((eq & F.pv) + F.pv) ^ F.pv) | eq;
    uint64_t ph = F.mv | ~(xh | F.pv);
    uint64_t mh = F.pv & xh;
    F.score += (int)((ph >> (u.m - 1)) & 1ull) - (int)((mh >> (u.m - 1)) & 1ull);
    ph <<= 1;
    mh <<= 1;
    F.pv = mh | ~(xv | ph);
    F.mv = ph & xv;
    if (F.score <= u.k) {
        if (F.j_first == 0) F.j_first = j;
        F.j_last = j;
        if (F.score == 0 && j >= u.m && F.j_exact == 0) F.j_exact = j;
    }
}

// Any last-column cell (row first_i..m, column n) that could pass the candidate test?
// The alignment length is at most the row, the threshold is monotone in the length.
ATR_DEV bool filter_last_column(const FilterState &F, const Uniform &u, const int16_t *thr) {
    int d = 0;                                      // D[0][n] = 0
    bool any = (u.er && 0 >= u.min_overlap && 0 <= (int)thr[0]);   // row 0 never qualifies (min_overlap >= 1)
    for (int i = 1; i <= u.m; ++i) {
        d += (int)((F.pv >> (i - 1)) & 1ull) - (int)((F.mv >> (i - 1)) & 1ull);
        if ((u.er || i == u.m) && i >= u.min_overlap && d <= (int)thr[i]) any = true;
    }
    return any;
}

// Decision for one read of length n after the sweep.  Returns the window word (0 when the
// read is resolved here, in which case rec[] holds its result record).
ATR_DEV uint32_t filter_decide(const FilterState &F, const Uniform &u, int n, const int16_t *thr, uint32_t rec[4]) {
    rec[0] = 0xFFFF0000u; rec[1] = 0; rec[2] = 0; rec[3] = 0;             // refstop = -1: None
    if (F.j_exact != 0 && u.m >= u.min_overlap) {
        // first perfect full-length occurrence: the reference breaks out here (:456-458)
        const int j = F.j_exact;
        rec[0] = (uint32_t)u.m << 16;                                   // refstart 0, refstop m
        rec[1] = (uint32_t)(j - u.m) | ((uint32_t)j << 16);             // querystart, querystop
        rec[2] = (uint32_t)u.m;                                         // matches m, errors 0
        return 0;
    }
    const bool lastcol = filter_last_column(F, u, thr);
    const bool rowm = F.j_first != 0 && u.m >= u.min_overlap;
    if (!lastcol && !rowm) return 0;
    const int first = rowm ? F.j_first : n;
    const int j_lo = atr_max(0, first - u.m - u.k);
    const int j_hi = lastcol ? n : F.j_last;
    return window_word(j_lo, j_hi, lastcol);
}

}  // namespace atr
#endif
