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
constexpr int FILTER_CERT_T = 3;                   // extension rows a perfect-overlap certificate covers (cost_l <= 3)
constexpr int FILTER_CERT_TW = 6;                  // ... of an adapter of more than 32 bases in the two-pass pre-pass (k <= 6)
constexpr int FILTER_BINS = 256;                   // 96 window-start bins of 8 columns (n <= 736) x 2 classes + 64 row-count bins

struct FilterParams {
    uint64_t peq[16];                               // peq[c]: rows matching query code c, top-aligned (see FilterState); pad bits 1
    int rows;                                       // rows the bit-vector sweeps: m, or 32 in NARROW mode (below)
    int and_mode;                                   // wildcard comparison (code & code) instead of equality
    uint32_t tail;                                  // NARROW mode: the codes of rows rows + 1 .. m, one nibble each
    int32_t thr_row[FILTER_MAX_M + 1];              // last-column test of row i: D[i][n] <= thr_row[i]; -1 for a row that is
                                                    // no candidate at all (below min_overlap, or not row m without
                                                    // STOP_WITHIN_SEQ1).  32-bit entries of a kernel argument: scalar loads.
    uint32_t cert[FILTER_CERT_T + 1];               // cert[t] bit i: a perfect overlap of i bases at the read end beats every
                                                    // longer last-column row up to i + t (filter_overlap_certificates); 0: ask the DP
    uint64_t cert_sub;                              // bit r - 1: the adapter with ONE substitution, at row r, is decided without the
                                                    // DP (filter_substitution_certificate); 0: never
    uint64_t rowsel[4];                             // rows whose code is the one-hot code of plane c, bit r - 1 (cert_sub != 0 only)
    uint32_t tailx[3];                              // extended NARROW mode (two-pass pre-pass, m = 41 .. 64): rows 41 .. 64, as `tail`
    uint64_t cert64[FILTER_CERT_TW + 1];            // `cert` for a two-word sweep (rows > 32): cert64[t] bit i, i < 64
};

// NARROW mode.  A 33 .. 40-base adapter needs two 32-bit words per bit-vector and ~28 VALU ops per
// column; its first 32 rows alone need one word and ~17.  An alignment of all m rows that ends in
// (m, j) with cost c <= k passes through row 32 in a cell (32, j') of cost <= c, j - T - k <= j' <= j
// (T = m - 32 tail rows, at most k of the remaining steps are deletions), so the columns where the
// 32-ROW matrix has D' <= k bracket every row-m candidate:
//   * window start: the alignment starts where its 32-row part starts, >= j'_first - 32 - k;
//   * window end: j <= j'_last + T + k;
//   * band: two cells of one traceback differ in diagonal by at most the indels between them, so
//     every traceback stays within k diagonals of its row-32 cell: diagonals
//     j'_first - 32 - k .. j'_last - 32 + k, the same span formula as with all rows;
//   * first perfect occurrence: D[m][j] == 0 implies D'[32][j - T] == 0, so if the first zero of
//     the 32-row matrix is followed by T matching tail bases that IS the first perfect occurrence;
//     if it is not, the read goes to the DP;
//   * last column: rows <= 32 are exact; a longer row i is kept when the lower bound
//     D'[32][n] - (i - 32) passes its threshold (a vertical delta is >= -1).
// All of it only widens what the exact DP kernels are asked to look at.  Not used with
// START_WITHIN_SEQ1 (an alignment may then start in column 0 below row 32).
constexpr int FILTER_NARROW_ROWS = 32, FILTER_NARROW_TAIL = 8;
inline bool filter_narrow_applies(int m, int flags) {          // host side
    return m > FILTER_NARROW_ROWS && m <= FILTER_NARROW_ROWS + FILTER_NARROW_TAIL && !(flags & ATR_START_WITHIN_SEQ1);
}
// EXTENDED NARROW mode (round 6, first cut; NOT used any more): every argument above holds for ANY number T of tail rows,
// so pass B of the two-pass pre-pass could keep its one-word bit-vector for adapters of up to 64 bases (the tail compare
// walks up to four dwords: filter_tail_cmp).  Measured on the README's 64-mer it lost: a fifth of 150-base reads end in
// 24 - 63 adapter bases, a last-column row above the 32 swept ones is only kept by the lower bound D'[32][n] - (i - 32),
// and every such read went to the column-window DP (1.6 of the call's 1.9 ms).  The two-pass pre-pass now sweeps ALL rows
// of a 41 .. 64-base adapter with two-word bit-vectors (piece_filter.hpp, sweep_decide<true>): 25 ops a column instead
// of 16, every last-column row exact, the partial adapters in the last-column band.
constexpr int FILTER_NARROW_TAIL_EXT = 32;
inline bool filter_narrow_ext_applies(int m, int flags) {
    return m > FILTER_NARROW_ROWS + FILTER_NARROW_TAIL && m <= FILTER_NARROW_ROWS + FILTER_NARROW_TAIL_EXT && !(flags & ATR_START_WITHIN_SEQ1);
}
// PERFECT OVERLAP + EXTENSION ROWS, decided without the DP (round 5).  The commonest unresolved read of a 3' adapter
// batch (1.2 M of C2's 2.26 M) ends with the adapter's first i bases verbatim: cell (i, n) costs 0 and holds i matches,
// and the rows i + 1 .. i + t below it are acceptable too (i insertions on: cost 1 .. t <= floor((i + t) e)), so the
// largest acceptable last-column row has errors and rounds 1 - 4 sent the read to the banded DP to learn that row i
// wins anyway.  It wins unless some acceptable row r in (i, i + t] holds MORE than i matches (_align.pyx:464-474: more
// matches, then fewer errors; a tie in matches loses to cost 0).  The matches of ANY path to (r, n) of cost <= kk =
// floor((i + t) e) are a common subsequence of the adapter's rows 1 .. r and the read's last r + kk bases (at most kk
// deletions): F . A[1 .. i], with F the t + kk bases before the overlap.  Whatever F holds,
//     LCS(A[1 .. i + t], F . A[1 .. i]) = max over a of LCS(A[1 .. a], F) + LCS(A[a + 1 .. i + t], A[1 .. i])
//                                      <= max over a of min(a, t + kk) + LCS(A[a + 1 .. i + t], A[1 .. i]),
// and the right side depends on the ADAPTER alone: if it is <= i for every a >= 1 (a = 0 gives exactly i: the overlap
// itself) no row can hold more than i matches and the record is (0, i, n - i, n, i, 0).  True for every i >= ~6 of an
// adapter without long self-similar stretches; a poly-A tail fails the test and keeps the DP.  Literal comparison only
// (with wildcards "matches A[j]" does not make the read's base equal to A[j]).  cert[t] bit i <=> the bound holds.
inline void filter_overlap_certificates(const uint8_t *codes, int m, int rows, const int32_t *thr_row, bool and_mode,
                                        uint32_t cert[FILTER_CERT_T + 1]) {
    for (int t = 0; t <= FILTER_CERT_T; ++t) cert[t] = 0u;
    if (and_mode || rows > 32 || rows < 2) return;
    const int R = m < rows ? m : rows;                             // rows whose last-column cost the sweep knows exactly
    // lcs[a][x][y] = LCS(A[a + 1 .. a + x], A[1 .. y]) for every a: one table per a serves every (i, t)
    static thread_local uint8_t tab[FILTER_MAX_M + 1][FILTER_MAX_M + 1];
    uint32_t bad[FILTER_CERT_T + 1];
    for (int t = 0; t <= FILTER_CERT_T; ++t) bad[t] = 0u;
    for (int a = 1; a <= R; ++a) {
        const int xs = R - a;
        for (int y = 0; y <= R; ++y) tab[0][y] = 0;
        for (int x = 1; x <= xs; ++x) {
            tab[x][0] = 0;
            for (int y = 1; y <= R; ++y) {
                const uint8_t up = tab[x - 1][y], left = tab[x][y - 1];
                uint8_t v = up > left ? up : left;
                if (codes[a + x - 1] == codes[y - 1] && (uint8_t)(tab[x - 1][y - 1] + 1) > v) v = (uint8_t)(tab[x - 1][y - 1] + 1);
                tab[x][y] = v;
            }
        }
        for (int t = 1; t <= FILTER_CERT_T; ++t)
            for (int i = 1; i + t <= R && i < 32; ++i) {
                const int top = i + t, kk = thr_row[top];
                if (kk < t || a > top) continue;                        // (row i + t is no candidate at cost t: never asked)
                const int free_bases = t + kk;
                const int bound = (a < free_bases ? a : free_bases) + (int)tab[top - a][i];
                if (bound > i) bad[t] |= 1u << i;
            }
    }
    for (int t = 1; t <= FILTER_CERT_T; ++t)
        for (int i = 1; i + t <= R && i < 32; ++i)
            if (thr_row[i] >= 0 && thr_row[i + t] >= t && !((bad[t] >> i) & 1u)) cert[t] |= 1u << i;
}

// The same bound for a sweep of more than 32 rows (the two-pass pre-pass on adapters of 41 .. 64 bases, round 6: a quarter
// of such an adapter's reads end in a perfect overlap whose extension rows qualify -- without the certificate every one
// of them went to the banded DP): t <= FILTER_CERT_TW, i < 64.
inline void filter_overlap_certificates64(const uint8_t *codes, int m, int rows, const int32_t *thr_row, bool and_mode,
                                          uint64_t cert[FILTER_CERT_TW + 1]) {
    for (int t = 0; t <= FILTER_CERT_TW; ++t) cert[t] = 0ull;
    if (and_mode || rows <= 32 || rows > FILTER_MAX_M) return;
    const int R = m < rows ? m : rows;
    static thread_local uint8_t tab[FILTER_MAX_M + 1][FILTER_MAX_M + 1];
    uint64_t bad[FILTER_CERT_TW + 1];
    for (int t = 0; t <= FILTER_CERT_TW; ++t) bad[t] = 0ull;
    for (int a = 1; a <= R; ++a) {
        const int xs = R - a;
        for (int y = 0; y <= R; ++y) tab[0][y] = 0;
        for (int x = 1; x <= xs; ++x) {
            tab[x][0] = 0;
            for (int y = 1; y <= R; ++y) {
                const uint8_t up = tab[x - 1][y], left = tab[x][y - 1];
                uint8_t v = up > left ? up : left;
                if (codes[a + x - 1] == codes[y - 1] && (uint8_t)(tab[x - 1][y - 1] + 1) > v) v = (uint8_t)(tab[x - 1][y - 1] + 1);
                tab[x][y] = v;
            }
        }
        for (int t = 1; t <= FILTER_CERT_TW; ++t)
            for (int i = 1; i + t <= R && i < 64; ++i) {
                const int top = i + t, kk = thr_row[top];
                if (kk < t || a > top) continue;
                const int free_bases = t + kk;
                const int bound = (a < free_bases ? a : free_bases) + (int)tab[top - a][i];
                if (bound > i) bad[t] |= 1ull << i;
            }
    }
    for (int t = 1; t <= FILTER_CERT_TW; ++t)
        for (int i = 1; i + t <= R && i < 64; ++i)
            if (thr_row[i] >= 0 && thr_row[i + t] >= t && !((bad[t] >> i) & 1ull)) cert[t] |= 1ull << i;
}

// ONE SUBSTITUTION, decided without the DP (round 5).  The commonest unresolved read after the overlap certificates
// holds the whole adapter with a single substituted base: on diagonal d the read disagrees with the adapter in exactly
// one row r, so cell W = (m, d + m) costs 1 (0 only with H = 0: the early exit) and -- the diagonal path being optimal
// at every prefix, the reference's tie order mismatch <= insertion <= deletion keeps to it (_align.pyx:405-419) -- holds
// m - 1 matches and origin d.  Every other candidate (m, j') (none in the last column: the caller checks) loses unless
//   (T1) it holds all m matches: the m rows in order on non-decreasing diagonal offsets o(i) in [-3k, 2k] off d, last minus
//        first <= k (its deletions), each row that lands inside the adapter's span matching the base W shows there --
//        A[i + o], or, at the substituted place, anything but A[r] -- and anything in the flanks (worst case); or
//   (T2) it comes EARLIER with m - 1 matches and cost 1: the adapter on another diagonal o in [-2k, -1] with at most one
//        disagreement, or one adapter row inserted (rows before it on offset o <= 0, rows after it on o - 1).
// Both are properties of the adapter and r alone: bit r - 1 is set when neither can happen.  For an adapter without long
// self-similar stretches that is every row but the first and last k (where the flank can finish the adapter).
inline uint64_t filter_substitution_certificate(const uint8_t *codes, int m, int k, bool and_mode) {
    if (and_mode || m < 8 || m > 64 || k < 1 || k > 4) return 0ull;
    for (int i = 0; i < m; ++i) if (codes[i] != 1 && codes[i] != 2 && codes[i] != 4 && codes[i] != 8) return 0ull;
    const int OLO = -3 * k, OHI = 2 * k, NO = OHI - OLO + 1;
    uint64_t cert = 0ull;
    for (int r = 1; r <= m; ++r) {
        // does row i (1-based) placed on offset o match what the read shows there?  (positions 1 .. m: the adapter's span)
        const auto ok = [&](int i, int o) {
            const int q = i + o;
            if (q < 1 || q > m) return true;                               // a flank base: whatever is needed
            return q == r ? codes[i - 1] != codes[r - 1] : codes[i - 1] == codes[q - 1];
        };
        bool threat = false;
        // (T1) reach[o][b]: rows 1 .. i placed, row i on offset OLO + o, b deletions used so far
        {
            bool cur[24][5], nxt[24][5];                                  // (5 k + 1 <= 21 offsets, k + 1 <= 5 budgets)
            for (int o = 0; o < NO; ++o) for (int b = 0; b <= k; ++b) cur[o][b] = (b == 0) && ok(1, OLO + o);
            for (int i = 2; i <= m; ++i) {
                for (int o = 0; o < NO; ++o) for (int b = 0; b <= k; ++b) nxt[o][b] = false;
                for (int o = 0; o < NO; ++o) for (int b = 0; b <= k; ++b) if (cur[o][b])
                    for (int step = 0; b + step <= k && o + step < NO; ++step) if (ok(i, OLO + o + step)) nxt[o + step][b + step] = true;
                for (int o = 0; o < NO; ++o) for (int b = 0; b <= k; ++b) cur[o][b] = nxt[o][b];
            }
            for (int o = 0; o < NO; ++o) for (int b = 0; b <= k; ++b) threat = threat || cur[o][b];
        }
        // (T2) another diagonal, earlier, with at most one disagreement
        for (int o = -2 * k; o <= -1 && !threat; ++o) {
            int bad = 0;
            for (int i = 1; i <= m; ++i) bad += ok(i, o) ? 0 : 1;
            threat = bad <= 1;
        }
        // (T2) one inserted row x: rows before it on offset o <= 0, rows after it on o - 1, every one of them matching
        for (int o = -2 * k; o <= 0 && !threat; ++o)
            for (int x = 1; x <= m && !threat; ++x) {
                bool all = true;
                for (int i = 1; i <= m && all; ++i) if (i != x) all = ok(i, i < x ? o : o - 1);
                threat = all;
            }
        if (!threat) cert |= 1ull << (r - 1);
    }
    return cert;
}

// peq64: the aligner's match masks (top-aligned in 64 bits when m > 32, in 32 bits otherwise)
// planes_path: the parameters of the two-pass pre-pass (extended NARROW mode for adapters of 41 .. 64 bases)
inline FilterParams filter_params(const uint64_t *peq64, const uint8_t *codes, int m, int flags, bool and_mode,
                                  const int16_t *thr, int min_overlap, bool planes_path = false) {
    FilterParams fp;
    fp.rows = m; fp.and_mode = and_mode ? 1 : 0; fp.tail = 0u;
    fp.tailx[0] = fp.tailx[1] = fp.tailx[2] = 0u;
    for (int i = 0; i <= FILTER_MAX_M; ++i)
        fp.thr_row[i] = (i >= 1 && i <= m && i >= min_overlap && ((flags & ATR_STOP_WITHIN_SEQ1) || i == m)) ? (int32_t)thr[i] : -1;
    for (int c = 0; c < 16; ++c) fp.peq[c] = peq64[c];
    (void)planes_path;     // (round 6, first cut: extended NARROW mode for m = 41 .. 64 -- replaced by the two-word sweep, see below)
    if (filter_narrow_applies(m, flags)) {
        fp.rows = FILTER_NARROW_ROWS;
        for (int c = 0; c < 16; ++c) fp.peq[c] = (peq64[c] >> (64 - m)) & 0xFFFFFFFFull;      // rows 1 .. 32 at bits 0 .. 31
        for (int t = 0; t < m - FILTER_NARROW_ROWS; ++t) {
            const uint32_t nib = (uint32_t)(codes[FILTER_NARROW_ROWS + t] & 15u) << (4 * (t & 7));
            if (t < 8) fp.tail |= nib; else fp.tailx[(t >> 3) - 1] |= nib;
        }
    }
    filter_overlap_certificates(codes, m, fp.rows, fp.thr_row, and_mode, fp.cert);
    filter_overlap_certificates64(codes, m, planes_path ? fp.rows : 0, fp.thr_row, and_mode, fp.cert64);
    fp.cert_sub = 0ull;
    fp.rowsel[0] = fp.rowsel[1] = fp.rowsel[2] = fp.rowsel[3] = 0ull;
    if (m <= FILTER_MAX_M && (flags & ATR_STOP_WITHIN_SEQ2) && !(flags & ATR_START_WITHIN_SEQ1) && m >= min_overlap && thr[m] >= 1) {
        const int k = (int)thr[m];                                         // floor(m e) = int(e m) for e >= 0
        fp.cert_sub = filter_substitution_certificate(codes, m, k, and_mode);
        if (fp.cert_sub)
            for (int i = 0; i < m; ++i) fp.rowsel[codes[i] == 1 ? 0 : codes[i] == 2 ? 1 : codes[i] == 4 ? 2 : 3] |= 1ull << i;
    }
    return fp;
}

// window word written per read by the pre-pass
//   [9:0] j_lo   [19:10] j_hi   [20] take last-column candidates   [27:21] highest row needed
//   [28] band: only row-m candidates, all of them inside BAND_W diagonals from j_lo on
//   [31] valid (needs the DP)
constexpr int BAND_W = 16;
ATR_DEV uint32_t window_word(int j_lo, int j_hi, bool scan, int rows, bool band) {
    return 0x80000000u | (uint32_t)j_lo | ((uint32_t)j_hi << 10) | (scan ? (1u << 20) : 0u) | ((uint32_t)rows << 21) |
           (band ? (1u << 28) : 0u);
}
ATR_DEV bool window_band(uint32_t w) { return ((w >> 28) & 1u) != 0; }
ATR_DEV int window_rows(uint32_t w) { return (int)((w >> 21) & 0x7Fu); }
ATR_DEV int window_lo(uint32_t w) { return (int)(w & 0x3FFu); }
ATR_DEV int window_hi(uint32_t w) { return (int)((w >> 10) & 0x3FFu); }
ATR_DEV bool window_scan(uint32_t w) { return ((w >> 20) & 1u) != 0; }
ATR_DEV bool window_valid(uint32_t w) { return (w >> 31) != 0; }
// Last-column band reads (band + scan bits; filter_decide): [9:0] first diagonal, rows field = the highest row
// looked at in the last column, the j_hi field holds (highest - lowest such row) in its low six bits and the
// number of diagonals - 1 above them, bit 29: row-m candidates inside the band as well.
ATR_DEV uint32_t last_band_word(int dlo, int top, int span, int width, bool rowm) {
    return window_word(dlo, span | (width << 6), true, top, true) | (rowm ? (1u << 29) : 0u);
}
ATR_DEV int last_band_span(uint32_t w) { return window_hi(w) & 63; }
ATR_DEV int last_band_width(uint32_t w) { return window_hi(w) >> 6; }
ATR_DEV bool last_band_rowm(uint32_t w) { return ((w >> 29) & 1u) != 0; }
// Scatter bin, chosen so that the 64 reads of a wave sweep nearly the same cells.
//   [0, 32)    band reads (band_locate below: row-m candidates on <= 16 diagonals), by band width, then window start;
//   [32, 96)   last-column band reads (band_locate_last), by row count, most rows first;
//   [96, 192)  reads that need every row of the column sweep, by window start / 8 -- and, in
//              ragged batches, the row-limited ones too;
//   [192, 256) equal-length batches (by_rows): the row-limited partial overlaps at the read end that the
//              last-column band does not take (indel cost > 1, START_WITHIN_SEQ1, a band of more than 16
//              diagonals; window = [n - rows - errors, n]) by row count, which makes a wave uniform in
//              rows AND window.
constexpr int BAND_BINS = 96, ROWS_BIN0 = 192;
// A wave made of [192, 256) reads only (equal read length n, last-column candidates only, at
// most `rows` rows) can skip the rows above rows - (n - j) + k in column j: a cell (i, j) on a
// path that ends in (i', n), i' <= rows, with b <= k deletions and a insertions has
// i = i' - (n - j) + b - a.  The skipped cells keep their initial value (cost i * indel, an upper
// bound of any cell of row i -- except in a real column 0 with START_WITHIN_SEQ1, where the
// window DP does not use the triangle), so they never win a comparison they would have lost.
ATR_DEV int triangle_rows(int rows, int n, int j, int k) { return rows - (n - j) + k; }
// Every wave of the window DP also skips, in column j, the rows above (j - jlo) + k (jlo = the wave's
// start column): a path that leaves row 0 at a column c0 >= jlo and reaches (i, j) with a insertions
// and b deletions has i = (j - c0) - b + a <= (j - jlo) + k.
// Ragged batches: the same row-count bins serve when the sweep runs in coordinates that count from the read
// END (window_kernel's tail mode: a lane's column j is its own column j - (max_len - n)), which needs an
// aligner without START_WITHIN_SEQ1 (a lane that joins the sweep later restarts from the ordinary initial
// column) -- see ragged_rows_bins().
ATR_DEV bool ragged_rows_bins(bool start_within_seq1) { return !start_within_seq1; }
constexpr int TAIL_COLUMNS = 64;                   // columns before the read end a tail-mode wave can sweep

// Band reads (band_kernel): [0, 32) by band width (a wave sweeps the diagonals of its widest read) and window start --
// a band read stages its own 16 diagonals, so the start only matters for locality -- and [32, 96) the LAST-COLUMN band reads (band_locate_last) by row count,
// which makes their waves uniform in the number of rows swept.
constexpr int LAST_BIN0 = 32;
ATR_DEV int window_bin(uint32_t w, int m, bool by_rows) {
    const int rows = window_rows(w), start = window_lo(w) >> 3;
    // (most rows first: the persistent grid of band_kernel then ends on its cheapest tasks)
    if (window_band(w)) {
        if (window_scan(w)) return LAST_BIN0 + 63 - atr_min(rows, 63);
        // row-m band reads: by the number of diagonals first -- band_locate sweeps as many as the WIDEST read of its wave
        // needs (8 .. 16 in steps of two; hi - lo - m is that number minus k + 1), the widest classes first -- then by
        // window start / 128 for locality
        const int wide = atr_min(7, atr_max(0, (window_hi(w) - window_lo(w) - m) >> 1));
        return (7 - wide) * 4 + atr_min(3, start >> 4);
    }
    if (rows >= m || !by_rows) return 96 + start;
    return ROWS_BIN0 + atr_min(rows, 63);
}

// The match masks of the eight bases of one dword w of a packed read, from the 16-entry table `peq` in LDS
// (8 bytes per query code).  Byte offset of base b: its nibble * 8 = (w >> (4 b - 3)) & 0x78 -- a right shift
// and an AND with a literal, both 2-cycle ops; only base 0 needs a left shift (one poisoning op in ~150).
#ifndef ATR_HOST_EMU
__device__ __forceinline__ void fetch_peq8(const uint2 *peq, uint32_t w, uint2 (&e)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        uint32_t off;
        if (b == 0) off = (w << 3) & 0x78u;
        else asm("v_lshrrev_b32 %0, %1, %2\n\tv_and_b32 %0, 0x78, %0" : "=v"(off) : "n"(4 * b - 3), "v"(w));
        e[b] = *(const uint2 *)((const char *)peq + off);
    }
}
#endif

struct FilterState {
    uint32_t pvl, pvh, mvl, mvh;                    // vertical +1 / -1 deltas of the current column (lo/hi words)
    int score;                                      // D[m][j] - (k + 1): negative <=> row m is a candidate in this column
    uint32_t hits;                                  // bit b: D[m][j - b] <= k, over the columns swept since the last fold
    uint32_t nz;                                    // bit b: 1 <= D[m][j - b] <= k (hits & ~nz = the zero-cost columns)
    int zfirst;                                     // first column j >= rows with D[m][j] == 0 (0: none), as of the last fold
    int j_first, j_last;                            // first / last column with D[m][j] <= k (0: none), as of the last fold
};

// Rows are TOP-ALIGNED in the bit-vector: row i (1-based) is bit off + i - 1 with
// off = W - m, W = 64 (adapter longer than 32 bases) or 32.  Row m is then the sign bit of
// the top word, so the bit that a left shift by one pushes out IS row m's horizontal delta:
// the shift is an add-with-carry-out and the score update an add/sub-with-carry-in.  The pad
// rows below row 1 match every base (peq pad bits = 1) and start with zero deltas; they stay
// at cost 0 in every column and play the part of the free row 0 (START_WITHIN_SEQ2).
// wide: the bit-vector has two words.  The single-aligner pipeline uses two words exactly when
// mf > 32; the linked-adapter pipeline (linked_core.hpp) sweeps adapters of different lengths
// side by side and gives all of them the same word count.
ATR_DEV int filter_row_offset(int m, bool wide) { return (wide ? 64 : 32) - m; }
ATR_DEV int filter_row_offset(int m) { return filter_row_offset(m, m > 32); }

// mf = FilterParams::rows in all of the following.
ATR_DEV void filter_init(FilterState &F, const Uniform &u, int mf, bool wide) {
    // column 0: cost i per row (not START_WITHIN_SEQ1) or 0 everywhere (_align.pyx:333-352)
    const int off = filter_row_offset(mf, wide);
    const uint32_t low = off >= 32 ? 0u : ~0u << (off & 31);           // rows living in the low word
    if (wide) { F.pvl = u.sr ? 0u : low; F.pvh = u.sr ? 0u : (off > 32 ? ~0u << (off & 31) : ~0u); }
    else { F.pvl = u.sr ? 0u : low; F.pvh = 0u; }
    F.mvl = F.mvh = 0u;
    F.score = (u.sr ? 0 : mf) - (u.k + 1);
    F.hits = F.nz = 0u;
    F.zfirst = 0;
    F.j_first = F.j_last = 0;
}
ATR_DEV void filter_init(FilterState &F, const Uniform &u, int mf) { filter_init(F, u, mf, mf > 32); }

// INSTRUCTION CLASSES (profiles/round4_valu_issue_sparse.txt).  On gfx950 the two-operand integer ops, right
// shifts, v_bitop3_b32 on three distinct registers and the carry ops issue in 2 cycles, and ONE op of the
// "poisoning" class (v_alignbit, v_min / v_max _u32, left shifts, v_lshl_add, v_bfe, ...) among eight puts the
// whole stream at 4 cycles per instruction.  The column update below therefore holds none: the shift of the
// horizontal deltas is an add-with-carry-out, the score update an add/sub-with-carry-in, and the two per-column
// flags -- "row m is a candidate" (D <= k) and "... with a cost of at least one" (1 <= D <= k) -- are pushed into
// their bit vectors as carries too (the sign of the biased score is the carry of score + score; score + k
// carries exactly for -k <= score <= -1).  The first zero-cost column (where the reference stops,
// _align.pyx:456-458) comes out of hits & ~nz when the vectors are folded, once per 32 columns; rounds 1-3
// kept a running v_min_u32 of (score << 10 | column) instead: three poisoning ops per column.
// A VALU write of a carry (VCC or an SGPR pair) needs two wait states before a VALU reads it on gfx950 (hipcc
// pads its own add/addc pairs with s_nop 1), so every pair of chains uses two carry registers and fills each
// other's wait states.
// ph <<= 1 and mh <<= 1 on one- or two-word vectors; the bit shifted out of the top of ph is added to score, the
// one out of mh subtracted.  The two flags are pushed ONE COLUMN LATE: the block first pushes the flags of the
// score it finds (the column before) -- hits = 2 hits + (D <= k), nz = 2 nz + (1 <= D <= k) -- and then updates the
// score, so that four carry registers are in flight and every carry is read three instructions after it was
// written: no s_nop (each cost 2.8 cycles of a 34-cycle column, profiles/round4_valu_issue_sparse.txt).
// filter_fold pushes the last column's flags before it reads the vectors.  kreg = k in a VGPR.
template <bool WIDE>
ATR_DEV void filter_shift_out2(uint32_t &phl, uint32_t &phh, uint32_t &mhl, uint32_t &mhh, int &score, uint32_t &hits,
                               uint32_t &nz, uint32_t kreg) {
#ifdef ATR_HOST_EMU
    hits = (hits << 1) | ((uint32_t)score >> 31);
    nz = (nz << 1) | (uint32_t)(((uint64_t)(uint32_t)score + (uint64_t)kreg) >> 32);
    const uint32_t pout = WIDE ? phh >> 31 : phl >> 31, mout = WIDE ? mhh >> 31 : mhl >> 31;
    if (WIDE) { phh = (phh << 1) | (phl >> 31); mhh = (mhh << 1) | (mhl >> 31); }
    phl <<= 1; mhl <<= 1;
    score += (int)pout - (int)mout;
#else
    uint64_t c0, c2, c3;                              // carry registers besides VCC (SGPR pairs)
    uint32_t t1, t2;                                  // sums nobody reads: only their carries count
    if (WIDE)
        asm("v_add_co_u32 %10, %8, %4, %4\n\t"
            "v_add_co_u32 %11, %9, %4, %12\n\t"
            "v_add_co_u32 %0, %7, %0, %0\n\t"
            "v_add_co_u32 %2, vcc, %2, %2\n\t"
            "v_addc_co_u32 %5, %8, %5, %5, %8\n\t"
            "v_addc_co_u32 %6, %9, %6, %6, %9\n\t"
            "v_addc_co_u32 %1, %7, %1, %1, %7\n\t"
            "v_addc_co_u32 %3, vcc, %3, %3, vcc\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %4, %7, 0, %4, %7\n\t"
            "v_subbrev_co_u32 %4, vcc, 0, %4, vcc"
            : "+v"(phl), "+v"(phh), "+v"(mhl), "+v"(mhh), "+v"(score), "+v"(hits), "+v"(nz), "=&s"(c0), "=&s"(c2), "=&s"(c3),
              "=&v"(t1), "=&v"(t2)
            : "v"(kreg) : "vcc");
    else
        asm("v_add_co_u32 %8, %6, %2, %2\n\t"
            "v_add_co_u32 %9, %7, %2, %10\n\t"
            "v_add_co_u32 %0, %5, %0, %0\n\t"
            "v_add_co_u32 %1, vcc, %1, %1\n\t"
            "v_addc_co_u32 %3, %6, %3, %3, %6\n\t"
            "v_addc_co_u32 %4, %7, %4, %4, %7\n\t"
            "v_addc_co_u32 %2, %5, 0, %2, %5\n\t"
            "v_subbrev_co_u32 %2, vcc, 0, %2, vcc"
            : "+v"(phl), "+v"(mhl), "+v"(score), "+v"(hits), "+v"(nz), "=&s"(c0), "=&s"(c2), "=&s"(c3), "=&v"(t1), "=&v"(t2)
            : "v"(kreg) : "vcc");
#endif
}

// One column of Myers' recurrence (Hyyro's formulation), row-0 delta 0 (free start in the
// read), written on explicit 32-bit halves: gfx950 has no full-rate 64-bit shift or add.
// WIDE = adapter longer than 32 bases (both words live); otherwise only the low word.
// kreg = k (wave-uniform in the single-aligner pipeline, the lane's own adapter's in the linked one).
// 16 VALU ops per column (25 WIDE), all of the 2-cycle classes.
template <bool WIDE>
ATR_DEV void filter_step(FilterState &F, uint32_t eql, uint32_t eqh, uint32_t kreg) {
    const uint32_t xvl = eql | F.mvl;
    const uint32_t tl = eql & F.pvl;
    const uint32_t sl = tl + F.pvl;
    const uint32_t xhl = ((sl ^ F.pvl) | eql);
    uint32_t phl = F.mvl | ~(xhl | F.pvl);
    uint32_t mhl = F.pvl & xhl;
    uint32_t xvh = 0, phh = 0, mhh = 0;
    if (WIDE) {
        xvh = eqh | F.mvh;
        const uint32_t th = eqh & F.pvh;
        const uint32_t sh = th + F.pvh + (sl < tl ? 1u : 0u);          // carry of the low-word add
        const uint32_t xhh = ((sh ^ F.pvh) | eqh);
        phh = F.mvh | ~(xhh | F.pvh);
        mhh = F.pvh & xhh;
    }
    filter_shift_out2<WIDE>(phl, phh, mhl, mhh, F.score, F.hits, F.nz, kreg);
    F.pvl = mhl | ~(xvl | phl);
    F.mvl = phl & xvl;
    if (WIDE) {
        F.pvh = mhh | ~(xvh | phh);
        F.mvh = phh & xvh;
    }
}

// Fold the flag bits of the (at most 32) columns swept since the last fold into j_first / j_last / zfirst;
// j = the last column this read has swept, mf = the rows swept.  A zero cost in a column j < mf (possible with
// START_WITHIN_SEQ1 only) is no full-length occurrence -- _align.pyx:456-458 needs the whole reference inside the
// read -- and is not recorded.
ATR_DEV void filter_fold(FilterState &F, int j, int mf, uint32_t kreg) {
    // the flags of column j itself (filter_shift_out2 pushes one column late); the bit a first step pushed for
    // column 0 -- no column of the read -- falls out here
    F.hits = (F.hits << 1) | ((uint32_t)F.score >> 31);
    F.nz = (F.nz << 1) | ((uint32_t)F.score + kreg < kreg ? 1u : 0u);
    if (j < 32) { const uint32_t keep = j <= 0 ? 0u : (1u << j) - 1u; F.hits &= keep; F.nz &= keep; }
    if (F.hits != 0u) {
        const int hi = 31 - atr_clz(F.hits), lo = atr_ctz(F.hits);
        if (F.j_first == 0) F.j_first = j - hi;
        F.j_last = j - lo;
        if (F.zfirst == 0) {
            const int span = j - mf;                                   // bit b is column j - b: b <= span
            const uint32_t ok = span >= 31 ? ~0u : span < 0 ? 0u : (2u << span) - 1u;
            const uint32_t z = F.hits & ~F.nz & ok;
            if (z != 0u) F.zfirst = j - (31 - atr_clz(z));
        }
    }
    F.hits = F.nz = 0u;
}

// q = 4 q + 2 (d1 < t1) + (d2 < t2), unsigned: the two tests as borrows of d - t (v_subrev_co_u32), taken over by two
// add-with-carry ops.  T: the thresholds are wave-uniform (an SGPR operand) or per lane.
template <class T>
ATR_DEV uint32_t filter_push_le2(uint32_t q, int d1, T t1, int d2, T t2) {
#ifdef ATR_HOST_EMU
    return (q << 2) | ((uint32_t)d1 < (uint32_t)t1 ? 2u : 0u) | ((uint32_t)d2 < (uint32_t)t2 ? 1u : 0u);
#else
    uint32_t x1, x2;
    uint64_t c;
    asm("v_subrev_co_u32 %1, vcc, %5, %4\n\t"
        "v_subrev_co_u32 %2, %3, %7, %6\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %0, %3, %0, %0, %3"
        : "+v"(q), "=&v"(x1), "=&v"(x2), "=&s"(c) : "v"(d1), "v"((uint32_t)t1), "v"(d2), "v"((uint32_t)t2) : "vcc");
    return q;
#endif
}

// Last column (column n): which rows could pass the candidate test?  The alignment length is
// at most the row and the threshold is monotone in the length, so D[i][n] <= thr[i] is a
// necessary condition.  Returns the largest such row (0: none) and its cost; exact = false
// when that row lies beyond the swept rows and was only kept by its lower bound (NARROW mode).
// P: FilterParams (wave-uniform kernel argument) or LaneFilterParams (linked_core.hpp: per lane,
// thresholds behind a pointer into LDS).
// NARROW mode, rows beyond the swept ones (i > mf): a candidate (i, n) of cost c <= thr_row[i] <= k leaves row mf
// for the last time in a cell (mf, j') of cost <= c -- a hit of the mf-row matrix -- with n - j' <= (i - mf) + k.
//   tail_rule 1: no hit in the columns n - T - k .. n  ->  no such row qualifies;
//   tail_rule 2: the only hit is column n itself  ->  the path ends with i - mf insertions below (mf, n), so
//                D'[mf][n] + (i - mf) <= thr_row[i] is necessary;
//   tail_rule 0: anything else -- the lower bound D'[mf][n] - (i - mf) (a vertical delta is >= -1).
// d_mf: D'[mf][n], for the caller's row-m test under rule 2.
template <bool WIDE, class P>
ATR_DEV int filter_last_column(const FilterState &F, const Uniform &u, const P &fp, int &cost_of_largest,
                               bool &exact, int tail_rule, int &d_mf, int &row_exact) {
    const int mf = fp.rows;
    int d = 0, largest = 0;                         // D[0][n] = 0; row 0 never qualifies (min_overlap >= 1)
    exact = true;
    const int off = filter_row_offset(mf, WIDE);
    const int low_rows = WIDE ? atr_max(0, atr_min(mf, 32 - off)) : mf;             // rows that live in the low word
    // Per row: d += the row's vertical delta, "d <= thr_row[i]" pushed into a bit vector -- bit 0 of the deltas walked down
    // by right shifts, the test as the BORROW of d - (thr + 1) taken over by an add-with-carry: eight ops of the 2-cycle
    // classes (round 3: two v_bfe, v_add3, v_cmp, v_cndmask -- twelve 4-cycle ops a row, a quarter of the decision).
    // q bit (rows - i) <=> row i qualifies; the largest qualifying row is the lowest set bit.
    const auto scan = [&](uint32_t pv, uint32_t mv, int first, int rows) -> uint32_t {
        uint32_t q = 0u;
        int i = 0;
        // (a real loop unless the row count is a constant -- the run-time compiled kernel: unrolled over a run-time count
        // the compiler keeps sixteen loop-invariant compare results in scalar registers and spills them)
#if !defined(ATR_SPEC) && !defined(ATR_HOST_EMU)
#pragma clang loop unroll(disable)
#endif
        for (; i + 2 <= rows; i += 2) {             // two rows per block: each carry is read three instructions after its write
            const int d1 = d + (int)(pv & 1u) - (int)(mv & 1u);
            const int d2 = d1 + (int)((pv >> 1) & 1u) - (int)((mv >> 1) & 1u);
            pv >>= 2; mv >>= 2;
            d = d2;
            q = filter_push_le2(q, d1, fp.thr_row[first + i] + 1, d2, fp.thr_row[first + i + 1] + 1);
        }
        if (i < rows) {
            d += (int)(pv & 1u) - (int)(mv & 1u);
            q = (q << 1) | ((uint32_t)d < (uint32_t)(fp.thr_row[first + i] + 1) ? 1u : 0u);
        }
        return q;
    };
    {
        const uint32_t q = scan(off >= 32 ? 0u : F.pvl >> (off & 31), off >= 32 ? 0u : F.mvl >> (off & 31), 1, low_rows);
        if (q != 0u) largest = low_rows - atr_ctz(q);
    }
    if (WIDE) {
        const int hoff = off > 32 ? off - 32 : 0, hrows = mf - low_rows;
        const uint32_t q = scan(F.pvh >> hoff, F.mvh >> hoff, low_rows + 1, hrows);
        if (q != 0u) largest = low_rows + hrows - atr_ctz(q);
    }
    if (WIDE) {   // its cost: the vertical deltas of rows 1 .. largest
        const uint64_t rows = (largest >= 64 ? ~0ull : ((1ull << largest) - 1ull)) << off;
        const uint64_t pv = ((uint64_t)F.pvh << 32) | F.pvl, mv = ((uint64_t)F.mvh << 32) | F.mvl;
        cost_of_largest = atr_popc64(pv & rows) - atr_popc64(mv & rows);
    } else {      // one word: no 64-bit view of (mvl, mvh) -- it made hipcc keep the pair together in the sweep (a v_mov_b64 per column)
        const uint32_t rows = (largest >= 32 ? ~0u : ((1u << largest) - 1u)) << off;
        cost_of_largest = atr_popc64((uint64_t)(F.pvl & rows)) - atr_popc64((uint64_t)(F.mvl & rows));
    }
    d_mf = d;
    row_exact = largest;                            // the largest qualifying row among the swept ones; cost_of_largest is its cost
    if (tail_rule != 1)
        for (int i = mf + 1; i <= u.m; ++i)
            if ((tail_rule == 2 ? d + (i - mf) : d - (i - mf)) <= fp.thr_row[i]) { largest = i; exact = false; }
    return largest;
}

// NARROW mode: do the T = m - mf bases after column jp equal the adapter's tail rows?
// q: this read's dwords (read_dword below).
ATR_DEV uint32_t read_dword(const uint32_t *q, int nchunks, int z8);
// dw(z): the dword of eight codes z (bases 8 z + 1 ..) of the caller's copy of the read; T <= 32 tail rows.
template <class P, class DW>
ATR_DEV bool filter_tail_cmp(const P &fp, int T, int jp, DW dw) {
    const uint32_t sh = 4u * (uint32_t)(jp & 7);
    bool ok = true;
    uint32_t lo = dw(jp >> 3);
#pragma unroll
    for (int q = 0; q < FILTER_NARROW_TAIL_EXT / 8; ++q) {
        if (8 * q >= T) break;                                                      // wave-uniform
        const uint32_t hi = dw((jp >> 3) + q + 1);
        const uint32_t w = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;             // bases jp + 8 q + 1 .. jp + 8 q + 8
        lo = hi;
        const uint32_t tw = q == 0 ? fp.tail : fp.tailx[q >= 1 ? q - 1 : 0];
        const uint32_t x = fp.and_mode ? (w & tw) : (w ^ tw);
        const int left = T - 8 * q;
        const uint32_t ones = left >= 8 ? 0x11111111u : (0x11111111u & ((1u << (4 * left)) - 1u));
        const uint32_t nz = (x | (x >> 1) | (x >> 2) | (x >> 3)) & ones;            // nibble != 0
        ok = ok && (fp.and_mode ? nz == ones : nz == 0u);
    }
    return ok;
}
template <class P>
ATR_DEV bool filter_tail_matches(const P &fp, int T, const uint32_t *q, int nchunks, int jp) {
    return filter_tail_cmp(fp, T, jp, [&](int z) { return read_dword(q, nchunks, z); });
}

// Decision for one read of length n after the sweep.  Returns the window word (0 when the
// read is resolved here, in which case rec[] holds its result record).
// s (linked adapters): the alignment is that of read[s:] -- the sweep saw the bases before s as
// "match nothing" columns, which leave the initial column of an aligner without
// START_WITHIN_SEQ1 unchanged (cost i in row i) -- so column s plays the part of column 0; all
// coordinates stay those of the whole read (the caller re-bases the record).
// tm(jp): do the T bases after column jp equal the adapter's tail rows (filter_tail_matches on the caller's copy of
// the read)?  have_last = false (two-pass pre-pass, piece_core.hpp): the sweep did not end in column n because no
// last-column cell can be acceptable -- F's vertical deltas are those of another column and are not looked at.
// cert64 of the parameter block (the linked pipeline's per-lane view has none)
ATR_DEV uint64_t filter_cert64(const FilterParams &fp, int t) { return fp.cert64[t]; }
template <class P> ATR_DEV uint64_t filter_cert64(const P &, int) { return 0ull; }

// (no diagonal view of the read: the substitution certificate is not asked for)
struct FilterNoDiag { ATR_DEV_MEMBER uint64_t operator()(int) const { return ~0ull; } };

// dg(d): the rows of the adapter that DISAGREE with the read on diagonal d (row r against column d + r), bit r - 1, all
// m rows; ~0 when the caller cannot tell.  Asked for at most once, and only when fp.cert_sub != 0.
// pa_dlo .. pa_dhi (two-pass pre-pass; pa_dhi < pa_dlo: none): diagonals that hold the whole traceback of every row-m
// candidate BY THE PIECES -- an alignment of all m rows with <= k errors keeps one of the k + 1 (or more) disjoint pieces
// intact, pass A has seen every exact occurrence of every piece (diagonals d_min .. d_max), and a path strays from the
// diagonal of its intact piece by at most its indels <= k: d_min - k .. d_max + k (the caller hands over max(0, d_min - k):
// a first diagonal of 0 may be a clamped one -- cells below the main diagonal exist -- and is not taken).  The hits of the swept rows spread
// over 2 (k - c) more columns than that (every column within k - c of a cost-c occurrence is a hit), which for k >= 5
// pushed every read with an error out of the 16-diagonal band into the column window (64-mers: 1.6 of 2.1 ms).
template <bool WIDE, class P, class TM, class DG = FilterNoDiag>
ATR_DEV uint32_t filter_decide_tm(const FilterState &F, const Uniform &u, const P &fp, TM tm, int n, uint32_t rec[4],
                                  int s = 0, bool have_last = true, DG dg = DG(), int pa_dlo = 0, int pa_dhi = -1) {
    const int mf = fp.rows, T = u.m - mf;                                // T > 0: NARROW mode
    rec[0] = 0xFFFF0000u; rec[1] = 0; rec[2] = 0; rec[3] = 0;             // refstop = -1: None
    if (F.zfirst != 0 && u.m >= u.min_overlap) {
        // first perfect full-length occurrence: the reference breaks out here (:456-458)
        const int j = F.zfirst + T;
        if (T == 0 || (j <= n && tm(j - T))) {
            rec[0] = (uint32_t)u.m << 16;                                   // refstart 0, refstop m
            rec[1] = (uint32_t)(j - u.m) | ((uint32_t)j << 16);             // querystart, querystop
            rec[2] = (uint32_t)u.m;                                         // matches m, errors 0
            return 0;
        }
    }
    int cost_l = 0, d_mf = 0, row_e = 0;
    bool exact_l = true;
    // NARROW mode: which hits of the mf-row matrix can a path to a longer row (or to row m) still use?
    const int tail_rule = T == 0 ? 0 : (F.j_first == 0 || F.j_last < n - T - u.k) ? 1 : F.j_first == n ? 2 : 0;
    const int row_l = have_last ? filter_last_column<WIDE>(F, u, fp, cost_l, exact_l, tail_rule, d_mf, row_e) : 0;
    const bool lastcol = row_l != 0;
    // rule 2: a row-m candidate would have to come down from (mf, n) by T insertions
    const bool rowm = F.j_first != 0 && u.m >= u.min_overlap && !(tail_rule == 2 && d_mf + T > u.k);
    if (!lastcol && !rowm) return 0;
    if (!rowm && !u.sr && exact_l && cost_l == 0) {
        // The only acceptable cells sit in the last column, and the longest of them is a
        // perfect overlap of row_l bases: a zero-cost cell is a pure diagonal, so its payload is
        // (matches row_l, origin n - row_l); every other acceptable cell lies in a smaller
        // row and therefore has fewer matches -> this cell wins (_align.pyx:464-474).
        rec[0] = (uint32_t)row_l << 16;
        rec[1] = (uint32_t)(n - row_l) | ((uint32_t)n << 16);
        rec[2] = (uint32_t)row_l;
        return 0;
    }
    if (!WIDE && !rowm && !u.sr && exact_l && u.indel == 1 && cost_l >= 1 && cost_l <= FILTER_CERT_T && row_l > cost_l) {
        // PERFECT OVERLAP + EXTENSION ROWS (filter_overlap_certificates): the rows row_l - cost_l + 1 .. row_l each add one
        // to the cost -- so (i, n), i = row_l - cost_l, costs 0: the read's last i bases are rows 1 .. i -- and the
        // adapter's own structure rules out a longer row with more than i matches.  (0, i, n - i, n, i, 0).
        const int i = row_l - cost_l, off = filter_row_offset(mf, false);
        const uint32_t seg = ((1u << cost_l) - 1u) << (off + i);             // rows i + 1 .. row_l (row r at bit off + r - 1)
        if ((F.pvl & seg) == seg && ((fp.cert[cost_l] >> i) & 1u) != 0u && n - i >= s) {
            rec[0] = (uint32_t)i << 16;
            rec[1] = (uint32_t)(n - i) | ((uint32_t)n << 16);
            rec[2] = (uint32_t)i;
            return 0;
        }
    }
    if (WIDE && !rowm && !u.sr && exact_l && u.indel == 1 && cost_l >= 1 && cost_l <= FILTER_CERT_TW && row_l > cost_l) {
        // ... the same with two words (filter_overlap_certificates64)
        const int i = row_l - cost_l, off = filter_row_offset(mf, true);
        const uint64_t seg = ((1ull << cost_l) - 1ull) << (off + i), pv = ((uint64_t)F.pvh << 32) | F.pvl;
        if ((pv & seg) == seg && ((filter_cert64(fp, cost_l) >> i) & 1ull) != 0ull && n - i >= s) {
            rec[0] = (uint32_t)i << 16;
            rec[1] = (uint32_t)(n - i) | ((uint32_t)n << 16);
            rec[2] = (uint32_t)i;
            return 0;
        }
    }
    if (!rowm && !u.sr && exact_l && u.indel == 1) {
        // LAST-COLUMN BAND.  Only last-column cells can be accepted, (row_l, n) is one of them (its cost is exact)
        // and holds at least row_l - cost_l matches, so no row below row_l - cost_l can win (a row holds at most
        // as many matches as it has bases, and the reference keeps the cell with MORE matches, :468).  The rows
        // row_l - cost_l .. row_l cost at most kk <= k each, so their tracebacks stay on the diagonals
        // n - row_l - kk .. n - row_l + cost_l + kk: a banded row-major DP over cost_l + 2 kk + 1 diagonals and
        // row_l rows (band_locate_last) instead of the column sweep over the window.  Word: j_lo = first
        // diagonal, j_hi field = cost_l, rows = row_l, band + scan bits.
        // (every row i <= row_l is accepted with cost <= thr_row[i] <= thr_row[row_l] only: kk, not k, bounds the indels)
        const int kk = atr_min(u.k, (int)fp.thr_row[row_l]), dlo = n - row_l - kk;
        if (dlo >= s && cost_l + 2 * kk <= BAND_W - 1) return last_band_word(dlo, row_l, atr_min(cost_l, row_l - 1), cost_l + 2 * kk, false);
    }
    if (rowm && lastcol && !u.sr && u.indel == 1 && row_e != 0) {
        // The same with row-m candidates besides (an adapter that ends at or near the read end).  (row_e, n), the
        // largest last-column candidate among the swept rows, is exact and holds >= row_e - cost matches, so the
        // last-column rows that can win are r0 = row_e - cost .. top (top = m when a longer row was only kept by a
        // bound); the row-m candidates keep to the diagonals j_first - mf - k .. j_last - mf + k as for a band read.
        const int top = exact_l ? row_l : u.m, r0 = atr_max(1, row_e - cost_l);
        const int dlo = atr_min(F.j_first - mf - u.k, n - top - u.k), dhi = atr_max(F.j_last - mf + u.k, n - r0 + u.k);
        if (dlo >= s && dhi - dlo <= BAND_W - 1) return last_band_word(dlo, top, top - r0, dhi - dlo, true);
    }
    // Window start: a cell (i, j) of cost c is reached from row 0 at a column >= j - i - (number
    // of deletions on its path), and that number is at most c (unit indel cost: c is the
    // exact D) or k.  Row-m candidates: the first one bounds them all (j - D[m][j] never
    // decreases with j); NARROW: see the note at FilterParams.  Last-column candidates: row + D
    // never decreases with the row, so the largest acceptable row bounds them all.
    int j_lo = 0x7fffffff;
    if (rowm) j_lo = F.j_first - mf - u.k;
    if (lastcol) j_lo = atr_min(j_lo, n - row_l - ((u.indel == 1 && exact_l) ? cost_l : u.k));
    j_lo = atr_max(s, j_lo);
    // Band: with row-m candidates only, a candidate ending in column j (j_first <= j <= j_last,
    // cost <= k) keeps to the diagonals j - m - k .. j - m + k (at most k of its steps are
    // indels), so every traceback lies on the diagonals j_lo .. j_last - m + k -- provided j_lo
    // was not clamped at 0 (no traceback then touches column 0).  The window end of a band
    // read is stored as (last diagonal) + m - k: j_last, or j'_last + T in NARROW mode.
    bool band = rowm && !lastcol && (F.j_first - mf - u.k >= s) &&
                (F.j_last - F.j_first + 2 * u.k <= BAND_W - 1);
    int band_hi = F.j_last + T;
    if (!band && rowm && !lastcol && !u.sr && pa_dhi >= pa_dlo && pa_dlo > s && pa_dhi - pa_dlo <= BAND_W - 1) {
        band = true;                                                     // the pieces' diagonals (see above)
        j_lo = pa_dlo;
        band_hi = pa_dhi + u.m - u.k;
    } else if (band && !u.sr && u.indel == 1 && fp.cert_sub != 0ull && ((F.j_first + F.j_last) & 1) == 0 &&
        F.j_last - F.j_first <= 2 * u.k) {
        // ONE SUBSTITUTION (filter_substitution_certificate).  The hits of the swept rows lie within k of their middle
        // column xc, so every row-m candidate ends within 2k of j = xc + T and starts within 3k before d = xc - mf:
        // what the certificate assumed.  Exactly one disagreeing row on diagonal d, and that row certified:
        // (0, m, d, d + m, m - 1, 1).
        const int xc = (F.j_first + F.j_last) >> 1, d = xc - mf, j = d + u.m;
        if (d >= s && j <= n) {
            const uint64_t mm = dg(d);
            if (mm != 0ull && (mm & (mm - 1ull)) == 0ull && (mm & fp.cert_sub) != 0ull) {
                rec[0] = (uint32_t)u.m << 16;
                rec[1] = (uint32_t)d | ((uint32_t)j << 16);
                rec[2] = (uint32_t)(u.m - 1) | (1u << 16);
                return 0;
            }
        }
    }
    const int j_hi = lastcol ? n : band ? band_hi : atr_min(n, F.j_last + (T ? T + u.k : 0));
    // rows: with a row-m candidate all m rows; otherwise nothing above the largest acceptable
    // last-column row can matter (a row only depends on the rows before it)
    return window_word(j_lo, j_hi, lastcol, rowm ? u.m : row_l, band);
}

// q: this read's dwords in the tile64 layout (read_dword).
template <bool WIDE, class P>
ATR_DEV uint32_t filter_decide(const FilterState &F, const Uniform &u, const P &fp, const uint32_t *q,
                               int nchunks, int n, uint32_t rec[4], int s = 0) {
    const int T = u.m - fp.rows;
    return filter_decide_tm<WIDE>(F, u, fp, [&](int jp) { return filter_tail_matches(fp, T, q, nchunks, jp); }, n, rec, s);
}

// ---- banded DP for the band reads --------------------------------------------------------
// Row-major sweep over the BAND_W diagonals d = j - i = j_lo + c, c = 0 .. BAND_W-1, of one read:
// the state is one register per diagonal, cell (i, c) is computed from (i-1, c) [diagonal
// step], (i-1, c+1) [insertion] and (i, c-1) [deletion], in place and in increasing c; cells
// outside the band count as unreachable, which is exact for the same reason the column
// window is (every traceback of an acceptable candidate lies inside; a neighbour that is too
// large never wins a comparison it would have lost).  Same packed cell word and v_min3
// tie-break as locate_core.hpp, payload = diagonal mismatches (XREP).  The reference row is
// wave-uniform, the read is a 16-base window sliding by one base per row.  About 9 VALU ops
// per cell, but m * (S + 1) cells instead of m * (m + 2k + S) in the column sweep.
struct BandParams {
    uint32_t rrep[FILTER_MAX_M];                     // reference code of row i, in all eight nibbles, at rrep[i - 1]
                                                     // (32-bit entries of a kernel argument: the row loop fetches its
                                                     // wave-uniform entry with a scalar load)
    int and_mode, noindel;
};

// The bases dlo + 1 .. dlo + 8 * BAND_STREAM of the read as BAND_STREAM dwords of eight
// bases (ns[k * nss] = bases dlo + 1 + 8k ..): the aligned dwords that hold them are gathered
// with independent loads and re-aligned to the band start with a per-lane funnel shift, so
// that the row loop never waits for memory.
constexpr int BAND_STREAM = (FILTER_MAX_M + BAND_W + 7) / 8 + 1;      // 11 dwords: rows 1 .. 64, 16 diagonals

ATR_DEV uint32_t read_dword(const uint32_t *q, int nchunks, int z8) {          // dword z8 (bases 8*z8 + 1 ..), 0 beyond the read
    if (z8 < 0 || z8 >= nchunks * 4) return 0u;
    return q[(size_t)(z8 >> 2) * 256 + (z8 & 3)];
}

// The row loop of an m-row adapter looks at the stream dwords 0 .. 2 + m / 8 only (band_stream_dwords(m)
// of them): the gathers beyond are skipped (wave-uniform), which for a 34-row adapter is five of thirteen
// loads and, most of the time, one 64-byte line of the read.
ATR_DEV int band_stream_dwords(int m) { return atr_min(BAND_STREAM, 3 + (m >> 3)); }
ATR_DEV void band_stage(const uint32_t *q, int nchunks, int dlo, uint32_t *ns, int nss, int ndw = BAND_STREAM) {
    const int z0 = dlo >> 3;                          // dlo + 1 is base (dlo & 7) of dword z0 (0-based base index dlo)
    const uint32_t sh = 4u * (uint32_t)(dlo & 7);
    uint32_t raw[BAND_STREAM + 1];
#pragma unroll
    for (int k = 0; k <= BAND_STREAM; ++k) raw[k] = k <= ndw ? read_dword(q, nchunks, z0 + k) : 0u;
#pragma unroll
    for (int k = 0; k < BAND_STREAM; ++k)
        if (k < ndw) ns[(size_t)k * nss] = sh ? ((raw[k] >> sh) | (raw[k + 1] << (32u - sh))) : raw[k];
}

// bit 3 of every nibble = "the nibble is not zero": v | v << 1, then | << 2 (bits that cross into
// the next nibble only land in its low three bits)
ATR_DEV uint32_t nibble_any(uint32_t v) {
    const uint32_t t = v | (v << 1);
    return t | (t << 2);
}

// The row loop over ND diagonals (compile-time: no per-cell test, the neighbours are registers).
// Diagonals smax + 1 .. ND - 1 are swept too; nothing on them is looked at afterwards.
// PREFIX (anchored 5' adapters, band_locate_prefix): diagonal c is j - i = c - k, and while i <= k the
// cells left of column 0 do not exist and column 0 holds the reference's initial value.
// rrep_of(i): the reference code of row i in all eight nibbles -- a wave-uniform table entry
// (BandParams::rrep) or, in the linked-adapter pipeline, an LDS entry of the lane's own adapter.
template <bool AND_MODE, int ND, bool PREFIX = false, class RR>
ATR_DEV void band_rows(const Uniform &u, RR rrep_of, const uint32_t *ns, int nss, uint32_t (&band)[BAND_W]) {
    const uint32_t inf = ((uint32_t)INIT_COST_CAP << CSH) | ORG_BIAS;
    // bases dlo + i + c of the read, c = 0 .. 15, for row i = 1: stream nibbles 0 .. 15
    uint32_t qw0 = ns[0], qw1 = ns[(size_t)nss];
    uint32_t feed = ns[(size_t)2 * nss];              // the dword the next bases come from
    for (int i = 1; i <= u.m; ++i) {
        const uint32_t rrep = rrep_of(i);
        uint32_t m0 = nibble_any(AND_MODE ? (qw0 & rrep) : (qw0 ^ rrep));          // nibble != 0, at bit 3
        uint32_t m1 = ND > 8 ? nibble_any(AND_MODE ? (qw1 & rrep) : (qw1 ^ rrep)) : 0u;
        if (AND_MODE) { m0 = ~m0; m1 = ~m1; }                                     // mismatch = no common bit
        uint32_t left = inf;
        if (PREFIX && i <= u.k) {                     // wave-uniform: the first k rows touch column 0
            // column 0 of this row sits at c = k - i: cost i insertions, origin 0 (:340-342)
            const uint32_t col0 = ORG_BIAS | ((uint32_t)atr_min(i * u.indel, INIT_COST_CAP) << CSH);
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                const int j = i + c - u.k;                                // wave-uniform
                const uint32_t bit = atr_bfe1(c < 8 ? m0 : m1, 4 * (c & 7) + 3);
                const uint32_t cd = atr_mad24(bit, COST1 + MATCH1, band[c]);
                const uint32_t up = c + 1 < ND ? band[c + 1 < ND ? c + 1 : 0] : inf;
                uint32_t nw = atr_minu(atr_minu(cd, left + u.delw), up + u.insw) & ~PRIO_MASK;
                if (j <= 0) nw = j == 0 ? col0 : inf;
                band[c] = nw;
                left = nw;
            }
        } else {
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                const uint32_t bit = atr_bfe1(c < 8 ? m0 : m1, 4 * (c & 7) + 3);
                const uint32_t cd = atr_mad24(bit, COST1 + MATCH1, band[c]);
                const uint32_t up = c + 1 < ND ? band[c + 1 < ND ? c + 1 : 0] : inf;
                const uint32_t nw = atr_minu(atr_minu(cd, left + u.delw), up + u.insw) & ~PRIO_MASK;
                band[c] = nw;
                left = nw;
            }
        }
        // slide the window by one base: stream nibble i + 15 comes in at the top
        qw0 = (qw0 >> 4) | (qw1 << 28);
        qw1 = (qw1 >> 4) | (feed << 28);
        feed >>= 4;
        if ((i & 7) == 0) {                           // wave-uniform: the next stream dword
            const int k = 2 + (i >> 3);
            feed = k < BAND_STREAM ? ns[(size_t)k * nss] : 0u;
        }
    }
}

// ns: the staged read (band_stage), stride nss.
template <bool AND_MODE>
ATR_DEV void band_locate(const Uniform &u, const uint32_t *rreps, bool noindel, const uint32_t *ns, int nss, int n,
                         uint32_t ww, int smax, const int16_t *thr, uint32_t rec[4]) {
    const int dlo = window_lo(ww);
    uint32_t band[BAND_W];
#pragma unroll
    for (int c = 0; c < BAND_W; ++c) band[c] = ORG_BIAS + (uint32_t)(dlo + c);     // row 0: cost 0, origin j (:385-388)
    const auto rr = [rreps](int i) { return rreps[i - 1]; };
    if (smax < 8) band_rows<AND_MODE, 8>(u, rr, ns, nss, band);                    // wave-uniform
    else if (smax < 10) band_rows<AND_MODE, 10>(u, rr, ns, nss, band);
    else if (smax < 12) band_rows<AND_MODE, 12>(u, rr, ns, nss, band);
    else if (smax < 14) band_rows<AND_MODE, 14>(u, rr, ns, nss, band);
    else band_rows<AND_MODE, 16>(u, rr, ns, nss, band);
    Best best;
    best.key = COST_FIELD_MAX - (u.m + n);
    best.word = (uint32_t)(u.m + n) << CSH;
    best.ref_stop = u.m; best.query_stop = n; best.matches = 0;
    const int cindel = noindel ? 0 : u.indel;
#pragma unroll
    for (int c = 0; c < BAND_W; ++c) {
        const int j = dlo + u.m + c;
        if (c <= smax && band[c] < u.klimit && j <= n) consider<true>(best, band[c], u.m, j, u.min_overlap, thr, cindel);
    }
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != u.m + n) {
        const int origin = (int)(best.word & ORG_MASK) - (int)ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}


// ---- last-column band reads (filter_decide, LAST-COLUMN BAND) --------------------------------------
// The same row-major sweep over ND diagonals from diagonal dlo = n - row_l - k on, rows 1 .. rows_max (the
// wave's largest row_l); cell (i, n) sits on band index row_l + k - i when row i is done, and the rows
// row_l - cost_l .. row_l go through the candidate test in the reference's order (ascending rows, :464-474).
// Columns beyond n hold garbage (bases read as code 0) that never feeds a column <= n.
// cap_lo: the wave's smallest row_l - cost_l (no lane looks at an earlier row).
template <bool AND_MODE, int ND>
ATR_DEV void band_rows_last(const Uniform &u, const uint32_t *rreps, const uint32_t *ns, int nss, uint32_t (&band)[BAND_W],
                            int rows_max, int cap_lo, bool active, int top, int r0, int at0, int n, const int16_t *thr,
                            int cindel, Best &best) {
    const uint32_t inf = ((uint32_t)INIT_COST_CAP << CSH) | ORG_BIAS;
    uint32_t qw0 = ns[0], qw1 = ns[(size_t)nss];
    uint32_t feed = ns[(size_t)2 * nss];
    for (int i = 1; i <= rows_max; ++i) {
        const uint32_t rrep = rreps[i - 1];
        uint32_t m0 = nibble_any(AND_MODE ? (qw0 & rrep) : (qw0 ^ rrep));
        uint32_t m1 = ND > 8 ? nibble_any(AND_MODE ? (qw1 & rrep) : (qw1 ^ rrep)) : 0u;
        if (AND_MODE) { m0 = ~m0; m1 = ~m1; }
        uint32_t left = inf;
#pragma unroll
        for (int c = 0; c < ND; ++c) {
            const uint32_t bit = atr_bfe1(c < 8 ? m0 : m1, 4 * (c & 7) + 3);
            const uint32_t cd = atr_mad24(bit, COST1 + MATCH1, band[c]);
            const uint32_t up = c + 1 < ND ? band[c + 1 < ND ? c + 1 : 0] : inf;
            const uint32_t nw = atr_minu(atr_minu(cd, left + u.delw), up + u.insw) & ~PRIO_MASK;
            band[c] = nw;
            left = nw;
        }
        if (i >= cap_lo) {                            // wave-uniform: some lane's rows of interest have begun
            const int at = at0 - i;                   // this lane's band index of cell (i, n): n - dlo - i
            if (active && i <= top && i >= r0 && (u.er || i == u.m)) {
                uint32_t cell = band[0];
#pragma unroll
                for (int c = 1; c < ND; ++c) cell = at == c ? band[c] : cell;
                if (cell < u.klimit) consider<true>(best, cell, i, n, u.min_overlap, thr, cindel);
            }
        }
        qw0 = (qw0 >> 4) | (qw1 << 28);
        qw1 = (qw1 >> 4) | (feed << 28);
        feed >>= 4;
        if ((i & 7) == 0) {
            const int k = 2 + (i >> 3);
            feed = k < BAND_STREAM ? ns[(size_t)k * nss] : 0u;
        }
    }
}

// active: this lane holds a last-column band read (other lanes of the wave run along and are ignored).
// rows_max: the wave's highest row to sweep (m as soon as a lane has row-m candidates too).
template <bool AND_MODE>
ATR_DEV void band_locate_last(const Uniform &u, const uint32_t *rreps, bool noindel, const uint32_t *ns, int nss, int n,
                              uint32_t ww, bool active, int smax, int rows_max, int cap_lo, const int16_t *thr,
                              uint32_t rec[4]) {
    const int dlo = window_lo(ww), top = window_rows(ww), r0 = top - last_band_span(ww), at0 = n - dlo;
    uint32_t band[BAND_W];
#pragma unroll
    for (int c = 0; c < BAND_W; ++c) band[c] = ORG_BIAS + (uint32_t)(dlo + c);     // row 0: cost 0, origin j (:385-388)
    Best best, bm;                                   // last-column candidates / row-m candidates
    best.key = COST_FIELD_MAX - (u.m + n);
    best.word = (uint32_t)(u.m + n) << CSH;
    best.ref_stop = u.m; best.query_stop = n; best.matches = 0;
    bm = best;
    const int cindel = noindel ? 0 : u.indel;
    // (the short overlaps -- floor(i e) = 1: cost 1, four diagonals -- are 40 % of C2's last-column band reads)
    if (smax < 4) band_rows_last<AND_MODE, 4>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else if (smax < 6) band_rows_last<AND_MODE, 6>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else if (smax < 8) band_rows_last<AND_MODE, 8>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else if (smax < 10) band_rows_last<AND_MODE, 10>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else if (smax < 12) band_rows_last<AND_MODE, 12>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else if (smax < 14) band_rows_last<AND_MODE, 14>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    else band_rows_last<AND_MODE, 16>(u, rreps, ns, nss, band, rows_max, cap_lo, active, top, r0, at0, n, thr, cindel, best);
    if (last_band_rowm(ww) && rows_max == u.m) {
        // the row-m cells of the band, in column order (:433-455) -- the reference sees them BEFORE the last
        // column, so on equal (matches, cost) they keep the place
        const int width = last_band_width(ww);
#pragma unroll
        for (int c = 0; c < BAND_W; ++c) {
            const int j = dlo + u.m + c;
            if (c <= width && band[c] < u.klimit && j <= n) consider<true>(bm, band[c], u.m, j, u.min_overlap, thr, cindel);
        }
        if (bm.key >= best.key) best = bm;
    }
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != u.m + n) {
        const int origin = (int)(best.word & ORG_MASK) - (int)ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}


// ---- anchored 5' adapters (PREFIX: only STOP_WITHIN_SEQ2) -------------------------------------
// The alignment must start at (0, 0), so with at most k indels every traceback stays on the
// diagonals j - i in [-k, k]: the same banded row-major DP, without any pre-pass, over the
// 2k + 1 diagonals around the main one (c = j - i + k).  Cells left of column 0 do not exist,
// column 0 and row 0 hold the reference's initial values (_align.pyx:333-352, :385-388).
// k < m: with k >= m (max_error_rate >= 1) row m of COLUMN 0 would pass the cost test, and the
// reference only looks at row m from column 1 on (_align.pyx:375, :433-455); such aligners take the
// full sweep.
inline bool prefix_band_applies(int flags, int m, int k) {          // host side
    return flags == ATR_STOP_WITHIN_SEQ2 && m <= FILTER_MAX_M && k >= 0 && k < m && 2 * k + 1 <= BAND_W;
}

// ns: the read staged from base 1 - k on (band_stage with dlo = -k), stride nss.
template <bool AND_MODE, class RR>
ATR_DEV void band_locate_prefix_rr(const Uniform &u, RR rr, bool noindel, const uint32_t *ns, int nss, int n,
                                   const int16_t *thr, uint32_t rec[4]) {
    const int k = u.k, smax = 2 * k;
    const uint32_t inf = ((uint32_t)INIT_COST_CAP << CSH) | ORG_BIAS;
    const int max_n = atr_min(n, u.m + k);                               // :317-319 (START_WITHIN_SEQ2 not set)
    uint32_t band[BAND_W];
#pragma unroll
    for (int c = 0; c < BAND_W; ++c) {                                   // row 0: cell (0, j = c - k): j deletions
        const int j = c - k;
        band[c] = j < 0 ? inf : (ORG_BIAS | ((uint32_t)atr_min(j * u.indel, INIT_COST_CAP) << CSH));
    }
    // the diagonals beyond 2k that a wider instantiation sweeps hold true DP cells; only c <= 2k is looked at
    if (smax < 4) band_rows<AND_MODE, 4, true>(u, rr, ns, nss, band);               // wave-uniform
    else if (smax < 6) band_rows<AND_MODE, 6, true>(u, rr, ns, nss, band);
    else if (smax < 8) band_rows<AND_MODE, 8, true>(u, rr, ns, nss, band);
    else if (smax < 12) band_rows<AND_MODE, 12, true>(u, rr, ns, nss, band);
    else band_rows<AND_MODE, 16, true>(u, rr, ns, nss, band);
    Best best;
    best.key = COST_FIELD_MAX - (u.m + n);
    best.word = (uint32_t)(u.m + n) << CSH;
    best.ref_stop = u.m; best.query_stop = n; best.matches = 0;
    const int cindel = noindel ? 0 : u.indel;
    uint32_t last = inf;
    bool have_last = false;
#pragma unroll
    for (int c = 0; c < BAND_W; ++c) {
        const int j = u.m + c - k;
        if (c <= smax && j >= 0 && j <= max_n) {
            if (band[c] < u.klimit) consider<true>(best, band[c], u.m, j, u.min_overlap, thr, cindel);   // :433-455
            if (j == n) { last = band[c]; have_last = true; }
        }
    }
    // last column (:461-474): only row m without STOP_WITHIN_SEQ1; (m, n) lies in the band unless
    // n < m - k, where its cost exceeds k anyway
    if (max_n == n && have_last) consider<true>(best, last, u.m, n, u.min_overlap, thr, cindel);
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != u.m + n) {
        const int origin = (int)(best.word & ORG_MASK) - (int)ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

template <bool AND_MODE>
ATR_DEV void band_locate_prefix(const Uniform &u, const uint32_t *rreps, bool noindel, const uint32_t *ns, int nss, int n,
                                const int16_t *thr, uint32_t rec[4]) {
    band_locate_prefix_rr<AND_MODE>(u, [rreps](int i) { return rreps[i - 1]; }, noindel, ns, nss, n, thr, rec);
}

}  // namespace atr
#endif
