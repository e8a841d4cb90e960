// piece_core.hpp -- the two-pass pre-pass of the batched Aligner.locate on BIT-PLANE reads (round 4).
//
// The one-pass pre-pass (filter_core.hpp) sweeps every column of every read with Myers' recurrence: 150 columns
// x ~18 VALU ops per read, although half of the reads hold no trace of the adapter and the other half holds it
// in ~40 of their 150 columns.  This header splits the pre-pass in two, per read, in ONE kernel:
//
//   pass A, every read, 32 positions per boolean op.  The read is stored as four BIT PLANES of its 4-bit codes
//     (plane64, include/atropos_hip.h: word p of chunk c, bit b = bit p of the code of base 32 c + b), so "base
//     == A" for 32 positions is one or two ops.  An alignment of the adapter's first rows with at most k errors
//     (_align.pyx:380-426; the candidate rules :440-455, :464-474) leaves one of k + 1 disjoint PIECES of those
//     rows intact -- an error touches one piece -- and the piece then sits in the read at most k diagonals off
//     the alignment's own.  Pass A computes, for all positions at once, where each piece occurs exactly:
//         occ_p = AND_s (eq[code of the piece's row L-1-s] << s)            (bit e: the piece ends at position e)
//     with the four per-base masks shifted left one bit per step by add-with-carry chains (2-cycle ops; a funnel
//     shift is a 4-cycle op that slows its neighbours, profiles/round4_valu_issue_sparse.txt).  Partial adapters
//     at the read end (last-column candidates of rows below the pieces' rows) get their own necessary
//     conditions: an overlap that must be exact (floor(i e) == 0) IS tested exactly on its diagonal, the
//     classes with t >= 1 errors through t + 1 short pieces of their first rows on the diagonals they can reach.
//     No piece, no end condition -> the read's result is None, exactly (there is no cell the reference could
//     accept).  Otherwise the hits bound the columns in which any acceptable cell and its whole traceback lie.
//   pass B, flagged reads only, the columns of their window only: the flagged lanes of a wave's tiles queue up
//     in LDS ({read, 64-base window of the planes ending at the window's last column}); whenever 64 are there a
//     wave sweeps them -- one task per lane, Myers' recurrence of filter_core.hpp from a FRESH column ("row i
//     reached by i insertions") over at most 64 columns that END at the lane's own last column, so that lanes with
//     windows in different places run in lock step without any sorting -- and decides with filter_decide.
//     A fresh start at or before the first column a traceback can touch reproduces every cell of cost <= k bit
//     for bit (note at the top of filter_core.hpp); columns before the read's first base hold code 0, match no
//     row and leave the fresh column unchanged, so a window may begin before the read.
//     Reads whose hits need more than PIECE_NARROW columns (a second occurrence, a chance hit somewhere else:
//     3 % of C2's reads) go to a global list and take the full sweep in a small kernel of their own, 64 to a wave.
//
// Exactness of the windowed decision: the swept window is itself a genuine DP matrix (of the read's infix with a
// free start), every acceptable cell of the whole matrix has its traceback inside it and therefore the same
// value, every other cell is at least its true value, and filter_decide only ever uses (a) exact values of
// acceptable cells and (b) lower-bound tests that hold in any such matrix.
//
// Envelope (piece_applies): aligners of the filtered pipeline without START_WITHIN_SEQ1 (3' adapters, the BACK
// type), at most 32 swept rows (adapters of up to 40 bases: NARROW mode), k <= 3, one-hot codes (A C G T) in
// the piece rows, pieces of at least 5 bases, equal-length batches of up to 320 bases.  Everything else keeps
// the one-pass pre-pass.
#ifndef ATR_PIECE_CORE_HPP
#define ATR_PIECE_CORE_HPP

#include "filter_core.hpp"

namespace atr {

constexpr int PIECE_NB = 8;                 // body pieces at most: max(4, k + 1), k <= 6 (a piece too many only weakens the filter);
                                            // 41 .. 64-base adapters take more when their last piece would get too long
constexpr int PIECE_KMAX = 6;               // int(e m) at most (the 64-mer of the reference's README at e = 0.1)
constexpr int PIECE_NT = 5;                 // read-end pieces at most (error classes t = 1 .. 4; classes beyond are served by body pieces)
constexpr int PIECE_TBIT = 2 * PIECE_NB;    // scode: body piece p at bits 2 p, read-end piece u at bits PIECE_TBIT + 2 u
constexpr int PIECE_STEPS = 8;              // longest read-end piece / regular body piece
constexpr int PIECE_LAST_STEPS = 24;        // the last body piece takes the rows that are left: up to 24
constexpr int PIECE_TAIL_WORDS = 3;         // plane words the read-end pieces are evaluated on: the read's last three
constexpr int PIECE_HEAD_WORDS = 2;         // ... the read-start pieces (START_WITHIN_SEQ1): its first two
constexpr int PIECE_WINDOW = 64;            // columns a pass-B task carries (two words per plane) ...
constexpr int PIECE_WINDOW_MAX = 96;        // ... three for adapters whose rows + 2 k exceed that (PieceParams::window)
constexpr int PIECE_NARROW = 40;            // ... and the most it sweeps when k <= 3 (PieceParams::narrow): one hit diagonal needs
                                            // rows + T + 2 k columns, a read-end condition rows + T + k (94 % of C2's flagged
                                            // reads); longer windows take the full sweep
constexpr int PIECE_MAX_WORDS = 10;         // reads of up to 320 bases
constexpr uint32_t PIECE_NODENSE = 1u << 30;   // window word of an `order` entry: no 64-code record in tdata, gather the read

struct PieceParams {
    int blen;                               // body piece p = rows [p blen, (p + 1) blen), 0-based rows; the LAST one (nb - 1) runs to
    int llen;                               //   row m: llen = m - (nb - 1) blen rows (all nb on one diagonal = the adapter, verbatim)
    int m;                                  // adapter length
    int tlen;                               // read-end piece u = rows [u tlen, (u + 1) tlen); 0: none
    int steps;                              // max(llen, tlen)
    int xlo, xhi;                           // rows i in [xlo, xhi]: the overlap of i bases must be exact (xhi < xlo: none)
    int tail_cols;                          // columns before the read end a read with a read-end condition sweeps
    int tw0;                                // first plane word of the read-end pieces' masks: max(0, ceil(n / 32) - 3)
    int and_mode;                           // wildcard comparison (code & code): a base matches code c iff its plane c is set
    int nb;                                 // body pieces: max(4, k + 1)
    int narrow;                             // columns a pass-B task sweeps at most: 8 ceil((rows + T + 2 k) / 8)  (40 for k <= 3)
    int window;                             // columns a pass-B task carries: PIECE_WINDOW, or PIECE_WINDOW_MAX when narrow exceeds it
    int plen[PIECE_NB];                     // rows of body piece p (0: not there)
    int pshift[PIECE_NB];                   // its last row minus piece 0's last row: hits of piece p, moved down by this, sit in piece 0's place
    uint32_t scode[PIECE_LAST_STEPS];       // per shift step s, two bits per piece: the plane index (0 .. 3) of the row it
                                            // compares in that step (its last row minus s); body piece p at bits 2p,
                                            // read-end piece u at bits PIECE_TBIT + 2u (one scalar load per step, s_bitcmp1 per term)
    uint32_t tmask[PIECE_NT][PIECE_TAIL_WORDS];   // END positions a read-end piece may have (words tw0 ..), 0: piece unused
    uint32_t xmask[32][4];                  // [i][c]: the rows r < i that hold the code of plane c, at bit 32 - i + r (the place of
                                            // row r in the read's last 32 positions when the overlap has i bases)
    // START_WITHIN_SEQ1 (round 6; the 5' and "anywhere" adapter types, flags 11 / 15): an alignment may begin in column 0
    // at any row i0 -- the adapter's last L = m - i0 rows against the read's first bases, cost <= floor(L e)
    // (_align.pyx:333-352, :440-455).  Necessary conditions, mirrored from the read end: overlaps that must be exact are
    // tested on the read's first 32 positions; class t >= 1 through t + 1 read-START pieces cut from the adapter's END.
    int sr;                                 // the aligner has START_WITHIN_SEQ1
    int slen;                               // read-start piece u = rows [m - (u + 1) slen, m - u slen); 0: none
    int sxlo, sxhi;                         // suffix overlaps of L in [sxlo, sxhi] bases must be exact (sxhi < sxlo: none)
    int head_cols;                          // columns from the read start a read with a read-start condition sweeps: m + k
    uint32_t scode2[PIECE_LAST_STEPS];      // as scode, the read-start pieces: piece u at bits 2u
    uint32_t smask[PIECE_NT][PIECE_HEAD_WORDS];   // END positions a read-start piece may have (words 0, 1)
    uint32_t xsmask[32][4];                 // [L][c]: the rows m - L + r (r < L) that hold the code of plane c, at bit r
    int aonly;                              // pass A ONLY (START_WITHIN_SEQ1 adapters of 33 .. 64 bases: pass B's one-word sweep cannot
                                            // take them -- no NARROW mode with column-0 starts): what pass A leaves goes to the window DP
                                            // with the columns pass A bounds, instead of the whole read as the one-pass pre-pass has it
};

// one-hot code (1, 2, 4, 8) -> plane index, -1 otherwise
inline int piece_plane_of(int code) { return code == 1 ? 0 : code == 2 ? 1 : code == 4 ? 2 : code == 8 ? 3 : -1; }

#ifndef __HIPCC_RTC__
// Host: does the two-pass pre-pass take this aligner (m rows, codes[], k, flags, thr_row as in FilterParams,
// rows = FilterParams::rows) on equal-length reads of n bases?  Fills pp.
// thr[L] = floor(L e) (LocateParams::thr), min_overlap: for the read-start classes of a START_WITHIN_SEQ1 aligner.
inline bool piece_params(const uint8_t *codes, int m, int rows, int k, int flags, bool and_mode, bool custom_table,
                         const int32_t *thr_row, int n, PieceParams &pp, const int16_t *thr = nullptr, int min_overlap = 1) {
    memset(&pp, 0, sizeof(pp));
    const int need = ATR_START_WITHIN_SEQ2 | ATR_STOP_WITHIN_SEQ2;
    if ((flags & need) != need || custom_table) return false;
    const bool sr = (flags & ATR_START_WITHIN_SEQ1) != 0;
    // (START_WITHIN_SEQ1: every row swept -- NARROW mode does not apply, filter_core.hpp -- and reads longer than any
    //  alignment that touches both column 0 and the last column)
    if (sr && (!thr || rows != m || n <= m + k)) return false;
    // (pass-A-only mode: START_WITHIN_SEQ1 adapters of 33 .. 64 bases before pass B had its two-word sweep; ATR_PIECE_AONLY=1
    //  brings it back for comparison)
    static const bool aonly_env = [] { const char *x = getenv("ATR_PIECE_AONLY"); return x && x[0] == '1'; }();
    const bool aonly = sr && rows > 32 && aonly_env;
    if (rows < 1 || rows > FILTER_MAX_M || k < 0 || k > PIECE_KMAX || k >= m || m > FILTER_MAX_M) return false;
    if (n < 1 || n > 32 * PIECE_MAX_WORDS) return false;
    // The pieces cover ALL m rows (an alignment of the whole adapter with <= k errors leaves one of any k + 1 disjoint
    // pieces intact): nb - 1 regular ones of blen rows, the last one takes what they leave.  All pieces on one diagonal are
    // then the adapter verbatim -- the reference's early exit (_align.pyx:456-458) -- which pass A resolves itself.
    // More pieces than k + 1 when the last one would get too long, or (below) when an error class of the read-end
    // overlaps has no read-end pieces of its own and must find t + 1 regular pieces inside its shortest overlap.
    int nb = k + 1 > 4 ? k + 1 : 4, blen = 0, llen = 0;
    const auto cut = [&]() {
        blen = std::min(PIECE_STEPS, m / nb);
        llen = m - (nb - 1) * blen;
        return blen >= 5 && k <= blen - 1 && llen >= blen;          // (k <= blen - 1: the diagonal mask keeps every hit)
    };
    if (!cut()) return false;
    while (llen > PIECE_LAST_STEPS && nb < PIECE_NB) { ++nb; if (!cut()) return false; }
    {   // classes t >= PIECE_NT of the last-column rows (adapters of 50 bases and more): t + 1 regular body pieces inside
        // the class's shortest overlap serve them -- the body pieces are looked for everywhere, the read end included
        for (;;) {
            bool need_more = false;
            for (int i = 1; i <= m - 1; ++i) {
                const int t = thr_row[i];
                if (t >= PIECE_NT && std::min(nb - 1, i / blen) < t + 1) need_more = true;
            }
            if (!need_more) break;
            if (nb >= PIECE_NB) return false;
            ++nb;
            if (!cut()) return false;
        }
    }
    if (llen > PIECE_LAST_STEPS) return false;
    const int body_rows = m;
    for (int i = 0; i < m; ++i) if (piece_plane_of(codes[i]) < 0) return false;
    pp.blen = blen; pp.llen = llen; pp.m = m; pp.and_mode = and_mode ? 1 : 0; pp.nb = nb;
    pp.narrow = std::max(PIECE_NARROW, (rows + (m - rows) + 2 * k + 7) & ~7);
    if (pp.narrow > PIECE_WINDOW_MAX) return false;
    pp.window = pp.narrow > PIECE_WINDOW ? PIECE_WINDOW_MAX : PIECE_WINDOW;
    for (int p = 0; p < nb; ++p) {
        const int len = p == nb - 1 ? llen : blen;
        pp.plen[p] = len;
        pp.pshift[p] = p == nb - 1 ? m - blen : p * blen;            // 0 .. 32
        for (int s = 0; s < len; ++s) pp.scode[s] |= (uint32_t)piece_plane_of(codes[p * blen + len - 1 - s]) << (2 * p);
    }
    for (int i = 1; i < 32; ++i)
        for (int r = 0; r < i && r < m; ++r) { const int c = piece_plane_of(codes[r]); if (c >= 0) pp.xmask[i][c] |= 1u << (32 - i + r); }
    // Last-column candidates of the rows i < m (row m in the last column is a row-m candidate: a body piece):
    // thr_row[i] = floor(i e), or -1 for a row that is no candidate (below min_overlap / no STOP_WITHIN_SEQ1).
    // Rows with thr 0 are tested exactly; rows with thr t >= 1 form class t.
    int ilo[PIECE_NT], ihi[PIECE_NT];
    for (int t = 0; t < PIECE_NT; ++t) { ilo[t] = 0; ihi[t] = -1; }
    const int top = m - 1;
    for (int i = 1; i <= top; ++i) {
        const int t = thr_row[i];
        if (t < 0) continue;
        if (i > 1 && thr_row[i - 1] > t) return false;               // (monotone thresholds: floor(i e))
        if (t >= PIECE_NT) continue;                                 // (served by the body pieces: checked above)
        if (ihi[t] < 0) ilo[t] = i;
        ihi[t] = i;
    }
    pp.xlo = 1; pp.xhi = 0;
    if (ihi[0] >= 0) { pp.xlo = ilo[0]; pp.xhi = ihi[0]; }
    if (pp.xhi > 31) return false;
    int tmax = 0;
    for (int t = 1; t < PIECE_NT; ++t) if (ihi[t] >= 0) tmax = t;
    pp.tlen = 0;
    const int wl = (n - 1) >> 5;                                     // word of the read's last base = NW - 1 of the kernel
    pp.tw0 = std::max(0, wl - (PIECE_TAIL_WORDS - 1));
    if (tmax >= 1) {
        // piece length: t + 1 pieces must fit the shortest overlap of every class t
        int tlen = PIECE_STEPS;
        for (int t = 1; t <= tmax; ++t) if (ihi[t] >= 0) tlen = std::min(tlen, ilo[t] / (t + 1));
        if (tlen < 4) return false;                                  // (four bases: a chance hit per 256 positions and piece)
        pp.tlen = tlen;
        for (int u = 0; u < PIECE_NT; ++u) {
            if (u > tmax) continue;
            if ((u + 1) * tlen > body_rows) return false;
            for (int s = 0; s < tlen; ++s) pp.scode[s] |= (uint32_t)piece_plane_of(codes[u * tlen + tlen - 1 - s]) << (PIECE_TBIT + 2 * u);
            // ... and piece u serves the classes t >= max(1, u) whose first t + 1 pieces fit their shortest overlap.
            for (int t = std::max(1, u); t <= tmax; ++t) {
                if (ihi[t] < 0) continue;
                if ((t + 1) * tlen > ilo[t]) return false;
                // overlap of i rows, i in [ilo, ihi], ends on diagonal n - i; the piece's own diagonal is at most t
                // off: d = n - x, x in [ilo - t, ihi + t]; it then ends at position d + (u + 1) tlen - 1.
                for (int x = ilo[t] - t; x <= ihi[t] + t; ++x) {
                    const int e = n - x + (u + 1) * tlen - 1;
                    if (e < 0 || e > n - 1 || n - x < -t) continue;
                    const int w = (e >> 5) - pp.tw0;
                    if (w < 0) return false;                         // (cannot happen: x <= body_rows + 3 < 64)
                    pp.tmask[u][w] |= 1u << (e & 31);
                }
            }
        }
    }
    pp.sxlo = 1; pp.sxhi = 0;
    if (sr) {
        pp.sr = 1;
        pp.head_cols = m + k;
        // Suffix overlaps: the adapter's last L rows on the read's first bases, L = 1 .. m - 1 (L = m starts in row 0: a
        // body piece), a candidate when L >= min_overlap (_align.pyx:446: length = m + min(origin, 0)), cost <= thr[L].
        int slo[PIECE_NT], shi[PIECE_NT];
        for (int t = 0; t < PIECE_NT; ++t) { slo[t] = 0; shi[t] = -1; }
        for (int L = 1; L <= m - 1; ++L) {
            const int t = L >= min_overlap ? (int)thr[L] : -1;
            if (t < 0) continue;
            if (t >= PIECE_NT) {
                // served by body pieces: the last one and the regular ones in front of it, inside the last L rows
                const int inside = L >= llen ? 1 + std::min(nb - 1, (L - llen) / blen) : 0;
                if (inside < t + 1) return false;
                continue;
            }
            if (shi[t] < 0) slo[t] = L;
            shi[t] = L;
        }
        if (shi[0] >= 0) { pp.sxlo = slo[0]; pp.sxhi = shi[0]; }
        if (pp.sxhi > 31) return false;
        for (int L = 1; L < 32 && L <= m; ++L)
            for (int r = 0; r < L; ++r) { const int c = piece_plane_of(codes[m - L + r]); if (c >= 0) pp.xsmask[L][c] |= 1u << r; }
        int smax = 0;
        for (int t = 1; t < PIECE_NT; ++t) if (shi[t] >= 0) smax = t;
        if (smax >= 1) {
            int slen = PIECE_STEPS;
            for (int t = 1; t <= smax; ++t) if (shi[t] >= 0) slen = std::min(slen, slo[t] / (t + 1));
            if (slen < 4) return false;
            pp.slen = slen;
            for (int u = 0; u <= smax; ++u) {
                if ((u + 1) * slen > m) return false;
                for (int s = 0; s < slen; ++s) pp.scode2[s] |= (uint32_t)piece_plane_of(codes[m - u * slen - 1 - s]) << (2 * u);
                for (int t = std::max(1, u); t <= smax; ++t) {
                    if (shi[t] < 0) continue;
                    if ((t + 1) * slen > slo[t]) return false;
                    // overlap of L rows from column 0: piece u ends at position L - u slen - 1, at most t off
                    for (int L = slo[t]; L <= shi[t]; ++L)
                        for (int dl = -t; dl <= t; ++dl) {
                            const int e = L - u * slen - 1 + dl;
                            if (e - slen + 1 < 0 || e > n - 1) continue;
                            if ((e >> 5) >= PIECE_HEAD_WORDS) return false;      // (cannot happen: L + t <= 31 + 4)
                            pp.smask[u][e >> 5] |= 1u << (e & 31);
                        }
                }
            }
        }
    }
    pp.aonly = aonly ? 1 : 0;
    pp.steps = std::max(std::max(pp.llen, pp.tlen), pp.slen);
    pp.tail_cols = rows + k + (m - rows);                            // see "tail_cols" in DESIGN.md 3.2b: rows + k + T
    return true;
}
#endif  // __HIPCC_RTC__ (host side)

// ---- pass A, one lane ---------------------------------------------------------------------------------
// The four per-base masks of one read, NW words each: Y[c][w] bit b <=> base 32 w + b matches code index c.
template <int NW>
struct PieceMasks {
    uint32_t y[4][NW];
};

// pl[w][p]: plane p of word w.
template <int NW>
ATR_DEV void piece_eq_masks(const uint32_t (&pl)[NW][4], bool and_mode, PieceMasks<NW> &Y) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t p0 = pl[w][0], p1 = pl[w][1], p2 = pl[w][2], p3 = pl[w][3];
        if (and_mode) { Y.y[0][w] = p0; Y.y[1][w] = p1; Y.y[2][w] = p2; Y.y[3][w] = p3; }
        else {                                                       // byte equality: exactly this plane and no other
            Y.y[0][w] = p0 & ~(p1 | p2 | p3); Y.y[1][w] = p1 & ~(p0 | p2 | p3);
            Y.y[2][w] = p2 & ~(p0 | p1 | p3); Y.y[3][w] = p3 & ~(p0 | p1 | p2);
        }
    }
}

// All four masks one position up (bit b of word w -> bit b + 1, across the words): four add-with-carry chains,
// interleaved so that each fills the others' carry wait states.
template <int NW>
ATR_DEV void piece_shift_masks(PieceMasks<NW> &Y) {
#ifdef ATR_HOST_EMU
    for (int c = 0; c < 4; ++c) {
        uint32_t carry = 0;
        for (int w = 0; w < NW; ++w) { const uint32_t v = Y.y[c][w]; Y.y[c][w] = (v << 1) | carry; carry = v >> 31; }
    }
#else
    uint64_t c0, c1, c2, c3;
    asm volatile("v_add_co_u32 %0, %4, %0, %0\n\tv_add_co_u32 %1, %5, %1, %1\n\t"
                 "v_add_co_u32 %2, %6, %2, %2\n\tv_add_co_u32 %3, %7, %3, %3"
                 : "+v"(Y.y[0][0]), "+v"(Y.y[1][0]), "+v"(Y.y[2][0]), "+v"(Y.y[3][0]), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3));
#pragma unroll
    for (int w = 1; w < NW; ++w)
        asm volatile("v_addc_co_u32 %0, %4, %0, %0, %4\n\tv_addc_co_u32 %1, %5, %1, %1, %5\n\t"
                     "v_addc_co_u32 %2, %6, %2, %2, %6\n\tv_addc_co_u32 %3, %7, %3, %3, %7"
                     : "+v"(Y.y[0][w]), "+v"(Y.y[1][w]), "+v"(Y.y[2][w]), "+v"(Y.y[3][w]), "+s"(c0), "+s"(c1), "+s"(c2), "+s"(c3));
#endif
}

// acc[i] &= Y[code][W0 + i], i < N: the code is wave-uniform, so this is a scalar branch to one of four
// straight-line blocks (no register indexing: an indexed read costs 6 - 8 cycles, the branch overlaps with the
// other waves' VALU work).
#ifndef ATR_HOST_EMU
// acc[i] &= (y0, y1, y2, y3)[code][i], i < G, code = bits BIT + 1 : BIT of the step's code word cw (wave-uniform): one
// asm block with the four-way scalar branch INSIDE, so that the compiler sees a straight-line update of the
// accumulators in place (with the branch in C++ it copies the accumulators into fresh registers in front of every
// case: five v_mov per five v_and).  s_bitcmp1 tests the code bits where they sit: 4 - 5 scalar instructions a term.
template <int G, int BIT>
__device__ __forceinline__ void piece_and_group(uint32_t *a, const uint32_t *y0, const uint32_t *y1, const uint32_t *y2,
                                                const uint32_t *y3, uint32_t cw) {
    static_assert(G >= 1 && G <= 5, "at most 5 accumulators + 20 masks + the code word: 28 asm operands");
    if constexpr (G == 1) {
        asm volatile("s_bitcmp1_b32 %5, %7\n\t"
                     "s_cbranch_scc1 .Lpthi_%=\n\t"
                     "s_bitcmp1_b32 %5, %6\n\t"
                     "s_cbranch_scc1 .Lpt1_%=\n\t"
                     "v_and_b32 %0, %0, %1\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt1_%=:\n\t"
                     "v_and_b32 %0, %0, %2\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpthi_%=:\n\t"
                     "s_bitcmp1_b32 %5, %6\n\t"
                     "s_cbranch_scc1 .Lpt3_%=\n\t"
                     "v_and_b32 %0, %0, %3\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt3_%=:\n\t"
                     "v_and_b32 %0, %0, %4\n\t"
                     ".Lptend_%=:"
                     : "+v"(a[0])
                     : "v"(y0[0]), "v"(y1[0]), "v"(y2[0]), "v"(y3[0]), "s"(cw), "n"(BIT), "n"(BIT + 1) : "scc");
    }
    else if constexpr (G == 2) {
        asm volatile("s_bitcmp1_b32 %10, %12\n\t"
                     "s_cbranch_scc1 .Lpthi_%=\n\t"
                     "s_bitcmp1_b32 %10, %11\n\t"
                     "s_cbranch_scc1 .Lpt1_%=\n\t"
                     "v_and_b32 %0, %0, %2\n\t"
                     "v_and_b32 %1, %1, %3\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt1_%=:\n\t"
                     "v_and_b32 %0, %0, %4\n\t"
                     "v_and_b32 %1, %1, %5\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpthi_%=:\n\t"
                     "s_bitcmp1_b32 %10, %11\n\t"
                     "s_cbranch_scc1 .Lpt3_%=\n\t"
                     "v_and_b32 %0, %0, %6\n\t"
                     "v_and_b32 %1, %1, %7\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt3_%=:\n\t"
                     "v_and_b32 %0, %0, %8\n\t"
                     "v_and_b32 %1, %1, %9\n\t"
                     ".Lptend_%=:"
                     : "+v"(a[0]), "+v"(a[1])
                     : "v"(y0[0]), "v"(y0[1]), "v"(y1[0]), "v"(y1[1]), "v"(y2[0]), "v"(y2[1]), "v"(y3[0]), "v"(y3[1]), "s"(cw), "n"(BIT), "n"(BIT + 1) : "scc");
    }
    else if constexpr (G == 3) {
        asm volatile("s_bitcmp1_b32 %15, %17\n\t"
                     "s_cbranch_scc1 .Lpthi_%=\n\t"
                     "s_bitcmp1_b32 %15, %16\n\t"
                     "s_cbranch_scc1 .Lpt1_%=\n\t"
                     "v_and_b32 %0, %0, %3\n\t"
                     "v_and_b32 %1, %1, %4\n\t"
                     "v_and_b32 %2, %2, %5\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt1_%=:\n\t"
                     "v_and_b32 %0, %0, %6\n\t"
                     "v_and_b32 %1, %1, %7\n\t"
                     "v_and_b32 %2, %2, %8\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpthi_%=:\n\t"
                     "s_bitcmp1_b32 %15, %16\n\t"
                     "s_cbranch_scc1 .Lpt3_%=\n\t"
                     "v_and_b32 %0, %0, %9\n\t"
                     "v_and_b32 %1, %1, %10\n\t"
                     "v_and_b32 %2, %2, %11\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt3_%=:\n\t"
                     "v_and_b32 %0, %0, %12\n\t"
                     "v_and_b32 %1, %1, %13\n\t"
                     "v_and_b32 %2, %2, %14\n\t"
                     ".Lptend_%=:"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])
                     : "v"(y0[0]), "v"(y0[1]), "v"(y0[2]), "v"(y1[0]), "v"(y1[1]), "v"(y1[2]), "v"(y2[0]), "v"(y2[1]), "v"(y2[2]), "v"(y3[0]), "v"(y3[1]), "v"(y3[2]), "s"(cw), "n"(BIT), "n"(BIT + 1) : "scc");
    }
    else if constexpr (G == 4) {
        asm volatile("s_bitcmp1_b32 %20, %22\n\t"
                     "s_cbranch_scc1 .Lpthi_%=\n\t"
                     "s_bitcmp1_b32 %20, %21\n\t"
                     "s_cbranch_scc1 .Lpt1_%=\n\t"
                     "v_and_b32 %0, %0, %4\n\t"
                     "v_and_b32 %1, %1, %5\n\t"
                     "v_and_b32 %2, %2, %6\n\t"
                     "v_and_b32 %3, %3, %7\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt1_%=:\n\t"
                     "v_and_b32 %0, %0, %8\n\t"
                     "v_and_b32 %1, %1, %9\n\t"
                     "v_and_b32 %2, %2, %10\n\t"
                     "v_and_b32 %3, %3, %11\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpthi_%=:\n\t"
                     "s_bitcmp1_b32 %20, %21\n\t"
                     "s_cbranch_scc1 .Lpt3_%=\n\t"
                     "v_and_b32 %0, %0, %12\n\t"
                     "v_and_b32 %1, %1, %13\n\t"
                     "v_and_b32 %2, %2, %14\n\t"
                     "v_and_b32 %3, %3, %15\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt3_%=:\n\t"
                     "v_and_b32 %0, %0, %16\n\t"
                     "v_and_b32 %1, %1, %17\n\t"
                     "v_and_b32 %2, %2, %18\n\t"
                     "v_and_b32 %3, %3, %19\n\t"
                     ".Lptend_%=:"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                     : "v"(y0[0]), "v"(y0[1]), "v"(y0[2]), "v"(y0[3]), "v"(y1[0]), "v"(y1[1]), "v"(y1[2]), "v"(y1[3]), "v"(y2[0]), "v"(y2[1]), "v"(y2[2]), "v"(y2[3]), "v"(y3[0]), "v"(y3[1]), "v"(y3[2]), "v"(y3[3]), "s"(cw), "n"(BIT), "n"(BIT + 1) : "scc");
    }
    else if constexpr (G == 5) {
        asm volatile("s_bitcmp1_b32 %25, %27\n\t"
                     "s_cbranch_scc1 .Lpthi_%=\n\t"
                     "s_bitcmp1_b32 %25, %26\n\t"
                     "s_cbranch_scc1 .Lpt1_%=\n\t"
                     "v_and_b32 %0, %0, %5\n\t"
                     "v_and_b32 %1, %1, %6\n\t"
                     "v_and_b32 %2, %2, %7\n\t"
                     "v_and_b32 %3, %3, %8\n\t"
                     "v_and_b32 %4, %4, %9\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt1_%=:\n\t"
                     "v_and_b32 %0, %0, %10\n\t"
                     "v_and_b32 %1, %1, %11\n\t"
                     "v_and_b32 %2, %2, %12\n\t"
                     "v_and_b32 %3, %3, %13\n\t"
                     "v_and_b32 %4, %4, %14\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpthi_%=:\n\t"
                     "s_bitcmp1_b32 %25, %26\n\t"
                     "s_cbranch_scc1 .Lpt3_%=\n\t"
                     "v_and_b32 %0, %0, %15\n\t"
                     "v_and_b32 %1, %1, %16\n\t"
                     "v_and_b32 %2, %2, %17\n\t"
                     "v_and_b32 %3, %3, %18\n\t"
                     "v_and_b32 %4, %4, %19\n\t"
                     "s_branch .Lptend_%=\n\t"
                     ".Lpt3_%=:\n\t"
                     "v_and_b32 %0, %0, %20\n\t"
                     "v_and_b32 %1, %1, %21\n\t"
                     "v_and_b32 %2, %2, %22\n\t"
                     "v_and_b32 %3, %3, %23\n\t"
                     "v_and_b32 %4, %4, %24\n\t"
                     ".Lptend_%=:"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                     : "v"(y0[0]), "v"(y0[1]), "v"(y0[2]), "v"(y0[3]), "v"(y0[4]), "v"(y1[0]), "v"(y1[1]), "v"(y1[2]), "v"(y1[3]), "v"(y1[4]), "v"(y2[0]), "v"(y2[1]), "v"(y2[2]), "v"(y2[3]), "v"(y2[4]), "v"(y3[0]), "v"(y3[1]), "v"(y3[2]), "v"(y3[3]), "v"(y3[4]), "s"(cw), "n"(BIT), "n"(BIT + 1) : "scc");
    }
}
#endif

template <int NW, int N, int W0, int BIT>
ATR_DEV void piece_and_term(uint32_t (&acc)[N], const PieceMasks<NW> &Y, uint32_t cw) {
    static_assert(W0 + N <= NW, "words of the read");
#ifdef ATR_HOST_EMU
    for (int i = 0; i < N; ++i) acc[i] &= Y.y[(cw >> BIT) & 3u][W0 + i];
#else
    // groups of at most five words (the asm operand limit)
    constexpr int G0 = N <= 5 ? N : (N + 1) / 2, G1 = N - G0;
    static_assert(G0 <= 5 && G1 <= 5, "reads of up to 320 bases");
    piece_and_group<G0, BIT>(&acc[0], &Y.y[0][W0], &Y.y[1][W0], &Y.y[2][W0], &Y.y[3][W0], cw);
    if constexpr (G1 > 0) piece_and_group<G1, BIT>(&acc[G0], &Y.y[0][W0 + G0], &Y.y[1][W0 + G0], &Y.y[2][W0 + G0], &Y.y[3][W0 + G0], cw);
#endif
}

ATR_DEV int piece_uniform(int v) {
#ifdef ATR_HOST_EMU
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// (hi:lo) >> sh, low word; sh in 0 .. 31
ATR_DEV uint32_t piece_funnel(uint32_t hi, uint32_t lo, int sh) {
#ifdef ATR_HOST_EMU
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
#else
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#endif
}

// What pass A knows about one read.
struct PieceScan {
    int j_exact;                             // first column with the whole adapter verbatim before it (0: none): the
                                             // reference's early exit, result (0, m, j - m, j, m, 0) if m >= min_overlap
    int j_s, j_e;                            // columns j_s + 1 .. j_e hold every acceptable cell and its traceback
    bool flagged;                            // false: the result is None
    bool tail;                               // a read-end condition holds (the window then ends at n)
    bool head;                               // START_WITHIN_SEQ1: a read-start condition holds (the window then starts at column 0;
                                             // the caller widens it: its columns are the read's own, j_s / j_e may be a moved read's)
};

// What pass B (or the full sweep) is asked to do with a read, in the read's OWN columns.  back: positions pass A saw the read
// moved up by (a ragged batch), nr: its length.  START_WITHIN_SEQ1: a read-start condition, or a body piece that allows a
// row-m candidate within m + k columns of the read start (only those can be reached from column 0 with <= k errors), makes
// the window begin in column 0, where the sweep starts from the all-zero column; a read so short that one alignment could
// touch both column 0 and the last column takes the full sweep whatever pass A found.
struct PieceTask {
    int j_e, need;                           // the window ends at column j_e and holds `need` columns
    bool flagged, full;                      // something to decide; ... by the full sweep only
};
ATR_DEV PieceTask piece_task(const PieceScan &S, int back, int nr, bool sr, int m, int k, int head_cols) {
    int js = atr_max(S.j_s, back) - back, je = S.j_e - back;
    PieceTask t;
    t.flagged = S.flagged;
    t.full = false;
    if (sr) {
        if (nr <= m + k) { t.flagged = true; t.full = true; }
        if (S.head) { js = 0; je = atr_max(je, atr_min(nr, head_cols)); }
        else if (js <= k) js = 0;
    }
    t.j_e = atr_max(je, 0);                                          // (nothing flagged: an empty window at column 0)
    t.need = atr_max(je - js, 0);
    return t;
}

// Pass A for one lane: the planes of a read of n bases (NW = ceil(n / 32) words) -> PieceScan.  twp[p]: plane p
// of the read's last 32 positions (bit 31 = the last base; zeros before the read).  mf = rows swept by pass B
// (FilterParams::rows), T = m - mf, k as Uniform::k.  n is wave-uniform (equal-length batch).
// NBMAX: the body pieces this instantiation holds accumulators for (pp.nb <= NBMAX): the generic kernels of the 64-column
// window keep five (eight of them spilled 37 registers at NW = 5)
// hpl: the read's first plane words where the read STARTS at position 0 (pl itself unless the caller moved a ragged read to
// the end of its words): what the read-start conditions of a START_WITHIN_SEQ1 aligner look at.
template <int NW, int NBMAX = PIECE_NB>
ATR_DEV PieceScan piece_scan(const PieceParams &pp, const uint32_t (&pl)[NW][4], const uint32_t (&twp)[4], int n, int mf,
                             int T, int k, const uint32_t (*hpl)[4] = nullptr) {
    static_assert(NBMAX >= 4 && NBMAX <= PIECE_NB, "body pieces");
    constexpr int TWN = NW < PIECE_TAIL_WORDS ? NW : PIECE_TAIL_WORDS, TW0 = NW - TWN;
    constexpr int HWN = NW < PIECE_HEAD_WORDS ? NW : PIECE_HEAD_WORDS;
    PieceMasks<NW> Y;
    const bool and_mode = piece_uniform(pp.and_mode) != 0;
    const int xlo = piece_uniform(pp.xlo), xhi = piece_uniform(pp.xhi), pm = piece_uniform(pp.m), pblen = piece_uniform(pp.blen);
    piece_eq_masks<NW>(pl, and_mode, Y);

    // (0) START_WITHIN_SEQ1: the suffix overlaps that must be exact -- the adapter's last L rows are the read's first L bases
    const bool sr = piece_uniform(pp.sr) != 0 && hpl != nullptr;
    const int slen = sr ? piece_uniform(pp.slen) : 0;
    bool head = false;
    PieceMasks<HWN> YH;
    uint32_t socc[PIECE_NT][HWN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int w = 0; w < HWN; ++w) YH.y[c][w] = 0u;
#pragma unroll
    for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
        for (int w = 0; w < HWN; ++w) socc[u][w] = 0u;
    if (sr) {
        uint32_t hp[HWN][4];
#pragma unroll
        for (int w = 0; w < HWN; ++w)
#pragma unroll
            for (int c = 0; c < 4; ++c) hp[w][c] = hpl[w][c];
        piece_eq_masks<HWN>(hp, and_mode, YH);
        const int sxlo = piece_uniform(pp.sxlo), sxhi = piece_uniform(pp.sxhi);
        for (int L = sxlo; L <= sxhi; ++L) {                   // wave-uniform trip count; L <= 31
            const uint32_t mw = (YH.y[0][0] & pp.xsmask[L][0]) | (YH.y[1][0] & pp.xsmask[L][1]) | (YH.y[2][0] & pp.xsmask[L][2]) |
                                (YH.y[3][0] & pp.xsmask[L][3]);
            head = head || mw == ((1u << L) - 1u);
        }
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < HWN; ++w) socc[u][w] = pp.smask[u][w];
    }

    // (1) overlaps that must be exact: rows [0, i) against the last i bases, i in [xlo, xhi]: row r of an overlap
    //     of i sits at bit 32 - i + r of the read's last 32 positions.
    bool tail = false;
    if (xhi >= xlo) {
        uint32_t tw[4];
        if (and_mode) { tw[0] = twp[0]; tw[1] = twp[1]; tw[2] = twp[2]; tw[3] = twp[3]; }
        else {
            tw[0] = twp[0] & ~(twp[1] | twp[2] | twp[3]); tw[1] = twp[1] & ~(twp[0] | twp[2] | twp[3]);
            tw[2] = twp[2] & ~(twp[0] | twp[1] | twp[3]); tw[3] = twp[3] & ~(twp[0] | twp[1] | twp[2]);
        }
        for (int i = xlo; i <= xhi; ++i) {                     // wave-uniform trip count
            // every row of the overlap matches <=> the four (mask & rows-of-that-code) words together fill bits 32 - i .. 31
            const uint32_t mw = (tw[0] & pp.xmask[i][0]) | (tw[1] & pp.xmask[i][1]) | (tw[2] & pp.xmask[i][2]) | (tw[3] & pp.xmask[i][3]);
            tail = tail || (i <= n && mw == (~0u << (32 - i)));
        }
    }

    // (2) the pieces: body pieces on every word, read-end pieces on the read's last TWN words only
    uint32_t occ[NBMAX][NW], tocc[PIECE_NT][TWN];
#pragma unroll
    for (int p = 0; p < NBMAX; ++p)
#pragma unroll
        for (int w = 0; w < NW; ++w) occ[p][w] = ~0u;
#pragma unroll
    for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
        for (int w = 0; w < TWN; ++w) tocc[u][w] = pp.tmask[u][w];
    // (wave-uniform parameters: said so, for callers whose PieceParams sit behind a pointer)
    int plen[PIECE_NB];
#pragma unroll
    for (int p = 0; p < PIECE_NB; ++p) plen[p] = p < NBMAX ? piece_uniform(pp.plen[p]) : 0;
    const int steps = piece_uniform(pp.steps), tlen = piece_uniform(pp.tlen);
    for (int s = 0; s < steps; ++s) {                                // wave-uniform
        const uint32_t cw = (uint32_t)piece_uniform((int)pp.scode[s]);
        if (s < plen[0]) piece_and_term<NW, NW, 0, 0>(occ[0], Y, cw);
        if (s < plen[1]) piece_and_term<NW, NW, 0, 2>(occ[1], Y, cw);
        if (s < plen[2]) piece_and_term<NW, NW, 0, 4>(occ[2], Y, cw);
        if (s < plen[3]) piece_and_term<NW, NW, 0, 6>(occ[3], Y, cw);
        if constexpr (NBMAX > 4) { if (s < plen[4]) piece_and_term<NW, NW, 0, 8>(occ[4 < NBMAX ? 4 : 0], Y, cw); }
        if constexpr (NBMAX > 5) {
            if (s < plen[5]) piece_and_term<NW, NW, 0, 10>(occ[5 < NBMAX ? 5 : 0], Y, cw);
            if (s < plen[6]) piece_and_term<NW, NW, 0, 12>(occ[6 < NBMAX ? 6 : 0], Y, cw);
            if (s < plen[7]) piece_and_term<NW, NW, 0, 14>(occ[7 < NBMAX ? 7 : 0], Y, cw);
        }
        if (s < tlen) {
            piece_and_term<NW, TWN, TW0, PIECE_TBIT + 0>(tocc[0], Y, cw);
            piece_and_term<NW, TWN, TW0, PIECE_TBIT + 2>(tocc[1], Y, cw);
            piece_and_term<NW, TWN, TW0, PIECE_TBIT + 4>(tocc[2], Y, cw);
            piece_and_term<NW, TWN, TW0, PIECE_TBIT + 6>(tocc[3], Y, cw);
            piece_and_term<NW, TWN, TW0, PIECE_TBIT + 8>(tocc[4], Y, cw);
        }
        if (s < slen) {                                              // (wave-uniform; 0 without START_WITHIN_SEQ1)
            const uint32_t cw2 = (uint32_t)piece_uniform((int)pp.scode2[s]);
            piece_and_term<HWN, HWN, 0, 0>(socc[0], YH, cw2);
            piece_and_term<HWN, HWN, 0, 2>(socc[1], YH, cw2);
            piece_and_term<HWN, HWN, 0, 4>(socc[2], YH, cw2);
            piece_and_term<HWN, HWN, 0, 6>(socc[3], YH, cw2);
            piece_and_term<HWN, HWN, 0, 8>(socc[4], YH, cw2);
            if (s + 1 < slen) piece_shift_masks<HWN>(YH);
        }
        if (s + 1 < steps) piece_shift_masks<NW>(Y);
    }
    if (slen > 0) {
        uint32_t any = 0u;
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < HWN; ++w) any |= socc[u][w];
        head = head || any != 0u;
    }
    // (a piece cannot end before its own length: after s shifts the bits below s are zero)
    if (tlen > 0) {
        uint32_t any = 0u;
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < TWN; ++w) any |= tocc[u][w];
        tail = tail || any != 0u;
    }

    // (3) diagonals with a body piece: piece p < 3 ending at position e sits on diagonal d = e - (p + 1) blen + 1, the
    //     last one on d = e - m + 1.  al[p] = the piece's hits moved to piece 0's place: bit b <=> on diagonal
    //     b - (blen - 1)   (k <= blen - 1: no possible diagonal is lost).  dm = any piece; pf = all four on one
    //     diagonal = the adapter verbatim.
    uint32_t dm[NW], pf[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) dm[w] = pf[w] = occ[0][w];
#pragma unroll
    for (int p = 1; p < NBMAX; ++p) {
        if (p >= piece_uniform(pp.nb)) continue;                     // (wave-uniform: four pieces unless k = 4)
        const int sh = piece_uniform(pp.pshift[p]);                  // 0 .. 32 (wave-uniform)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            // words w, w + 1 (+ 1 when sh == 32: the funnel then takes the next pair with shift 0)
            const uint32_t lo = sh >= 32 ? (w + 1 < NW ? occ[p][w + 1 < NW ? w + 1 : w] : 0u) : occ[p][w];
            const uint32_t hi = sh >= 32 ? (w + 2 < NW ? occ[p][w + 2 < NW ? w + 2 : w] : 0u) : (w + 1 < NW ? occ[p][w + 1 < NW ? w + 1 : w] : 0u);
            const uint32_t al = piece_funnel(hi, lo, sh & 31);
            dm[w] |= al;
            pf[w] &= al;
        }
    }
    int b_first = -1, b_last = -1, z_first = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) if (dm[w] != 0u) b_first = 32 * w + atr_ctz(dm[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w) if (dm[w] != 0u) b_last = 32 * w + 31 - atr_clz(dm[w]);
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) if (pf[w] != 0u) z_first = 32 * w + atr_ctz(pf[w]);

    const int tail_cols = piece_uniform(pp.tail_cols);
    PieceScan S;
    S.tail = tail;
    S.head = head;
    S.flagged = tail || head || b_first >= 0;
    S.j_s = 0; S.j_e = 0;
    // the first zero-cost column of row m (all m rows verbatim on diagonal d: column d + m): where the reference stops
    S.j_exact = z_first >= 0 ? z_first - (pblen - 1) + pm : 0;
    if (b_first >= 0) {
        const int d_min = b_first - (pblen - 1), d_max = b_last - (pblen - 1);
        // a traceback through a piece on diagonal d leaves row 0 at a column >= d - k and passes row mf at a column
        // <= mf + d + k; + T: the bases filter_decide compares with the adapter's tail rows (NARROW mode).
        // If mf + T + d_max + k >= n a last-column cell of a row beyond the pieces' rows is possible: the window
        // then ends at n anyway and the last column is evaluated.
        S.j_s = atr_max(0, d_min - k);
        S.j_e = atr_min(n, mf + d_max + k + T);
    }
    if (tail) {
        S.j_s = b_first >= 0 ? atr_min(S.j_s, atr_max(0, n - tail_cols)) : atr_max(0, n - tail_cols);
        S.j_e = n;
    }
    return S;
}

#ifdef ATR_SPEC
#include "piece_spec_config.h"   // written by the host for this aligner (jit.hpp): atr::spec::{N, P, FP, PP, ROWC}
// ---- pass A for ONE aligner, compiled at run time (jit.hpp, piece_spec.hip) ------------------------------------
// atr::spec::PP / FP / P / N are constexpr objects written by the host for this aligner and read length (the
// generated piece_spec_config.h): every piece code, end mask and overlap mask is a literal, the step loop is
// straight-line code -- acc &= Y[c] with c known -- and the 54 scalar dispatches per tile of the generic kernel
// (s_bitcmp1 + branch per term: 40 % of its issue time) are gone, as are the ~700 bytes of kernel
// argument that kept 116 SGPRs spilled.
template <int V> struct PieceIC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void piece_static_for(F &&f) {
    if constexpr (I < N) { f(PieceIC<I>{}); piece_static_for<I + 1, N>(f); }
}

template <int NW>
ATR_DEV PieceScan piece_scan_spec(const uint32_t (&pl)[NW][4], const uint32_t (&twp)[4], int mf, int T, int k,
                                  const uint32_t (*hpl)[4] = nullptr) {
    constexpr int TWN = NW < PIECE_TAIL_WORDS ? NW : PIECE_TAIL_WORDS, TW0 = NW - TWN;
    constexpr int HWN = NW < PIECE_HEAD_WORDS ? NW : PIECE_HEAD_WORDS;
    constexpr int n = spec::N;
    PieceMasks<NW> Y;
    piece_eq_masks<NW>(pl, spec::PP.and_mode != 0, Y);

    // (0) START_WITHIN_SEQ1: the read-start conditions (piece_scan), on the read's first words where it starts at position 0
    bool head = false;
    PieceMasks<HWN> YH;
    uint32_t socc[PIECE_NT][HWN];
    if constexpr (spec::PP.sr != 0) {
        uint32_t hp[HWN][4];
#pragma unroll
        for (int w = 0; w < HWN; ++w)
#pragma unroll
            for (int c = 0; c < 4; ++c) hp[w][c] = hpl[w][c];
        piece_eq_masks<HWN>(hp, spec::PP.and_mode != 0, YH);
        // W_{L + 1} = eq[row m - L - 1] & (W_L >> 1): bit 0 <=> the first L + 1 bases are the adapter's last L + 1 rows
        if constexpr (spec::PP.sxhi >= spec::PP.sxlo) {
            uint32_t x = ~0u, any = 0u;
            piece_static_for<0, spec::PP.sxhi>([&](auto ic) {
                constexpr int L = decltype(ic)::value;                   // row m - L - 1 joins: W_{L + 1}
                x = (x >> 1) & YH.y[spec::ROWC[spec::PP.m - L - 1]][0];
                if constexpr (L + 1 >= spec::PP.sxlo) any |= x;
            });
            head = (any & 1u) != 0u;
        }
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < HWN; ++w) socc[u][w] = spec::PP.smask[u][w];
    }

    // (1) overlaps that must be exact.  X_i = AND_{r < i} (tw[c_r] << (i - 1 - r)) has bit 31 set iff the last i bases
    //     are rows 0 .. i - 1; X_{i + 1} = (X_i << 1) & tw[c_i]: one add and one and per overlap length.
    bool tail = false;
    if constexpr (spec::PP.xhi >= spec::PP.xlo) {
        uint32_t tw[4];
        if constexpr (spec::PP.and_mode != 0) { tw[0] = twp[0]; tw[1] = twp[1]; tw[2] = twp[2]; tw[3] = twp[3]; }
        else {
            tw[0] = twp[0] & ~(twp[1] | twp[2] | twp[3]); tw[1] = twp[1] & ~(twp[0] | twp[2] | twp[3]);
            tw[2] = twp[2] & ~(twp[0] | twp[1] | twp[3]); tw[3] = twp[3] & ~(twp[0] | twp[1] | twp[2]);
        }
        uint32_t x = ~0u, any = 0u;
        piece_static_for<0, spec::PP.xhi>([&](auto ic) {
            constexpr int i = decltype(ic)::value;                   // row i joins: X_{i + 1}
            x = (x + x) & tw[spec::ROWC[i]];
            if constexpr (i + 1 >= spec::PP.xlo && i + 1 <= n) any |= x;
        });
        tail = (any >> 31) != 0u;
    }

    // (2) the pieces
    uint32_t occ[PIECE_NB][NW], tocc[PIECE_NT][TWN];
#pragma unroll
    for (int p = 0; p < PIECE_NB; ++p)
#pragma unroll
        for (int w = 0; w < NW; ++w) occ[p][w] = ~0u;
#pragma unroll
    for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
        for (int w = 0; w < TWN; ++w) tocc[u][w] = spec::PP.tmask[u][w];
    // Every term is a volatile one-instruction asm, in source order: left to itself the scheduler moves the ands of a
    // step behind the in-place shift that follows them, keeps copies of every older mask and spills a hundred
    // registers (round 5, first try).  In this order the live set is Y + occ + tocc = 52 registers.
    const auto term = [](uint32_t &acc, uint32_t y, bool first) {
        if (first) acc = y;                                          // (occ starts as all ones: the first term is a copy;
        else asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc) : "v"(y));   //  a read-end piece starts from its end mask)
    };
    piece_static_for<0, spec::PP.steps>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr uint32_t cw = spec::PP.scode[s];
        // (the regular pieces word by word, then the last piece: the order rounds 5's kernel was tuned in)
#pragma unroll
        for (int w = 0; w < NW; ++w)
            piece_static_for<0, PIECE_NB>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if constexpr (p < spec::PP.nb - 1 && s < spec::PP.plen[p]) term(occ[p][w], Y.y[(cw >> (2 * p)) & 3u][w], s == 0);
            });
        {
            constexpr int p = spec::PP.nb - 1;
            if constexpr (s < spec::PP.plen[p]) {
#pragma unroll
                for (int w = 0; w < NW; ++w) term(occ[p][w], Y.y[(cw >> (2 * p)) & 3u][w], s == 0);
            }
        }
        if constexpr (s < spec::PP.tlen) {
            piece_static_for<0, PIECE_NT>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr ((spec::PP.tmask[u][0] | spec::PP.tmask[u][1] | spec::PP.tmask[u][2]) != 0u) {
#pragma unroll
                    for (int w = 0; w < TWN; ++w) term(tocc[u][w], Y.y[(cw >> (PIECE_TBIT + 2 * u)) & 3u][TW0 + w], false);
                }
            });
        }
        if constexpr (spec::PP.sr != 0 && s < spec::PP.slen) {
            constexpr uint32_t cw2 = spec::PP.scode2[s];
            piece_static_for<0, PIECE_NT>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr ((spec::PP.smask[u][0] | spec::PP.smask[u][1]) != 0u) {
#pragma unroll
                    for (int w = 0; w < HWN; ++w) term(socc[u][w], YH.y[(cw2 >> (2 * u)) & 3u][w], false);
                }
            });
            if constexpr (s + 1 < spec::PP.slen) piece_shift_masks<HWN>(YH);
        }
        if constexpr (s + 1 < spec::PP.steps) piece_shift_masks<NW>(Y);
    });
    if constexpr (spec::PP.sr != 0 && spec::PP.slen > 0) {
        uint32_t any = 0u;
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < HWN; ++w) any |= socc[u][w];
        head = head || any != 0u;
    }
    if constexpr (spec::PP.tlen > 0) {
        uint32_t any = 0u;
#pragma unroll
        for (int u = 0; u < PIECE_NT; ++u)
#pragma unroll
            for (int w = 0; w < TWN; ++w) any |= tocc[u][w];
        tail = tail || any != 0u;
    }

    // (3) diagonals with a body piece (as piece_scan; the alignment shifts are literals)
    uint32_t dm[NW], pf[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) dm[w] = pf[w] = occ[0][w];
    piece_static_for<1, spec::PP.nb>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int sh = spec::PP.pshift[p];                       // 0 .. 32
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t lo = sh >= 32 ? (w + 1 < NW ? occ[p][w + 1 < NW ? w + 1 : w] : 0u) : occ[p][w];
            const uint32_t hi = sh >= 32 ? (w + 2 < NW ? occ[p][w + 2 < NW ? w + 2 : w] : 0u) : (w + 1 < NW ? occ[p][w + 1 < NW ? w + 1 : w] : 0u);
            const uint32_t al = (sh & 31) ? __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(sh & 31)) : lo;
            dm[w] |= al;
            pf[w] &= al;
        }
    });
    int b_first = -1, b_last = -1, z_first = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) if (dm[w] != 0u) b_first = 32 * w + atr_ctz(dm[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w) if (dm[w] != 0u) b_last = 32 * w + 31 - atr_clz(dm[w]);
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) if (pf[w] != 0u) z_first = 32 * w + atr_ctz(pf[w]);

    PieceScan S;
    S.tail = tail;
    S.head = head;
    S.flagged = tail || head || b_first >= 0;
    S.j_s = 0; S.j_e = 0;
    S.j_exact = z_first >= 0 ? z_first - (spec::PP.blen - 1) + spec::PP.m : 0;
    if (b_first >= 0) {
        const int d_min = b_first - (spec::PP.blen - 1), d_max = b_last - (spec::PP.blen - 1);
        S.j_s = atr_max(0, d_min - k);
        S.j_e = atr_min(n, mf + d_max + k + T);
    }
    if (tail) {
        S.j_s = b_first >= 0 ? atr_min(S.j_s, atr_max(0, n - spec::PP.tail_cols)) : atr_max(0, n - spec::PP.tail_cols);
        S.j_e = n;
    }
    return S;
}
#endif  // ATR_SPEC

// ---- planes -> 4-bit codes ------------------------------------------------------------------------------
// spread[t][b]: bit i of the byte b at bit 4 i + t.  The nibble dword of eight bases = the OR of the four
// planes' bytes looked up in the four tables.
#ifndef __HIPCC_RTC__
inline void piece_spread_tables(uint32_t (*tab)[256]) {
    for (int t = 0; t < 4; ++t)
        for (int b = 0; b < 256; ++b) {
            uint32_t v = 0;
            for (int i = 0; i < 8; ++i) if (b & (1 << i)) v |= 1u << (4 * i + t);
            tab[t][b] = v;
        }
}
#endif
// byte q (0 .. 3) of the four plane words -> the dword of the eight 4-bit codes
ATR_DEV uint32_t piece_nibbles(const uint32_t (*spread)[256], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, int q) {
    return spread[0][(p0 >> (8 * q)) & 255u] | spread[1][(p1 >> (8 * q)) & 255u] |
           spread[2][(p2 >> (8 * q)) & 255u] | spread[3][(p3 >> (8 * q)) & 255u];
}

#ifndef ATR_HOST_EMU
// the four tables in LDS, filled by the whole block (256 threads; __syncthreads() afterwards)
__device__ __forceinline__ void piece_spread_fill(uint32_t (*tab)[256]) {
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const int t = i >> 8, b = i & 255;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v |= (uint32_t)((b >> k) & 1) << (4 * k + t);
        tab[t][b] = v;
    }
}
#endif

// ---- reads in the plane64 layout, for the DP kernels ------------------------------------------------------
// chunk c of read r: four plane words at packed[((r >> 6) nchunks + c) 64 + (r & 63)] (uint4).
// The nibble dword z8 (bases 8 z8 + 1 ..) of a read; 0 beyond the read.
ATR_DEV uint32_t read_dword_planes(const uint32_t *q4, int nchunks, int z8, const uint32_t (*spread)[256]) {
    if (z8 < 0 || z8 >= nchunks * 4) return 0u;
    const uint32_t *c = q4 + (size_t)(z8 >> 2) * 256;                // (lane's chunk: 64 lanes x 4 dwords apart)
    return piece_nibbles(spread, c[0], c[1], c[2], c[3], z8 & 3);
}

// band_stage (filter_core.hpp) for a plane64 read: the bases dlo + 1 .. as dwords of eight codes, re-aligned to the
// band start.  q4: the lane's first chunk.
ATR_DEV void band_stage_planes(const uint32_t *q4, int nchunks, int dlo, uint32_t *ns, int nss, const uint32_t (*spread)[256],
                               int ndw = BAND_STREAM) {
    const int z0 = dlo >> 3;
    const uint32_t sh = 4u * (uint32_t)(dlo & 7);
    uint32_t raw[BAND_STREAM + 1];
#pragma unroll
    for (int k = 0; k <= BAND_STREAM; ++k) raw[k] = k <= ndw ? read_dword_planes(q4, nchunks, z0 + k, spread) : 0u;
#pragma unroll
    for (int k = 0; k < BAND_STREAM; ++k)
        if (k < ndw) ns[(size_t)k * nss] = sh ? ((raw[k] >> sh) | (raw[k + 1] << (32u - sh))) : raw[k];
}

}  // namespace atr
#endif
