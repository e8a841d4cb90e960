// pairs_core.hpp -- Aligner.locate for independent (reference, query) PAIRS: the aligner
// MergeOverlapping builds per read pair (commands/trim/modifiers.py:889-894:
// `Aligner(read2_rc, error_rate, flags).locate(read1.sequence)`, i.e. _align.pyx:266-491
// with a different reference for every call).
//
// One lane owns one pair.  The register-column kernel of locate_core.hpp needs a
// wave-uniform reference (its mismatch masks live in LDS/SGPRs), so here the DP column of a
// lane sits in LDS instead -- same packed cell word, same one-v_min3 tie-break, payload =
// matches -- strided by 64 so that lane l touches bank l, and the lane's reference is a
// second LDS array of 4-bit codes, eight rows per dword.  A column costs, per eight rows:
// one ds_read of reference codes, ~6 VALU ops to turn them into eight mismatch bits, and per
// row ds_read + 6 VALU + ds_write.  The sweep is the full (unbanded) matrix: Ukkonen's band is
// a CPU shortcut that does not change the result.
// Limits: int(e*m) < 256; m, n <= 255 (8-bit matches field) -- or <= 320 when STOP_WITHIN_SEQ2 is set (both
// flag sets of MergeOverlapping): the payload then counts the diagonal MISMATCHES of the path (XREP, as in
// locate_core.hpp), which stay below 256 on every cell that can still be accepted (cost <= k < 256; a cell
// beyond that may wrap its count, but its cost field is exact and only grows along a path, so it never wins
// against an acceptable cell and is never accepted itself).  The origin field is biased by 384 here (an
// alignment may start 320 rows into the reference), which leaves 0 .. 639 for starts inside the query.
//
// Compiled for gfx950 and, with -DATR_HOST_EMU, for the CPU test emulation.
#ifndef ATR_PAIRS_CORE_HPP
#define ATR_PAIRS_CORE_HPP

#include <cmath>
#include "locate_core.hpp"

namespace atr {

constexpr int PAIRS_MAX_LEN = ATR_PAIRS_MAX_LEN;
constexpr int PAIRS_MATCH_COUNT_MAX_LEN = 255;       // longest side the matches payload (no STOP_WITHIN_SEQ2) can count
constexpr int PAIRS_ORG_BIAS = 384;

struct PairParams {
    int16_t thr[PAIRS_MAX_LEN + 3];       // thr[L] = floor(L * max_error_rate), -1 = accept nothing
    double e;
    int flags, min_overlap, indel_cost, and_mode;
};

// 4-bit code j (0-based) of a tile64-packed read; q points at the lane's chunk 0 (uint32 view)
ATR_DEV uint32_t packed_code(const uint32_t *q, int j) {
    return (q[(size_t)(j >> 5) * 256 + ((j >> 3) & 3)] >> (4 * (j & 7))) & 15u;
}

ATR_DEV uint32_t bitrev4(uint32_t c) { return ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3); }

// Reference codes of one pair, 8 rows per dword, into refw[d * rs]; revcomp: the reference is
// the reverse complement of the packed sequence (complement of a DNA/IUPAC bit code = its
// bit reversal: util/__init__.py:67-88).
ATR_DEV void stage_reference(uint32_t *refw, int rs, const uint32_t *rp, int m, bool revcomp) {
    for (int d = 0; d * 8 < m; ++d) {
        uint32_t w = 0;
        for (int b = 0; b < 8; ++b) {
            const int i = d * 8 + b;
            if (i < m) {
                uint32_t c = packed_code(rp, revcomp ? m - 1 - i : i);
                if (revcomp) c = bitrev4(c);
                w |= c << (4 * b);
            }
        }
        refw[(size_t)d * rs] = w;
    }
}

// Aligner(ref, e, flags, ..., min_overlap, indel_cost).locate(query) for one pair.
// col: m + 1 cell words (stride cs); refw: staged reference (stride rs); qp: the lane's
// packed query.  rec: (refstart, refstop, querystart, querystop, matches, errors) as int16 x 8,
// refstop = -1 for None.
template <bool AND_MODE, bool XREP>
ATR_DEV void locate_pair_one(uint32_t *col, int cs, const uint32_t *refw, int rs, int m, const uint32_t *qp, int n,
                             const PairParams &p, const int16_t *thr, uint32_t rec[4]) {
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
    int k = (int)(p.e * m);                                            // _align.pyx:312
    if (k < 0) k = -1;
    int indel = p.indel_cost > k ? k + 1 : p.indel_cost;              // beyond k + 1 every indel is unaffordable alike
    if (indel < 1) indel = 1;
    const uint32_t insw = (uint32_t)indel * COST1 + PRIO_INS, delw = (uint32_t)indel * COST1 + PRIO_DEL;
    const uint32_t klimit = (uint32_t)(k + 1) << CSH;
    const int max_n = sq ? n : atr_min(n, m + k);                      // :314-321
    const int min_n = eq ? 0 : atr_max(0, n - m - k);
    for (int i = 0; i <= m; ++i)                                       // :333-352, with this file's origin bias
        col[(size_t)i * cs] = init_word(i, min_n, sr, sq, indel) + (uint32_t)(PAIRS_ORG_BIAS - (int)ORG_BIAS);
    Best best;
    best.key = COST_FIELD_MAX - (m + n);                               // (matches 0, cost m + n): :358-363
    best.word = (uint32_t)(m + n) << CSH;
    best.ref_stop = m; best.query_stop = n; best.matches = 0;
    uint32_t qword = 0;
    for (int j = min_n + 1; j <= max_n; ++j) {
        if (((j - 1) & 7) == 0 || j == min_n + 1) qword = qp[(size_t)((j - 1) >> 5) * 256 + (((j - 1) >> 3) & 3)];
        const uint32_t qrep = ((qword >> (4 * ((j - 1) & 7))) & 15u) * 0x11111111u;
        // row 0 (:385-388)
        const uint32_t row0 = sq ? ((uint32_t)PAIRS_ORG_BIAS + (uint32_t)j)
                                 : ((uint32_t)PAIRS_ORG_BIAS | ((uint32_t)atr_min(j * indel, INIT_COST_CAP) << CSH));
        uint32_t old_prev = col[0];                                    // old cell of the row above (diagonal source)
        uint32_t new_prev = row0;                                      // new cell of the row above (insertion source)
        col[0] = row0;
        // Full blocks of eight rows: the eight old cells are fetched up front (and the next
        // block's while this block's chain runs), so the serial min3 chain never waits for LDS.
        const int full = m >> 3;
        uint32_t oldc[8], nxt[8];
        if (full > 0) {
#pragma unroll
            for (int b = 0; b < 8; ++b) nxt[b] = col[(size_t)(b + 1) * cs];
        }
        for (int d = 0; d < full; ++d) {
#pragma unroll
            for (int b = 0; b < 8; ++b) oldc[b] = nxt[b];
            if (d + 1 < full) {
#pragma unroll
                for (int b = 0; b < 8; ++b) nxt[b] = col[(size_t)((d + 1) * 8 + b + 1) * cs];
            }
            const uint32_t rw = refw[(size_t)d * rs];
            const uint32_t v = AND_MODE ? (rw & qrep) : (rw ^ qrep);
            const uint32_t nz = (v | (v >> 1) | (v >> 2) | (v >> 3)) & 0x11111111u;     // nibble != 0
            const uint32_t mis = AND_MODE ? (nz ^ 0x11111111u) : nz;                    // :390-393
            uint32_t nw[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t bit = (mis >> (4 * b)) & 1u;
                const uint32_t cd = XREP ? old_prev + bit * (COST1 + MATCH1) : old_prev + MATCH1 + bit * DIAG_DELTA;               // :394-404
                const uint32_t cl = oldc[b] + delw, cu = new_prev + insw;               // :405-419
                nw[b] = atr_minu(atr_minu(cd, cl), cu) & ~PRIO_MASK;
                old_prev = oldc[b];
                new_prev = nw[b];
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) col[(size_t)(d * 8 + b + 1) * cs] = nw[b];
        }
        if (full * 8 < m) {                                             // the ragged last block, row by row
            const int d = full;
            const uint32_t rw = refw[(size_t)d * rs];
            const uint32_t v = AND_MODE ? (rw & qrep) : (rw ^ qrep);
            const uint32_t nz = (v | (v >> 1) | (v >> 2) | (v >> 3)) & 0x11111111u;
            const uint32_t mis = AND_MODE ? (nz ^ 0x11111111u) : nz;
            for (int b = 0; d * 8 + b < m; ++b) {
                const int i = d * 8 + b + 1;
                const uint32_t old = col[(size_t)i * cs];
                const uint32_t bit = (mis >> (4 * b)) & 1u;
                const uint32_t cd = XREP ? old_prev + bit * (COST1 + MATCH1) : old_prev + MATCH1 + bit * DIAG_DELTA;
                const uint32_t cl = old + delw, cu = new_prev + insw;
                const uint32_t nw = atr_minu(atr_minu(cd, cl), cu) & ~PRIO_MASK;
                col[(size_t)i * cs] = nw;
                old_prev = old;
                new_prev = nw;
            }
        }
        // row-m candidate: looked at only when the band reached row m, i.e. cost <= k (:433-455)
        if (eq && new_prev < klimit) consider<XREP, PAIRS_ORG_BIAS>(best, new_prev, m, j, p.min_overlap, thr, indel);
    }
    if (max_n == n) {                                                   // :461-474
        for (int i = er ? 0 : m; i <= m; ++i) consider<XREP, PAIRS_ORG_BIAS>(best, col[(size_t)i * cs], i, n, p.min_overlap, thr, indel);
    }
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != m + n) {                                                // :476-480
        const int origin = (int)(best.word & ORG_MASK) - PAIRS_ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

// ---- register-column variant (references of up to PAIRS_REG_MAX rows) -----------------------
// The same alignment with the DP column in VGPRs (locate_core.hpp's column layout, row i at
// position i) and the lane's reference turned into per-query-code MATCH masks in LDS
// (tab[(code * NW + w) * ts], NW = ceil(MT / 32) words; bit i-1 <=> row i matches): a column
// then costs NW ds_reads and six VALU ops per cell, with no LDS traffic per cell.  Row m sits at
// a per-lane position, so its cell is picked up by a select in the rows from the wave's
// smallest / largest m on (mlo / mhi, wave-uniform); rows beyond a lane's m compute don't-care cells.
constexpr int PAIRS_REG_MAX = 152;

template <int MT, bool AND_MODE>
ATR_DEV void build_match_masks(uint32_t *tab, int ts, const uint32_t *rp, int m, bool revcomp) {
    constexpr int NW = (MT + 31) / 32;
    for (int t = 0; t < 16 * NW; ++t) tab[(size_t)t * ts] = 0u;
    for (int i = 0; i < m; ++i) {
        uint32_t c = packed_code(rp, revcomp ? m - 1 - i : i);
        if (revcomp) c = bitrev4(c);
        const uint32_t bit = 1u << (i & 31);
        if (AND_MODE) {
            for (uint32_t q = 1; q < 16; ++q)
                if (q & c) tab[(size_t)(q * NW + (i >> 5)) * ts] |= bit;                // _align.pyx:392-393
        } else {
            tab[(size_t)(c * NW + (i >> 5)) * ts] |= bit;                               // :390-391 on 4-bit codes
        }
    }
}

// One column; returns the new cell of row m (position m, per lane).
// XREP (== STOP_WITHIN_SEQ2 set): the payload counts mismatches, as in locate_core.hpp -- one op
// less per cell.
template <int MT, bool XREP>
ATR_DEV uint32_t column_step_pairs(uint32_t (&col)[MT + 1], const uint32_t (&nm)[(MT + 31) / 32], uint32_t row0,
                                   uint32_t insw, uint32_t delw, int mlo, int mhi, int m) {
    uint32_t cd = diag_candidate<XREP>(col[0], nm, 0);
    col[0] = row0;
    uint32_t wm = row0;
#pragma unroll
    for (int i = 1; i <= MT; ++i) {
        uint32_t cd_next = 0;
        if (i < MT) cd_next = diag_candidate<XREP>(col[i], nm, i);
        const uint32_t cl = col[i] + delw;                   // deletion:  (i, j-1) -> (i, j)
        const uint32_t cu = col[i - 1] + insw;               // insertion: (i-1, j) -> (i, j)
        const uint32_t nw = atr_minu(atr_minu(cd, cl), cu) & ~PRIO_MASK;
        col[i] = nw;
        cd = cd_next;
        // the row-m cell is a per-lane pick; only the blocks of eight rows that hold some lane's m
        // (mlo .. mhi, wave-uniform) run the select chain -- one scalar test per block
        if ((i & 7) == 0 || i == MT) {
            const int b0 = ((i - 1) & ~7) + 1;                           // first row of this block
            if (b0 <= mhi && i >= mlo) {
#pragma unroll
                for (int r = b0; r <= i; ++r) wm = (r == m) ? col[r] : wm;
            }
        }
    }
    return wm;
}

template <int MT, bool AND_MODE, bool XREP>
ATR_DEV void locate_pair_reg(const uint32_t *tab, int ts, int m, int mlo, int mhi, const uint32_t *qp, int n, const PairParams &p,
                             const int16_t *thr, uint32_t rec[4]) {
    constexpr int NW = (MT + 31) / 32;
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
    int k = (int)(p.e * m);                                            // _align.pyx:312
    if (k < 0) k = -1;
    int indel = p.indel_cost > k ? k + 1 : p.indel_cost;
    if (indel < 1) indel = 1;
    const uint32_t insw = (uint32_t)indel * COST1 + PRIO_INS, delw = (uint32_t)indel * COST1 + PRIO_DEL;
    const uint32_t klimit = (uint32_t)(k + 1) << CSH;
    const int max_n = sq ? n : atr_min(n, m + k);                      // :314-321
    const int min_n = eq ? 0 : atr_max(0, n - m - k);
    uint32_t col[MT + 1];
#pragma unroll
    for (int i = 0; i <= MT; ++i) {
        // init_word(i, min_n, sr, sq, indel) (:333-352) as straight-line selects on the wave-uniform
        // flags -- 153 unrolled copies of its if-chain would be 153 branch diamonds (rows > m: don't care)
        const int d = min_n - i;
        const int cost = (sr ? (sq ? atr_min(i, min_n) : min_n) : (sq ? i : atr_max(i, min_n))) * indel;
        const int origin = sr ? (sq ? d : atr_min(0, d)) : (sq ? atr_max(0, d) : 0);
        col[i] = ((uint32_t)atr_min(cost, INIT_COST_CAP) << CSH) | (uint32_t)(origin + PAIRS_ORG_BIAS);
    }
    Best best;
    best.key = COST_FIELD_MAX - (m + n);
    best.word = (uint32_t)(m + n) << CSH;
    best.ref_stop = m; best.query_stop = n; best.matches = 0;
    uint32_t qword = 0;
    for (int j = min_n + 1; j <= max_n; ++j) {
        if (((j - 1) & 7) == 0 || j == min_n + 1) qword = qp[(size_t)((j - 1) >> 5) * 256 + (((j - 1) >> 3) & 3)];
        const uint32_t qc = (qword >> (4 * ((j - 1) & 7))) & 15u;
        uint32_t nm[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) nm[w] = ~tab[(size_t)(qc * NW + w) * ts];
        const uint32_t row0 = sq ? ((uint32_t)PAIRS_ORG_BIAS + (uint32_t)j)
                                 : ((uint32_t)PAIRS_ORG_BIAS | ((uint32_t)atr_min(j * indel, INIT_COST_CAP) << CSH));
        int mrow = m;
#ifndef ATR_HOST_EMU
        asm volatile("" : "+v"(mrow));                       // keep the 152 (i == m) lane masks out of the loop preheader
#endif
        const uint32_t wm = column_step_pairs<MT, XREP>(col, nm, row0, insw, delw, mlo, mhi, mrow);
        if (eq && wm < klimit) consider<XREP, PAIRS_ORG_BIAS>(best, wm, m, j, p.min_overlap, thr, indel);   // :433-455
    }
    if (max_n == n) {                                                   // :461-474
        const int first = er ? 0 : m;
#pragma unroll
        for (int i = 0; i <= MT; ++i)
            if (i >= first && i <= m) consider<XREP, PAIRS_ORG_BIAS>(best, col[i], i, n, p.min_overlap, thr, indel);
    }
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != m + n) {                                                // :476-480
        const int origin = (int)(best.word & ORG_MASK) - PAIRS_ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

// ---- register strips (references of more than PAIRS_REG_MAX rows) ----------------------------
// The matrix is swept in horizontal strips of PAIRS_STRIP_ROWS rows, each with the register column of the
// variant above: strip s holds rows R + 1 .. R + 128 (R = 128 s), its "row 0" is row R -- the free / costly
// top row for s = 0, else the bottom row the strip before left behind in `bnd` (one word per column, read at
// column j before this strip's own bottom cell of column j overwrites it).  A lane's row m is picked up in
// the strip that holds it; strips above a lane's m sweep don't-care rows, strips the whole wave is done with
// are skipped.  Candidate order is the reference's: the in-loop row-m cells of all columns come before the
// last-column cells (_align.pyx:433-474), so the two groups keep separate trackers and a last-column cell
// only wins when it is strictly better.
constexpr int PAIRS_STRIP_ROWS = 128;
constexpr int PAIRS_MAX_STRIPS = (ATR_PAIRS_MAX_LEN + PAIRS_STRIP_ROWS - 1) / PAIRS_STRIP_ROWS;

// match masks of rows r0 + 1 .. r0 + MT (bit i: row r0 + 1 + i)
template <int MT, bool AND_MODE>
ATR_DEV void build_match_masks_rows(uint32_t *tab, int ts, const uint32_t *rp, int m, bool revcomp, int r0) {
    constexpr int NW = (MT + 31) / 32;
    for (int t = 0; t < 16 * NW; ++t) tab[(size_t)t * ts] = 0u;
    for (int i = r0; i < m && i < r0 + MT; ++i) {
        uint32_t c = packed_code(rp, revcomp ? m - 1 - i : i);
        if (revcomp) c = bitrev4(c);
        const int b = i - r0;
        const uint32_t bit = 1u << (b & 31);
        if (AND_MODE) {
            for (uint32_t q = 1; q < 16; ++q)
                if (q & c) tab[(size_t)(q * NW + (b >> 5)) * ts] |= bit;
        } else {
            tab[(size_t)(c * NW + (b >> 5)) * ts] |= bit;
        }
    }
}

// mlo_s / mhi_s: per strip, the smallest / largest LOCAL row (m - R) over the wave's lanes whose m lies in that
// strip (mlo > mhi: none); mtop: the wave's largest m.  bnd: n + 1 words per lane, stride bs.
template <bool AND_MODE, bool XREP>
ATR_DEV void locate_pair_strips(uint32_t *tab, int ts, uint32_t *bnd, int bs, const uint32_t *rp, bool revcomp, int m,
                                const int *mlo_s, const int *mhi_s, int mtop, const uint32_t *qp, int n,
                                const PairParams &p, const int16_t *thr, uint32_t rec[4]) {
    constexpr int MT = PAIRS_STRIP_ROWS, NW = MT / 32;
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
    int k = (int)(p.e * m);
    if (k < 0) k = -1;
    int indel = p.indel_cost > k ? k + 1 : p.indel_cost;
    if (indel < 1) indel = 1;
    const uint32_t insw = (uint32_t)indel * COST1 + PRIO_INS, delw = (uint32_t)indel * COST1 + PRIO_DEL;
    const uint32_t klimit = (uint32_t)(k + 1) << CSH;
    const int max_n = sq ? n : atr_min(n, m + k);
    const int min_n = eq ? 0 : atr_max(0, n - m - k);
    Best loop_best, last_best;
    loop_best.key = last_best.key = COST_FIELD_MAX - (m + n);
    loop_best.word = last_best.word = (uint32_t)(m + n) << CSH;
    loop_best.ref_stop = last_best.ref_stop = m;
    loop_best.query_stop = last_best.query_stop = n;
    loop_best.matches = last_best.matches = 0;
    for (int s = 0; s * MT < mtop; ++s) {                                  // wave-uniform
        const int R = s * MT;
        const bool more = (s + 1) * MT < mtop;                             // another strip follows: leave the bottom row behind
        const int mloc = m - R;                                            // this lane's row m in strip rows (in 1 .. MT: here)
        const bool here = mloc >= 1 && mloc <= MT;
        build_match_masks_rows<MT, AND_MODE>(tab, ts, rp, m, revcomp, R);
        uint32_t col[MT + 1];
#pragma unroll
        for (int i = 0; i <= MT; ++i) {
            const int row = R + i, d = min_n - row;                        // init_word(row, min_n, sr, sq, indel), :333-352
            const int cost = (sr ? (sq ? atr_min(row, min_n) : min_n) : (sq ? row : atr_max(row, min_n))) * indel;
            const int origin = sr ? (sq ? d : atr_min(0, d)) : (sq ? atr_max(0, d) : 0);
            col[i] = ((uint32_t)atr_min(cost, INIT_COST_CAP) << CSH) | (uint32_t)(origin + PAIRS_ORG_BIAS);
        }
        if (more) bnd[(size_t)min_n * bs] = col[MT];
        uint32_t qword = 0;
        for (int j = min_n + 1; j <= max_n; ++j) {
            if (((j - 1) & 7) == 0 || j == min_n + 1) qword = qp[(size_t)((j - 1) >> 5) * 256 + (((j - 1) >> 3) & 3)];
            const uint32_t qc = (qword >> (4 * ((j - 1) & 7))) & 15u;
            uint32_t nm[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) nm[w] = ~tab[(size_t)(qc * NW + w) * ts];
            uint32_t row0;
            if (s == 0) row0 = sq ? ((uint32_t)PAIRS_ORG_BIAS + (uint32_t)j)
                                  : ((uint32_t)PAIRS_ORG_BIAS | ((uint32_t)atr_min(j * indel, INIT_COST_CAP) << CSH));
            else row0 = bnd[(size_t)j * bs];                               // row R of this column, from the strip above
            int mrow = here ? mloc : 0;
#ifndef ATR_HOST_EMU
            asm volatile("" : "+v"(mrow));
#endif
            const uint32_t wm = column_step_pairs<MT, XREP>(col, nm, row0, insw, delw, mlo_s[s], mhi_s[s], mrow);
            if (more) bnd[(size_t)j * bs] = col[MT];
            if (eq && here && wm < klimit) consider<XREP, PAIRS_ORG_BIAS>(loop_best, wm, m, j, p.min_overlap, thr, indel);
        }
        if (max_n == n) {                                                   // :461-474, rows of this strip in increasing order
            const int first = er ? 0 : m;
#pragma unroll
            for (int i = 0; i <= MT; ++i) {
                const int row = R + i;
                if ((i > 0 || s == 0) && row >= first && row <= m)
                    consider<XREP, PAIRS_ORG_BIAS>(last_best, col[i], row, n, p.min_overlap, thr, indel);
            }
        }
    }
    const Best &best = last_best.key > loop_best.key ? last_best : loop_best;
    const int cost = (int)(best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != m + n) {
        const int origin = (int)(best.word & ORG_MASK) - PAIRS_ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

// Host side: thresholds and envelope check shared by the library and the emulation.
inline int pairs_params(double e, int flags, int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                        int ref_max_len, int query_max_len, PairParams &p) {
    if (flags < 0 || flags > 15 || min_overlap < 1 || indel_cost < 1) return ATR_ERR_INVALID;
    if (ref_max_len < 0 || query_max_len < 0) return ATR_ERR_INVALID;
    if (ref_max_len > PAIRS_MAX_LEN || query_max_len > PAIRS_MAX_LEN) return ATR_ERR_UNSUPPORTED;
    if ((ref_max_len > PAIRS_MATCH_COUNT_MAX_LEN || query_max_len > PAIRS_MATCH_COUNT_MAX_LEN) && !(flags & ATR_STOP_WITHIN_SEQ2))
        return ATR_ERR_UNSUPPORTED;
    const double kd = e * (double)ref_max_len;
    if (!(kd < 256.0) || !(kd > -1.0e9)) return ATR_ERR_UNSUPPORTED;
    // The cost field never overflows: initial cells and row 0 are saturated at INIT_COST_CAP (always > k, so a
    // saturated cell and everything derived from it stays out of every candidate test), a cell is at most its
    // diagonal neighbour + 1, and a candidate word formed on the way adds one indel (capped at k + 1 <= 256 inside
    // the kernels): INIT_COST_CAP + min(m, n) + 257 < 4096 for every supported length.
    static_assert(INIT_COST_CAP + PAIRS_MAX_LEN + 257 < COST_FIELD_MAX, "cost field of the pair kernels");
    for (int L = 0; L < PAIRS_MAX_LEN + 3; ++L) {
        double t = std::floor((double)L * e);                           // cost <= L*e  <=>  cost <= floor(L*e)
        if (t > COST_FIELD_MAX) t = COST_FIELD_MAX;
        p.thr[L] = t < 0 ? (int16_t)-1 : (int16_t)t;
    }
    p.e = e; p.flags = flags; p.min_overlap = min_overlap; p.indel_cost = indel_cost;
    p.and_mode = (wildcard_ref || wildcard_query) ? 1 : 0;
    return ATR_OK;
}

}  // namespace atr
#endif
