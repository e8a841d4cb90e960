// wave_core.hpp -- Aligner.locate for SHORT batches: one read per WAVEFRONT, one DP row per lane.
// (reference: atropos/align/_align.pyx:266-491; the cell word, the candidate test and the
// initial column are those of locate_core.hpp.)
//
// The kernels of locate_kernel.hpp / filter_core.hpp give every read one lane: a column of m rows is
// a chain of 3 m dependent VALU ops, so a call over a few hundred reads -- the 1000-read batches of the
// unchanged trim command (commands/base.py:179), the per-read API -- runs at the latency of ONE lane
// (a 150-base read against a 33-row adapter: ~40 k dependent ops) while 63 lanes of the wave and most of
// the chip idle.  Here lane l owns ROW l (lane 0: row 0) and the wave sweeps ANTI-DIAGONALS: in step t
// lane l computes cell (l, j) with j = min_n + t - l, from
//     left  (l,     j - 1)  its own cell of step t - 1,
//     up    (l - 1, j)      lane l - 1's cell of step t - 1  (one DPP wave_shr:1),
//     diag  (l - 1, j - 1)  the `up` of step t - 1,
// i.e. n + m steps of 7 - 12 VALU ops instead of n * m * 7 (references of 64 bases and more: two or three rows per
// lane, see WaveRows below).  Lane 0 has no lane below it: the DPP move leaves
// its destination alone there, and that register holds WAVE_HUGE from the start, so lane 0 always takes the
// `left` candidate, with its own increment instead of the deletion cost: row 0 of the reference
// (:385-388: origin j, or cost j * indel).  A lane is active while 1 <= j - min_n <= span and keeps its
// last cell afterwards, so at the end lane l holds (l, max_n): the last column.  Row-m candidates are taken
// by lane m in column order, the last column by a wave reduction that prefers the smallest row among equal
// keys (the reference scans it in ascending rows and keeps the first on ties).  A single wave issues one
// instruction every few cycles whatever its kind, so the sweep is split: the steps in which every row is
// active (m < t <= span) run without the per-lane activity test, eight per trip (four with more than one row per
// lane), with one look at lane m's cells per trip.  The query codes sit in LDS, one dword per column (a byte would
// cost a zero-extension per use); every lane fetches the codes of its next columns half a trip ahead.
#ifndef ATR_WAVE_CORE_HPP
#define ATR_WAVE_CORE_HPP

#include "locate_core.hpp"

namespace atr {

constexpr int WAVE_MAX_M = ATR_MAX_REF_LEN;    // rows 0 .. m on the 64 lanes, up to three per lane
constexpr long long WAVE_MAX_READS = 32768;    // beyond: the lane-per-read kernels (the chip is full, fewer instructions win)
constexpr int WAVE_CODE_PAD = 64;              // LDS entries in front of the codes, twice as many behind (inactive lanes read there)
constexpr uint32_t WAVE_HUGE = 0xBFF00000u;    // "no such cell": cost 3071, beyond every real cell, no overflow after + indel

#ifdef ATR_HOST_EMU
static inline uint32_t atr_bfe1v(uint32_t w, uint32_t i) { return (w >> (i & 31u)) & 1u; }
#else
static __device__ __forceinline__ uint32_t atr_bfe1v(uint32_t w, uint32_t i) {
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "v"(i));
    return r;
}
#endif

// bit c of the result: reference row `row` (1-based) does NOT match query code c
ATR_DEV uint32_t wave_rowmask(const LocateParams &p, int p0, int row) {
    const int pos = p0 + row - 1, w = pos >> 5, b = pos & 31;
    // (masks, not a select chain over the four words: the compiler turns that into p.nmask[c][w], a run-time index
    //  that moves the whole parameter block to scratch memory)
    const uint32_t s0 = w == 0 ? ~0u : 0u, s1 = w == 1 ? ~0u : 0u, s2 = w == 2 ? ~0u : 0u, s3 = w == 3 ? ~0u : 0u;
    uint32_t mask = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const uint32_t word = (p.nmask[c][0] & s0) | (p.nmask[c][1] & s1) | (p.nmask[c][2] & s2) | (p.nmask[c][3] & s3);
        mask |= ((word >> b) & 1u) << c;
    }
    return mask;
}

// row 0 of column j (_align.pyx:385-388)
ATR_DEV uint32_t wave_row0(const Uniform &u, int j) {
    return u.sq ? (ORG_BIAS + (uint32_t)j) : (ORG_BIAS | ((uint32_t)atr_min(j * u.indel, INIT_COST_CAP) << CSH));
}
// what a step adds to the `left` candidate of lane l: the deletion word, or row 0's increment per column
ATR_DEV uint32_t wave_left_step(const Uniform &u, int lane) {
    return lane > 0 ? u.delw : u.sq ? 1u : (uint32_t)u.indel << CSH;
}
constexpr uint32_t WAVE_ROW0_CAP = ORG_BIAS | ((uint32_t)INIT_COST_CAP << CSH);     // row 0 without START_WITHIN_SEQ2, saturated

// one cell: the three-way choice of locate_core.hpp's column_step.  SQ == false: row 0's cost is capped as in
// wave_row0 (as a minimum over the whole wave: a cell of cost >= 2047 never decides anything).
template <bool XREP, bool SQ>
ATR_DEV uint32_t wave_cell(uint32_t diag, uint32_t left, uint32_t up, uint32_t rowmask, uint32_t q, uint32_t insw,
                           uint32_t left_step) {
    const uint32_t bit = atr_bfe1v(rowmask, q);
    const uint32_t cd = XREP ? atr_mad24(bit, COST1 + MATCH1, diag) : atr_mad24(bit, DIAG_DELTA, diag + MATCH1);
    uint32_t nw = atr_minu(atr_minu(cd, left + left_step), up + insw) & ~PRIO_MASK;
    if (!SQ) nw = atr_minu(nw, WAVE_ROW0_CAP);
    return nw;
}

// the column window of a read of n bases (_align.pyx:314-321)
struct WaveWindow { int min_n, max_n, span; bool scan; };
template <bool XREP>
ATR_DEV WaveWindow wave_window(const Uniform &u, int n) {
    WaveWindow w;
    w.max_n = u.sq ? n : atr_min(n, u.m + u.k);
    w.min_n = XREP ? 0 : atr_max(0, n - u.m - u.k);
    w.span = atr_max(0, w.max_n - w.min_n);
    w.scan = w.max_n == n && (w.span > 0 || n == 0);          // :461; an empty read scans its initial column
    return w;
}

ATR_DEV void wave_best_init(Best &b, const Uniform &u, int n) {
    b.key = COST_FIELD_MAX - (u.m + n);                      // (matches 0, cost m + n): :358-363
    b.word = (uint32_t)(u.m + n) << CSH;
    b.ref_stop = u.m; b.query_stop = n; b.matches = 0;
}

// the result record of locate_core.hpp's lane_result from a Best
ATR_DEV void wave_result(const Best &best, const Uniform &u, int n, uint32_t rec[4]) {
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

inline bool wave_applies(int m, long long nreads) { return m >= 1 && m <= WAVE_MAX_M && nreads <= WAVE_MAX_READS; }

// ---- R rows per lane (the per-pair aligner: references of up to 64 R - 1 bases) -------------------------------
// Lane l owns R consecutive rows and sweeps them top-down inside a step (the column_step of locate_core.hpp on R
// cells); its top row takes `up` / `diag` from lane l - 1's BOTTOM row through the DPP move.  Rows are
// BOTTOM-ALIGNED: row m is the bottom row of lane L - 1 (L = lanes in use), so its cell is found without a
// per-pair select; the `off` = L R - 1 - m slots above row 0 in lane 0 are padding that stays at WAVE_HUGE
// (left step 0), which makes row 0 -- wherever it sits in lane 0 -- take its `left` candidate as in the
// one-row kernel.
constexpr int WAVE_PAIR_ROWS_MAX = 5;          // 64 * 5 - 1 = 319 rows
constexpr long long WAVE_MAX_PAIRS = 32768;
inline int wave_pair_rows(int m) { return (m + 1 + 63) / 64; }                 // rows per lane for a reference of m bases
inline bool wave_pairs_applies(int ref_max_len, long long npairs) {
    return ref_max_len >= 0 && ref_max_len <= 64 * WAVE_PAIR_ROWS_MAX - 1 && npairs <= WAVE_MAX_PAIRS;
}

template <int R>
struct WaveRows {
    uint32_t col[R];           // the lane's cells of its current column
    uint32_t rowmask[R];       // bit c: the row does NOT match query code c
    uint32_t lstep[R];         // what a step adds to the `left` candidate: deletion word / row 0's increment / 0 (padding)
};

// the new cells of a lane for the next column (not stored); CAPW: see wave_cell
template <bool XREP, bool SQ, int R, uint32_t CAPW>
ATR_DEV void wave_rows_step(const WaveRows<R> &W, uint32_t diag_in, uint32_t up_in, uint32_t q, uint32_t insw, uint32_t (&out)[R]) {
    uint32_t prev_old = diag_in, prev_new = up_in;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t bit = atr_bfe1v(W.rowmask[r], q);
        const uint32_t cd = XREP ? atr_mad24(bit, COST1 + MATCH1, prev_old) : atr_mad24(bit, DIAG_DELTA, prev_old + MATCH1);
        uint32_t nw = atr_minu(atr_minu(cd, W.col[r] + W.lstep[r]), prev_new + insw) & ~PRIO_MASK;
        if (!SQ) nw = atr_minu(nw, CAPW);
        prev_old = W.col[r];
        prev_new = nw;
        out[r] = nw;
    }
}

// geometry of one pair on the wave: rows per lane R, lanes in use, padding slots above row 0
struct WaveGeom { int lanes, off; };
ATR_DEV WaveGeom wave_geom(int m, int R) {
    WaveGeom g;
    g.lanes = (m + 1 + R - 1) / R;
    g.off = g.lanes * R - 1 - m;
    return g;
}
// global row of slot rr of `lane` (negative: padding)
ATR_DEV int wave_slot_row(const WaveGeom &g, int R, int lane, int rr) { return lane * R + rr - g.off; }

}  // namespace atr
#endif
