// locate_core.hpp -- per-lane arithmetic of the batched Aligner.locate kernel
// (reference: atropos/align/_align.pyx:266-491).
//
// Everything here is written once and compiled twice:
//   * by hipcc for gfx950 (ATR_DEV = __device__ __forceinline__), where one lane of
//     a wavefront owns one read and `LaneState` lives entirely in VGPRs;
//   * by g++ with -DATR_HOST_EMU for tests/emu (a lock-step 64-lane emulation of the
//     kernel's control flow that lets the CPU test-suite check the kernel logic
//     against the oracle without a GPU).  It is test infrastructure only: the
//     product library (libatropos_hip.so) contains the GPU build and nothing else.
//
// Cell word (one 32-bit VGPR per DP cell):
//     [31:20] cost | [19:18] tie-break priority | [17:10] matches OR mismatches | [9:0] origin + 256
// With STOP_WITHIN_SEQ2 (every adapter type but the anchored 3' one) the payload counts the
// diagonal MISMATCHES x of the chosen path instead of its matches: a matching diagonal step
// then leaves the word unchanged and a mismatching one adds COST1 + X1, i.e. the diagonal
// candidate is one v_bfe_u32 (the row's bit of the column's mismatch mask) and one
// v_mad_u32_u24.  The matches are recovered only when a cell is actually considered as a
// candidate: rows R and columns C consumed since the path's start, cost = x + c*(a + b),
// R - C = a - b  =>  insertions a, diagonal steps R - a, matches R - a - x.
// The reference's three-way choice with its tie order (mismatch <= insertion <=
// deletion, _align.pyx:405-419) becomes ONE v_min3_u32: the three candidates carry
// priority 0/1/2 just below the cost field, the payload (matches, origin) rides in
// the low bits and never decides a comparison because the priorities differ.
#ifndef ATR_LOCATE_CORE_HPP
#define ATR_LOCATE_CORE_HPP

#include <stdint.h>
#include "atropos_hip.h"

#ifdef ATR_HOST_EMU
#define ATR_DEV static inline
#ifndef ATR_DEV_MEMBER
#define ATR_DEV_MEMBER inline
#endif
static inline int atr_min(int a, int b) { return a < b ? a : b; }
static inline int atr_max(int a, int b) { return a > b ? a : b; }
static inline uint32_t atr_minu(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t atr_mad24(uint32_t a, uint32_t b, uint32_t c) { return a * b + c; }
static inline uint32_t atr_bfe1(uint32_t w, int i) { return (w >> i) & 1u; }
static inline int atr_clz(uint32_t w) { return __builtin_clz(w); }     // w != 0
static inline int atr_ctz(uint32_t w) { return __builtin_ctz(w); }     // w != 0
static inline int atr_popc64(uint64_t w) { return __builtin_popcountll(w); }
#else
#define ATR_DEV __device__ __forceinline__
#ifndef ATR_DEV_MEMBER
#define ATR_DEV_MEMBER __device__ __forceinline__
#endif
#define atr_min min
#define atr_max max
#define atr_minu min
static __device__ __forceinline__ int atr_clz(uint32_t w) { return __clz((int)w); }
static __device__ __forceinline__ int atr_ctz(uint32_t w) { return __ffs((int)w) - 1; }
static __device__ __forceinline__ int atr_popc64(uint64_t w) { return __popcll(w); }
// bit i of w as 0/1, and a*b+c with a 24-bit product: written as inline asm because hipcc
// otherwise rewrites the 0/1 multiply into and + cmp + cndmask + add
static __device__ __forceinline__ uint32_t atr_bfe1(uint32_t w, int i) {
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "i"(i));
    return r;
}
static __device__ __forceinline__ uint32_t atr_mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
#endif

namespace atr {

constexpr int MSH = 10, PSH = 18, CSH = 20;
constexpr uint32_t ORG_BIAS = 256;
constexpr uint32_t ORG_MASK = 0x3FFu, MAT_MASK = 0xFFu;
constexpr uint32_t COST1 = 1u << CSH, MATCH1 = 1u << MSH;
constexpr uint32_t DIAG_DELTA = COST1 - MATCH1;      // mismatch instead of match on the diagonal (fits 24 bits)
constexpr uint32_t PRIO_INS = 1u << PSH, PRIO_DEL = 2u << PSH, PRIO_MASK = 3u << PSH;
constexpr int COST_FIELD_MAX = 4095;
constexpr int INIT_COST_CAP = 2047;       // saturated initial-column cost (always > k)
constexpr int ROW_GRAN = 4;               // register-column sizes come in multiples of this

// Wave-uniform description of one aligner; passed by value as a kernel argument so
// that the adapter codes sit in SGPRs.
struct LocateParams {
    // nmask[c] : 128-bit mask over column positions, bit p set <=> the reference row sitting at
    // position p+1 does NOT match query code c (byte equality or 4-bit AND, decided by the
    // host).  Staged in LDS; a lane fetches the mask of its query code once per column and
    // every row tests its own bit, so no per-row scalar state is needed.
    uint32_t nmask[16][4];
    int16_t thr[ATR_MAX_REF_LEN + 2];     // thr[L] = floor(L * max_error_rate), clamped; -1 = accept nothing
    int m, flags, k, min_overlap, indel;  // indel = effective cost, min(indel_cost, k+1)
};

struct Best {
    int key;            // (matches << 12) | (4095 - cost): "more matches, then fewer errors"
    uint32_t word;      // winning cell
    int ref_stop, query_stop, matches;
};

// Candidate test of _align.pyx:440-455 / :464-474.  XREP: the payload field holds the
// diagonal mismatches (see the header); indel = the aligner's effective indel cost, 0 for
// the no-indel kernels.
// obase: the column the origin field counts from (long reads: the rolling base of locate_long, else 0).
template <bool XREP, int BIAS = (int)ORG_BIAS>
ATR_DEV void consider(Best &b, uint32_t w, int ref_stop, int query_stop, int min_overlap,
                      const int16_t *thr, int indel, int obase = 0) {
    const int cost = (int)(w >> CSH);
    const int origin = (int)(w & ORG_MASK) - BIAS + obase;
    const int length = ref_stop + atr_min(origin, 0);
    if (length >= min_overlap && cost <= (int)thr[atr_max(length, 0)]) {
        int matches = (int)((w >> MSH) & MAT_MASK);
        if (XREP) {
            const int x = matches;
            const int cols = query_stop - atr_max(origin, 0);
            const int a = indel > 0 ? (((cost - x) / indel + length - cols) >> 1) : 0;
            matches = length - a - x;
        }
        const int key = (matches << 12) | (COST_FIELD_MAX - cost);
        if (key > b.key) {                                    // strict: first seen wins ties
            b.key = key; b.word = w; b.ref_stop = ref_stop; b.query_stop = query_stop; b.matches = matches;
        }
    }
}

// Initial column (_align.pyx:333-352) for row i at column min_n; rows < 0 are padding.
ATR_DEV uint32_t init_word(int i, int min_n, bool sr, bool sq, int indel) {
    int cost, origin;
    if (!sr && !sq)      { cost = atr_max(i, min_n) * indel; origin = 0; }
    else if (sr && !sq)  { cost = min_n * indel;             origin = atr_min(0, min_n - i); }
    else if (!sr && sq)  { cost = i * indel;                 origin = atr_max(0, min_n - i); }
    else                 { cost = atr_min(i, min_n) * indel; origin = min_n - i; }
    if (i < 0) { cost = INIT_COST_CAP; origin = 0; }
    return ((uint32_t)atr_min(cost, INIT_COST_CAP) << CSH) | (uint32_t)(origin + (int)ORG_BIAS);
}

// One DP column over MT register-resident positions; nm = this lane's mismatch mask for the
// column's query code (LocateParams::nmask).  The adapter occupies the LAST m
// positions (row i sits at position p0 + i, p0 = MT - m in 0..ROW_GRAN-1), so row m is
// always col[MT] and the code is straight-line: positions below p0 compute don't-care
// cells that nothing above depends on, and position p0 is overwritten with the row-0
// word.  The diagonal candidate of row i+1 is formed from the OLD col[i] before row i
// overwrites it, so every cell is updated in place (no register rotation).
// Equal characters take the diagonal unconditionally (_align.pyx:394-398).  Because
// neighbouring cells differ by at most one indel, diag.cost <= left/up cost + indel,
// and the priority bits break that tie for the diagonal, so one v_min3 serves both
// the "equal" and the "three-way choice" case.  Returns the new cell of row m.
// Diagonal candidate of the row at position i+1: old cell i plus MATCH1, or plus COST1 when
// bit i of the mismatch mask is set (v_bfe_u32 + v_mad_u32_u24).
template <bool XREP, int W>
ATR_DEV uint32_t diag_candidate(uint32_t cell, const uint32_t (&nm)[W], int i) {
    const uint32_t bit = atr_bfe1(nm[i >> 5], i & 31);
    return XREP ? atr_mad24(bit, COST1 + MATCH1, cell) : atr_mad24(bit, DIAG_DELTA, cell + MATCH1);
}

template <int MT, bool NOINDEL, bool XREP>
ATR_DEV uint32_t column_step(uint32_t (&col)[MT + 1], const uint32_t (&nm)[(MT + 31) / 32], int p0, uint32_t row0,
                             uint32_t insw, uint32_t delw) {
    uint32_t cd = diag_candidate<XREP>(col[0], nm, 0);
    col[0] = row0;
#pragma unroll
    for (int i = 1; i <= MT; ++i) {
        uint32_t cd_next = 0;
        if (i < MT) cd_next = diag_candidate<XREP>(col[i], nm, i);
        uint32_t nw;
        if (NOINDEL) {
            nw = cd;
        } else {
            const uint32_t cl = col[i] + delw;               // deletion:  (i, j-1) -> (i, j)
            const uint32_t cu = col[i - 1] + insw;           // insertion: (i-1, j) -> (i, j)
            nw = atr_minu(atr_minu(cd, cl), cu);
            // position p0 holds the row-0 word (only the first positions can be p0): the select is folded
            // into the priority-clearing AND as one v_and_or_b32 with wave-uniform operands
            if (i < ROW_GRAN) nw = (nw & (i == p0 ? 0u : ~PRIO_MASK)) | (i == p0 ? row0 : 0u);
            else nw &= ~PRIO_MASK;
        }
        if (NOINDEL && i < ROW_GRAN) nw = (i == p0) ? row0 : nw;
        col[i] = nw;
        cd = cd_next;
    }
    return col[MT];
}

// Row-limited variant for the window DP: rows above `plimit` (a wave-uniform POSITION) are
// not needed -- rows only depend on the rows before them -- so the sweep stops at the first
// block of four positions that starts beyond it.  Nested ifs (template recursion), so that
// nothing is live across a skipped block and every cell is still updated in place.
template <int MT, bool NOINDEL, bool XREP, int B>
ATR_DEV void row_blocks(uint32_t (&col)[MT + 1], const uint32_t (&nm)[(MT + 31) / 32], int p0, uint32_t row0,
                        uint32_t insw, uint32_t delw, uint32_t cd, int plimit) {
    if constexpr (B < MT / 4) {
        if (4 * B + 1 > plimit) return;                          // wave-uniform
#pragma unroll
        for (int i = 4 * B + 1; i <= 4 * B + 4; ++i) {
            uint32_t cd_next = 0;
            if (i < MT) cd_next = diag_candidate<XREP>(col[i], nm, i);
            uint32_t nw;
            if (NOINDEL) {
                nw = cd;
            } else {
                const uint32_t cl = col[i] + delw;
                const uint32_t cu = col[i - 1] + insw;
                nw = atr_minu(atr_minu(cd, cl), cu);
                if (i < ROW_GRAN) nw = (nw & (i == p0 ? 0u : ~PRIO_MASK)) | (i == p0 ? row0 : 0u);
                else nw &= ~PRIO_MASK;
            }
            if (NOINDEL && i < ROW_GRAN) nw = (i == p0) ? row0 : nw;
            col[i] = nw;
            cd = cd_next;
        }
        row_blocks<MT, NOINDEL, XREP, B + 1>(col, nm, p0, row0, insw, delw, cd, plimit);
    }
}

template <int MT, bool NOINDEL, bool XREP>
ATR_DEV uint32_t column_step_limited(uint32_t (&col)[MT + 1], const uint32_t (&nm)[(MT + 31) / 32], int p0,
                                     uint32_t row0, uint32_t insw, uint32_t delw, int plimit) {
    const uint32_t cd = diag_candidate<XREP>(col[0], nm, 0);
    col[0] = row0;
    row_blocks<MT, NOINDEL, XREP, 0>(col, nm, p0, row0, insw, delw, cd, plimit);
    return col[MT];
}

// Last-column candidates (_align.pyx:461-474): every row from first_i on, increasing.
// last_p (wave-uniform): the highest position the sweep kept up to date (window DP).
// klimit = (k + 1) << CSH: no threshold exceeds thr[m] = k, so a cell of cost > k is skipped with one
// compare instead of the full test (threshold fetch, length, ...).
template <int MT, bool XREP>
ATR_DEV void scan_last_column(Best &best, const uint32_t (&col)[MT + 1], int p0, int first_p, int n,
                              int min_overlap, const int16_t *thr, int indel, uint32_t klimit, int last_p = MT, int obase = 0) {
#pragma unroll
    for (int i = 0; i <= MT; ++i) {
        if (i >= first_p && i <= last_p && col[i] < klimit) consider<XREP>(best, col[i], i - p0, n, min_overlap, thr, indel, obase);
    }
}

// Everything one lane carries through the sweep.
template <int MT>
struct LaneState {
    uint32_t col[MT + 1];
    Best best;
    int n, min_n, max_n;
    bool scan;
};

// Wave-uniform constants derived from LocateParams once per kernel.
struct Uniform {
    int m, k, p0, first_p, indel, min_overlap;
    bool sr, sq, er, eq;
    uint32_t insw, delw, klimit;
};

ATR_DEV int round_up_rows_dev(int m) { return (m + ROW_GRAN - 1) / ROW_GRAN * ROW_GRAN; }

ATR_DEV Uniform make_uniform(const LocateParams &p, int MT) {
    Uniform u;
    u.m = p.m; u.k = p.k; u.p0 = MT - p.m; u.indel = p.indel; u.min_overlap = p.min_overlap;
    u.sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0; u.sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    u.er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0;  u.eq = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
    u.first_p = u.p0 + (u.er ? 0 : u.m);
    u.insw = (uint32_t)u.indel * COST1 + PRIO_INS;
    u.delw = (uint32_t)u.indel * COST1 + PRIO_DEL;
    u.klimit = (uint32_t)(u.k + 1) << CSH;                   // cost <= k  <=>  word < klimit
    return u;
}

// Window, initial column and "no match yet" state of one read of length n.
// s_init holds the min_n == 0 initial column (by position).
// XREP (== STOP_WITHIN_SEQ2 set): min_n is 0 and the payload counts mismatches.
template <int MT, bool NOINDEL, bool XREP>
ATR_DEV void lane_init(LaneState<MT> &L, const Uniform &u, int n, const uint32_t *s_init, const int16_t *thr) {
    L.n = n;
    L.max_n = u.sq ? n : atr_min(n, u.m + u.k);              // _align.pyx:314-321
    L.min_n = XREP ? 0 : atr_max(0, n - u.m - u.k);
    L.scan = (L.max_n == n);                                 // :461
    if (XREP) {
#pragma unroll
        for (int i = 0; i <= MT; ++i) L.col[i] = s_init[i];
    } else {
#pragma unroll
        for (int i = 0; i <= MT; ++i) L.col[i] = init_word(i - u.p0, L.min_n, u.sr, u.sq, u.indel);
    }
    L.best.key = COST_FIELD_MAX - (u.m + n);                 // (matches 0, cost m+n): :358-363
    L.best.word = (uint32_t)(u.m + n) << CSH;
    L.best.ref_stop = u.m; L.best.query_stop = n; L.best.matches = 0;
    // Empty reads never enter the column loop: their "last column" is the initial one.
    if (L.scan && n == 0)
        scan_last_column<MT, XREP>(L.best, L.col, u.p0, u.first_p, n, u.min_overlap, thr, NOINDEL ? 0 : u.indel, u.klimit);
}

// Window mode (filter_core.hpp): the DP is started afresh at column j_lo as if row i had
// been reached from (0, j_lo) by i insertions; column 0 keeps the aligner's own initial
// column.  Only used with START_WITHIN_SEQ2 and STOP_WITHIN_SEQ2.
ATR_DEV uint32_t window_init_word(int i, int j_lo, int indel) {
    const int cost = i < 0 ? INIT_COST_CAP : atr_min(i * indel, INIT_COST_CAP);
    return ((uint32_t)cost << CSH) | (uint32_t)(j_lo + (int)ORG_BIAS);
}

// All lanes of a wave start at the SAME column j_start (the smallest window start in the
// wave): starting earlier than a read's own window start is just as exact, and it makes
// the initial column wave-uniform.
template <int MT, bool NOINDEL>
ATR_DEV void lane_init_window(LaneState<MT> &L, const Uniform &u, int n, int j_start, int j_hi, bool scan,
                              const uint32_t *s_init, const int16_t *thr) {
    L.n = n; L.min_n = j_start; L.max_n = j_hi; L.scan = scan;
    if (j_start == 0) {
#pragma unroll
        for (int i = 0; i <= MT; ++i) L.col[i] = s_init[i];
    } else {
#pragma unroll
        for (int i = 0; i <= MT; ++i) L.col[i] = window_init_word(i - u.p0, j_start, u.indel);
    }
    L.best.key = COST_FIELD_MAX - (u.m + n);
    L.best.word = (uint32_t)(u.m + n) << CSH;
    L.best.ref_stop = u.m; L.best.query_stop = n; L.best.matches = 0;
    // an empty window only arises for an empty read whose initial column already qualifies
    if (scan && j_hi <= j_start)
        scan_last_column<MT, true>(L.best, L.col, u.p0, u.first_p, n, u.min_overlap, thr, NOINDEL ? 0 : u.indel, u.klimit);
}

// Linked adapters (linked_core.hpp): a lane whose own alignment starts at column j_start, later
// than the wave's common start column, starts afresh there.
template <int MT>
ATR_DEV void lane_restart_window(LaneState<MT> &L, const Uniform &u, int j_start) {
#pragma unroll
    for (int i = 0; i <= MT; ++i) L.col[i] = window_init_word(i - u.p0, j_start, u.indel);
    L.best.key = COST_FIELD_MAX - (u.m + L.n);
    L.best.word = (uint32_t)(u.m + L.n) << CSH;
    L.best.ref_stop = u.m; L.best.query_stop = L.n; L.best.matches = 0;
}

// Column j of the wave-uniform sweep, query code q.  Every lane of the wave executes
// the column update unmasked; a lane whose own window (min_n, max_n] is narrower than
// the wave's range
//   * is re-initialised when the sweep reaches its min_n (only without STOP_WITHIN_SEQ2),
//   * takes its last-column candidates when the sweep reaches its max_n,
//   * ignores row-m candidates outside its window,
// and whatever it computes past max_n is never looked at.
// obase (long reads, locate_long_kernel): the origin fields count from column obase instead of 0.
template <int MT, bool NOINDEL, bool XREP, bool WIN = false>
ATR_DEV void lane_step(LaneState<MT> &L, const Uniform &u, int j, const uint32_t (&nm)[(MT + 31) / 32],
                       const int16_t *thr, int plimit = MT, int obase = 0) {
    const int indel = NOINDEL ? 0 : u.indel;
    // row 0 (:385-388): origin j, or cost j*indel (saturated: it is > k long before).
    // Matches 0; without START_WITHIN_SEQ2 the row-0 origin is 0 in every init case.
    const uint32_t row0 = u.sq ? (ORG_BIAS + (uint32_t)(j - obase))
                               : (ORG_BIAS | ((uint32_t)atr_min(j * u.indel, INIT_COST_CAP) << CSH));
    const uint32_t wm = WIN ? column_step_limited<MT, NOINDEL, XREP>(L.col, nm, u.p0, row0, u.insw, u.delw, plimit)
                            : column_step<MT, NOINDEL, XREP>(L.col, nm, u.p0, row0, u.insw, u.delw);
    if (WIN) {
        // window mode: the sweep starts at the wave's common start column; a read has no
        // acceptable cell before its own window (the pre-pass saw none), so only the upper
        // end needs a per-lane test.  Row m is only swept when some lane of the wave needs it;
        // a lane that does not has no row-m candidate anyway.
        if (plimit == MT && wm < u.klimit && j <= L.max_n)
            consider<XREP>(L.best, wm, u.m, j, u.min_overlap, thr, indel);
    } else if (XREP) {
        // row-m candidate: the reference looks at it only when the band reached row m,
        // i.e. cost <= k (:433-455); min_n is 0 here.
        if (wm < u.klimit && j <= L.max_n) consider<XREP>(L.best, wm, u.m, j, u.min_overlap, thr, indel, obase);
    } else if (j == L.min_n && L.max_n > L.min_n) {
        int mn = L.min_n;
#ifndef ATR_HOST_EMU
        asm volatile("" : "+v"(mn));                         // keep the re-init out of the loop preheader
#endif
#pragma unroll
        for (int i = 0; i <= MT; ++i) L.col[i] = init_word(i - u.p0, mn, u.sr, u.sq, u.indel) - (uint32_t)obase;
    }
    if (L.scan && j == L.max_n && L.max_n > L.min_n)
        scan_last_column<MT, XREP>(L.best, L.col, u.p0, u.first_p, L.n, u.min_overlap, thr, indel, u.klimit, WIN ? plimit : MT, obase);
}

// ---- reads longer than the origin field reaches (locate_long_kernel) ------------------------------------------
// The origin field of a cell word holds 10 bits (columns -256 .. 767).  A long read is swept with a ROLLING BASE:
// the fields count from column long_base(j), which follows the sweep in steps of 256 columns, 257 .. 512 columns
// behind it; at a step every cell's field is lowered by 256, and a cell whose origin lies before the new range is
// given the lowest value instead.  Such a cell belongs to a path that has consumed more than 512 columns on at most
// m rows: its cost is above k (m + k <= LONG_MAX_SPAN) and stays there, so it is never looked at as a candidate, never
// wins against a cell that could be, and what its origin says does not matter.  A best match keeps the base it was
// found under (the kernel's own variable).
constexpr int LONG_BASE_STEP = 256, LONG_MAX_SPAN = 384;
ATR_DEV int long_base(int j) { return j > 2 * LONG_BASE_STEP ? (((j - 1) / LONG_BASE_STEP) - 1) * LONG_BASE_STEP : 0; }
template <int MT>
ATR_DEV void lane_rebase(LaneState<MT> &L) {
#pragma unroll
    for (int i = 0; i <= MT; ++i) {
        const uint32_t w = L.col[i];
        L.col[i] = (w & ORG_MASK) >= (uint32_t)LONG_BASE_STEP ? w - (uint32_t)LONG_BASE_STEP : (w & ~ORG_MASK);
    }
}

// (refstart, refstop, querystart, querystop, matches, errors, 0, 0) as 8 x int16.
template <int MT>
ATR_DEV void lane_result(const LaneState<MT> &L, const Uniform &u, uint32_t rec[4], int obase = 0) {
    const int cost = (int)(L.best.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    // :476-480: None when the best cost is m + n -- no candidate seen (the key lane_init set; its word does not hold
    // m + n of a long read), or, as in the reference, a candidate of exactly that cost
    if (L.best.key != COST_FIELD_MAX - (u.m + L.n) && cost != u.m + L.n) {
        const int origin = (int)(L.best.word & ORG_MASK) - (int)ORG_BIAS + obase;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = L.best.ref_stop; querystop = L.best.query_stop;
        matches = L.best.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

// ---- tile64 packing arithmetic (shared by the pack kernel and the emulation) -------

// One 32-bit word = 8 bases, base b in bits 4b..4b+3.  `zero_seen` is set when a base
// inside the read translates to code 0.
ATR_DEV uint32_t pack_word(const uint8_t *row, int j0, int n, const uint8_t *table, bool &zero_seen) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int j = j0 + b;
        if (j < n) {
            const uint32_t code = (uint32_t)table[row[j]] & 15u;
            zero_seen = zero_seen || code == 0;
            w |= code << (4 * b);
        }
    }
    return w;
}

// ---- plane64 packing (insert aligner) ---------------------------------------------------
// Same tiles and chunk addresses as tile64, but the 16 bytes of a chunk are FOUR BIT PLANES of
// its 32 bases: word p, bit b = bit p of the 4-bit code of base 32c + b.  The insert kernel
// compares reads 32 bases per boolean op in this form (insert_core.hpp).
// spread[c]: the code of byte c with bit p moved to bit 8p (one plane per byte).
ATR_DEV uint32_t spread_code(uint32_t code) {
    return (code & 1u) | ((code & 2u) << 7) | ((code & 4u) << 14) | ((code & 8u) << 21);
}

// The four plane words of the 32 bases starting at j0 (bases >= n read as code 0).
ATR_DEV void pack_planes_chunk(const uint8_t *row, int j0, int n, const uint32_t *spread, bool &zero_seen, uint32_t out[4]) {
    uint32_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t a = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = j0 + 8 * g + b;
            if (j < n) {
                const uint32_t sp = spread[row[j]];
                zero_seen = zero_seen || sp == 0;
                a |= sp << b;                                  // byte p of a: plane-p bits of bases 8g .. 8g+7
            }
        }
        acc[g] = a;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
        out[p] = ((acc[0] >> (8 * p)) & 0xFFu) | (((acc[1] >> (8 * p)) & 0xFFu) << 8) |
                 (((acc[2] >> (8 * p)) & 0xFFu) << 16) | (((acc[3] >> (8 * p)) & 0xFFu) << 24);
}

}  // namespace atr
#endif
