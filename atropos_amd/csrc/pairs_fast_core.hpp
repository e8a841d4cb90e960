// pairs_fast_core.hpp -- Aligner.locate with a per-pair reference WITHOUT sweeping the whole matrix
// (the aligner MergeOverlapping builds per read pair, commands/trim/modifiers.py:889-894; the DP it
// restates is _align.pyx:266-491).
//
// pairs_core.hpp computes all m x n cells of every pair with the packed-word recurrence (7 VALU ops
// per cell).  Here a pair goes through three cheaper steps; the result is the same record, bit for bit,
// or the pair is handed to pairs_core.hpp ("fallback"):
//
//  1. COSTS.  With unit costs (mismatch = indel = 1, what MergeOverlapping uses) the reference's matrix
//     is the plain edit-distance matrix with the free starts its flags allow, and that matrix is swept
//     32 rows per VALU op with Myers' bit-vector recurrence (pf_myers_step; ~10 ops per 32 cells).  The
//     sweep keeps the EXACT cost of every cell the reference can accept: row m in every column
//     (STOP_WITHIN_SEQ2, _align.pyx:433-455) and the last column in every row (:461-474).
//  2. WHICH CANDIDATES MATTER.  The reference keeps the candidate with the most matches, then the lowest
//     cost, then the first seen.  From a candidate's end cell (i, j) and exact cost c alone:
//        rows consumed  R in [min(i, j - c), min(i, j + c)],  columns consumed C likewise with i and j swapped,
//        matches in [max(R_lb, C_lb) - c, min(i, j, (i + j - c) / 2)].
//     "accept-sure": c <= thr[R_lb] (it IS accepted, with >= lbM matches); "accept-possible": c <= thr[R_ub].
//     M_lb = the largest lbM of an accept-sure candidate (or the caller's `need`) bounds the winner's
//     matches from below, so only accept-possible candidates with ubM >= M_lb -- the THREATS -- can be
//     the result.  (pf_stream_rowm / pf_stream_lastcol / pf_decide)
//  3. PAYLOAD.  The (matches, origin) payload of the threats comes from the packed-word DP on a BAND of
//     diagonals (row-major, one register per diagonal, cells outside count as unreachable;
//     pf_band_sweep).  A path that ends on diagonal d with cost c never leaves the diagonals
//     [d - c, d + c] (it would have to come back by more than c indels), so a band that holds
//     [d_Y - c_Y, d_Y + c_Y] of every threat Y -- c_Y is known exactly from step 1 -- holds ALL optimal
//     paths of every threat and of every cell on them: those cells get their true cost, every
//     predecessor that ties for the minimum is exact too, every other one is at least its true value, so
//     each choice and with it the payload is the reference's.  Cells of the band that are no threats
//     may be off (too high a cost, never too low): with its own cost instead of the exact one such a
//     cell fails the threat test again (the test is monotone in the cost), so it is never looked at.
//     Pairs whose threats need more than PF_MAX_W diagonals take the full sweep.
//
// Requires STOP_WITHIN_SEQ2, indel cost 1, the literal compare (no wildcard flags) on A C G T N codes;
// everything else takes pairs_core.hpp as before.
//
// Compiled for gfx950 and, with -DATR_HOST_EMU, for the CPU test emulation.
#ifndef ATR_PAIRS_FAST_CORE_HPP
#define ATR_PAIRS_FAST_CORE_HPP

#include "pairs_core.hpp"

namespace atr {

constexpr long long PAIRS_FAST_MIN_PAIRS = 262144;      // below: the full sweep (the pipeline's fixed cost, pairs_kernel.hip)
constexpr int PF_CLASSES = 8;                       // band widths 16, 32, .. 128 cells
constexpr int PF_MAX_W = 16 * PF_CLASSES;
constexpr int PF_ROW_BINS = 28;                     // (rows swept) / 12, clamped
constexpr int PF_ROW_BIN_SHIFT = 12;
constexpr int PF_FALLBACK_BIN = PF_CLASSES * PF_ROW_BINS;        // pairs for the full sweep
constexpr int PF_BINS = 256;                        // histogram width (a multiple of 256 for the scan kernels)
constexpr int PF_MAX_K = 126;                       // costs live in 7 bits of the row-m byte
constexpr int PF_TAB_ROWS = 5;                      // match masks by pf_code_row: A C G T N
ATR_DEV int pf_class_width(int cls) { return 16 * (cls + 1); }

struct PairFastParams {
    PairParams pp;
    int16_t g_ap[PAIRS_MAX_LEN + 2];                // g_ap[x] = max c with c <= floor((x + c) * e)   (accept-possible)
    int16_t g_as[PAIRS_MAX_LEN + 2];                // g_as[x] = max c <= x with c <= floor((x - c) * e), -1: none (accept-sure)
};

// task of the banded pass: 16 bytes
struct PairTask {
    uint32_t pair;
    int16_t d_lo, row_first;                        // first diagonal (j - i) of the band; first row swept
    int16_t row_last, mlb;                          // last row swept; lower bound of the winner's matches
    int16_t cand_first, reserved;                   // first row with a last-column threat (0: none)
};

// ---- 1. costs: Myers / Hyyro bit-vector sweep, rows END-ALIGNED in NW words ------------------------
// Row i (1-based) is bit p0 + i - 1, p0 = 32 NW - m, so row m is the top bit of the last word in every
// lane whatever its m: the bit a left shift pushes out of the horizontal deltas IS row m's delta.  The
// pad bits below row 1 play the part of row 0: with START_WITHIN_SEQ2 (cost 0 along row 0) they match
// every base and stay at cost 0; without it (cost j in column j) they match nothing and take the +1
// that enters at the bottom of every column (hin).
template <int NW>
struct PfMyers {
    uint32_t pv[NW], mv[NW];                        // vertical +1 / -1 deltas of the current column
    int score;                                      // D[m][j]
};

template <int NW>
ATR_DEV void pf_myers_init(PfMyers<NW> &S, int m, bool sr) {
    const int p0 = 32 * NW - m;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int lo = 32 * w;
        const uint32_t rows = p0 <= lo ? ~0u : (p0 >= lo + 32 ? 0u : ~0u << (p0 - lo));
        S.pv[w] = sr ? 0u : rows;                   // column 0: cost i (rows) or 0 everywhere (START_WITHIN_SEQ1)
        S.mv[w] = 0u;
    }
    S.score = sr ? 0 : m;
}

// eq[w]: rows (and pads) matching this column's base.  hin: the horizontal delta of row 0 (0 or 1).
template <int NW>
ATR_DEV void pf_myers_step(PfMyers<NW> &S, const uint32_t (&eq)[NW], uint32_t hin) {
    uint32_t xv[NW], ph[NW], mh[NW];
    uint32_t carry = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t e = eq[w], pv = S.pv[w], mv = S.mv[w];
        xv[w] = e | mv;
        const uint32_t t = e & pv;
#ifdef ATR_HOST_EMU
        const uint64_t sum = (uint64_t)t + (uint64_t)pv + (uint64_t)carry;       // the add runs through all words
        const uint32_t s = (uint32_t)sum;
        carry = (uint32_t)(sum >> 32);
#else
        unsigned cout;
        const uint32_t s = __builtin_addc(t, pv, carry, &cout);                  // v_addc_co_u32 (the 64-bit form compiled to v_lshl_add_u64)
        carry = cout;
#endif
        const uint32_t xh = (s ^ pv) | e;
        ph[w] = mv | ~(xh | pv);
        mh[w] = pv & xh;
    }
    S.score += (int)(ph[NW - 1] >> 31) - (int)(mh[NW - 1] >> 31);
#pragma unroll
    for (int w = NW - 1; w > 0; --w) {
        ph[w] = (ph[w] << 1) | (ph[w - 1] >> 31);
        mh[w] = (mh[w] << 1) | (mh[w - 1] >> 31);
    }
    ph[0] = (ph[0] << 1) | hin;
    mh[0] <<= 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        S.pv[w] = mh[w] | ~(xv[w] | ph[w]);
        S.mv[w] = ph[w] & xv[w];
    }
}

// table row of a 4-bit DNA15 code: A C G T N -> 0 .. 4, eight codes at a time (one per nibble): code / 2 for the
// one-bit codes 1 2 4, minus one for T (8 -> 3), minus three for N (15 -> 7 -> 4).  Other codes land on some row
// too (possibly beyond the table); pf_codes_known tells, and such a pair takes the full sweep.
ATR_DEV uint32_t pf_code_rows8(uint32_t codes) {
    const uint32_t hi = (codes >> 3) & 0x11111111u;                        // T or N
    const uint32_t nn = hi & (codes >> 2);                                 // N (bits 3 and 2)
    return (((codes >> 1) & 0x77777777u) - hi - (nn << 1)) & 0x77777777u;
}
ATR_DEV int pf_code_row(uint32_t code) { return (int)(pf_code_rows8(code) & 7u); }
// are all eight nibbles one of 0 (beyond the read), A C G T (one bit) or N (four bits)?
ATR_DEV bool pf_codes_known(uint32_t codes) {
    uint32_t p = codes - ((codes >> 1) & 0x55555555u);
    p = (p & 0x33333333u) + ((p >> 2) & 0x33333333u);           // bits per nibble: 0 .. 4
    return (p & 0x22222222u) == 0u;                             // two or three bits: some other IUPAC code
}

ATR_DEV uint32_t pf_bitrev32(uint32_t v) {
#ifdef ATR_HOST_EMU
    uint32_t r = 0u;
    for (int b = 0; b < 32; ++b) r |= ((v >> b) & 1u) << (31 - b);
    return r;
#else
    return __builtin_bitreverse32(v);
#endif
}

// dword z (eight codes) of a tile64-packed read of ndw dwords, 0 outside
ATR_DEV uint32_t pf_read_dword(const uint32_t *q, int ndw, int z) {
    if (z < 0 || z >= ndw) return 0u;
    return q[(size_t)(z >> 2) * 256 + (z & 3)];
}
// the eight codes at the 0-based positions s .. s + 7 (any s; positions outside the read give 0)
ATR_DEV uint32_t pf_codes8(const uint32_t *q, int ndw, int s) {
    const int z = s >> 3;
    const uint32_t sh = 4u * (uint32_t)(s & 7);
    const uint32_t lo = pf_read_dword(q, ndw, z), hi = pf_read_dword(q, ndw, z + 1);
    return sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
}
// the reference codes of rows i0 .. i0 + 7 (1-based; 0 outside 1 .. m).  The reverse complement of a packed
// read is its code string read backwards with every code bit-reversed (util/__init__.py:67-88 on bit codes):
// one v_bfrev_b32 per dword.
ATR_DEV uint32_t pf_ref_codes8(const uint32_t *rp, int ndw, int m, bool revcomp, int i0) {
    if (!revcomp) return pf_codes8(rp, ndw, i0 - 1);
    return pf_bitrev32(pf_codes8(rp, ndw, m - i0 - 7));
}

// flags at bits 0, 4, .. 28 -> bits 0 .. 7
ATR_DEV uint32_t pf_compact8(uint32_t x) {
    uint32_t t = (x | (x >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}

// match masks of one lane: tab[(row * NW + w) * ts], end-aligned, pads = `pad_match`.  Returns false when
// the reference holds a code outside A C G T N.
template <int NW>
ATR_DEV bool pf_build_masks(uint32_t *tab, int ts, const uint32_t *rp, int ndw, int m, bool revcomp, bool pad_match) {
    const int p0 = 32 * NW - m;
    bool known = true;
#pragma unroll 1
    for (int w = 0; w < NW; ++w) {
        uint32_t pl[4] = {0u, 0u, 0u, 0u};                     // bit planes of the codes at positions 32 w .. 32 w + 31
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = pf_ref_codes8(rp, ndw, m, revcomp, 32 * w + 8 * q - p0 + 1);
            known = known && pf_codes_known(v);
#pragma unroll
            for (int b = 0; b < 4; ++b) pl[b] |= pf_compact8((v >> b) & 0x11111111u) << (8 * q);
        }
        const int lo = 32 * w;
        const uint32_t pad = pad_match ? (p0 <= lo ? 0u : (p0 >= lo + 32 ? ~0u : ~(~0u << (p0 - lo)))) : 0u;
        const uint32_t all = pl[0] & pl[1] & pl[2] & pl[3];
        tab[(size_t)(0 * NW + w) * ts] = (pl[0] & ~(pl[1] | pl[2] | pl[3])) | pad;       // A = 1
        tab[(size_t)(1 * NW + w) * ts] = (pl[1] & ~(pl[0] | pl[2] | pl[3])) | pad;       // C = 2
        tab[(size_t)(2 * NW + w) * ts] = (pl[2] & ~(pl[0] | pl[1] | pl[3])) | pad;       // G = 4
        tab[(size_t)(3 * NW + w) * ts] = (pl[3] & ~(pl[0] | pl[1] | pl[2])) | pad;       // T = 8
        tab[(size_t)(4 * NW + w) * ts] = all | pad;                                      // N = 15 equals N only (:390-391)
    }
    return known;
}

// ---- 2. which candidates matter --------------------------------------------------------------------
struct PfDecision {
    int kind;                                       // 0: no alignment (record None), 1: band task, 2: full sweep
    int cls;
    PairTask task;
};

ATR_DEV int pf_min3i(int a, int b, int c) { return atr_min(atr_min(a, b), c); }

// One candidate end cell (ie, je) of exact cost c.  Pass 1: the lower bound of the winner's matches; pass 2: the
// threats' diagonals and rows.
struct PfScan {
    int mlb, dbest;                                 // pass 1
    int lo, hi, rl, cand_first;                     // pass 2
    bool have;
};

ATR_DEV void pf_pass1(PfScan &S, int ie, int je, int c, int thr_rows, int gas, int min_overlap) {
    // accept-sure: c <= thr[R_lb], R_lb = min(ie, je - c) >= min_overlap  <=>  c <= thr[ie] and c <= g_as[je]
    if (c <= thr_rows && c <= gas) {
        const int rlb = atr_min(ie, je - c);
        if (rlb >= min_overlap && rlb >= 1) {
            // matches = rows - insertions - mismatches = columns - deletions - mismatches >= max(rows, columns) - c
            const int lbm = atr_max(rlb, atr_min(je, ie - c)) - c;
            if (lbm > S.mlb) { S.mlb = lbm; S.dbest = je - ie; }
        }
    }
}

// Is the candidate ending in (ie, je) with cost c a threat?  Monotone in c: with a cost that is too high a
// non-threat stays one.  thr_rows = thr[ie], gap = g_ap[je].
ATR_DEV bool pf_is_threat(int ie, int je, int c, int thr_rows, int gap, int min_overlap, int mlb) {
    // accept-possible: c <= thr[R_ub], R_ub = min(ie, je + c)  <=>  c <= thr[ie] and c <= g_ap[je]
    return c <= thr_rows && c <= gap && atr_min(ie, je + c) >= min_overlap && pf_min3i(ie, je, (ie + je - c) >> 1) >= mlb;
}

// The analysis keeps no candidate list (round 5: the per-lane list -- 96 entries of 16 bits -- was two thirds of the
// cost kernel's LDS and held it at two waves per SIMD).  Pass 1 (the lower bound of the winner's matches) runs along
// the sweep and then over the last column; pass 2 (the threats under the FINAL bound) needs the row-m costs again:
// the sweep leaves row m's horizontal deltas behind -- two bits per column, D[m][j] - D[m][j-1] in {-1, 0, +1}, which
// Myers' recurrence produces anyway -- and the exact costs are replayed from them.  (Taking the threats up during the
// sweep under the bound as it stood was tried first: the cells in front of the winner on row m come before the bound
// rises, and the bands went from 64 to 96 diagonals.)  The last column comes from the final vertical deltas, which
// are still in registers, in two walks.
ATR_DEV void pf_stream_init(PfScan &S, int need) {
    S.mlb = atr_max(need, 1) - 1; S.dbest = 0;
    S.lo = 0x7fff; S.hi = -0x7fff; S.rl = 0; S.cand_first = 0; S.have = false;
}
ATR_DEV int pf_mlb_eff(const PfScan &S, int need) { return atr_max(S.mlb, atr_max(need, 1)); }

// threat test and band extents under the bound `mlb`
ATR_DEV void pf_take_threat(PfScan &S, int ie, int je, int c, int thr_rows, int gap, int min_overlap, bool lastcol, int mlb) {
    if (pf_is_threat(ie, je, c, thr_rows, gap, min_overlap, mlb)) {
        const int d = je - ie;
        S.have = true;
        S.lo = atr_min(S.lo, d - c);                // all optimal paths of the threat: diagonals d - c .. d + c
        S.hi = atr_max(S.hi, d + c);
        S.rl = atr_max(S.rl, ie);
        if (lastcol && S.cand_first == 0) S.cand_first = ie;
    }
}

// row m, column j, exact cost `score`: accept-possible  <=>  c <= thr[m] and c <= g_ap[j].  Pass 1 during the sweep
// (returns whether the cell is a candidate at all), pass 2 in the replay with the final bound.
ATR_DEV bool pf_rowm_pass1(PfScan &S, int m, int j, int score, int k, int gap_j, int gas_j, int min_overlap) {
    if (score > atr_min(k, gap_j)) return false;
    pf_pass1(S, m, j, score, k, gas_j, min_overlap);
    return true;
}
ATR_DEV void pf_rowm_pass2(PfScan &S, int m, int j, int score, int k, int gap_j, int min_overlap, int mlb) {
    if (score <= atr_min(k, gap_j)) pf_take_threat(S, m, j, score, k, gap_j, min_overlap, false, mlb);
}

// last column: D[i][n] = D[0][n] + the vertical deltas up to row i (the pads below row 1 carry none).  The word index
// is static (pv / mv live in registers), the row of a bit depends on the lane's p0.  second == false: pass 1 alone.
template <int NW>
ATR_DEV void pf_stream_lastcol(PfScan &S, const uint32_t (&pv)[NW], const uint32_t (&mv)[NW], int m, int n, bool sq, bool er,
                               const int16_t *thr, int gap_n, int gas_n, int min_overlap, int need, bool second) {
    const int p0 = 32 * NW - m;
    const int mlb = pf_mlb_eff(S, need);            // (second pass: the final bound)
    int d = sq ? 0 : n;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t pw = pv[w], mw = mv[w];
#ifndef ATR_PF_WALK_UNROLL
#define ATR_PF_WALK_UNROLL 8                         // (the per-row threshold is an LDS read: eight of them in flight)
#endif
#pragma unroll ATR_PF_WALK_UNROLL
        for (int b = 0; b < 32; ++b) {
            const int i = 32 * w + b - p0 + 1;
            d += (int)(pw & 1u) - (int)(mw & 1u);
            pw >>= 1; mw >>= 1;
            if (i >= 1 && (er || i == m) && d <= gap_n && d <= (int)thr[i]) {
                if (!second) pf_pass1(S, i, n, d, (int)thr[i], gas_n, min_overlap);
                else pf_take_threat(S, i, n, d, (int)thr[i], gap_n, min_overlap, true, mlb);
            }
        }
    }
}

// What the scan found -> the pair's decision.  need: the caller only cares about alignments with at least this many
// matches (>= 1).
ATR_DEV void pf_decide(PfScan &S, int m, int n, int need, uint32_t pair, PfDecision &D) {
    S.mlb = pf_mlb_eff(S, need);
    D.cls = 0;
    if (!S.have) { D.kind = 0; return; }                        // nothing can be accepted (with >= need matches)
    int lo = atr_max(S.lo, -m), hi = atr_min(S.hi, n);
    const int width = hi - lo + 1;
    if (width > PF_MAX_W) {
        D.kind = 2;
        D.task.d_lo = (int16_t)lo; D.task.row_first = (int16_t)hi; D.task.mlb = (int16_t)S.mlb; D.task.row_last = (int16_t)S.rl;   // (diagnostics)
        return;
    }
    const int cls = (width + 15) / 16 - 1;
    // the band of the class, kept inside the matrix' diagonals where that is possible
    const int wb = pf_class_width(cls);
    if (lo + wb - 1 > n) lo = atr_max(-m, n - wb + 1);
    D.kind = 1; D.cls = cls;
    D.task.pair = pair;
    D.task.d_lo = (int16_t)lo;
    D.task.row_first = (int16_t)atr_max(1, -(lo + wb - 1));     // the first row with a cell at column >= 0
    D.task.row_last = (int16_t)S.rl;
    D.task.mlb = (int16_t)S.mlb;
    D.task.cand_first = (int16_t)S.cand_first;
    D.task.reserved = 0;
}

ATR_DEV int pf_task_bin(const PfDecision &D) {
    if (D.kind != 1) return PF_FALLBACK_BIN;
    const int rows = (int)D.task.row_last - (int)D.task.row_first + 1;
    return D.cls * PF_ROW_BINS + atr_min(PF_ROW_BINS - 1, atr_max(rows, 0) / PF_ROW_BIN_SHIFT);
}

// ---- 3. payload: the banded packed-word DP -----------------------------------------------------------
// Cell word and tie-break as locate_core.hpp (XREP: the payload counts diagonal mismatches), origin bias
// PAIRS_ORG_BIAS as pairs_core.hpp.
constexpr uint32_t PF_INF = ((uint32_t)INIT_COST_CAP << CSH) | (uint32_t)PAIRS_ORG_BIAS;

// bit 3 of every nibble = "the nibble is not zero"
ATR_DEV uint32_t pf_nibble_any(uint32_t v) {
    return ((v & 0x77777777u) + 0x77777777u) | v;   // (the low three bits carry into bit 3; the other bits are not looked at)
}

// Keeps the instruction scheduler from hoisting the diagonal candidates of the whole (unrolled) row in front of
// the serial min3 chain: that doubles the live registers of a wide band.
ATR_DEV void pf_sched_fence() {
#ifndef ATR_HOST_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

ATR_DEV void pf_opaque(uint32_t &v) {
#ifndef ATR_HOST_EMU
    asm volatile("" : "+v"(v));
#endif
}

struct PfBandLane {
    int d_lo, row_first, row_last, cand_first, mlb, m, n, n_sweep;
    bool scan_last, live;
};

// The row-major sweep of one lane over its rows row_first .. row_last (`nrows` = the wave's trip count; a lane
// that has done its own rows keeps its last row).  rs / qs: the lane's reference codes and query bases as
// streams of eight 4-bit codes per dword (PfRefStream / PfQueryStream): nibble t of rs = the code of row row_first + t,
// nibble s of qs = the base of column row_first + d_lo + s, 0 outside the read.
// Column 0 without special cases: code 0 (the columns j <= 0 and those behind the read) matches every row, and
// the band starts from VIRTUAL cells left of the matrix -- with START_WITHIN_SEQ1 cost 0 and origin j - i (the
// diagonal through them reaches column 0 in the cell (i - j, 0) = cost 0, origin -(i - j): _align.pyx:343-346,
// :349-352), else unreachable, so that column 0 holds the i insertions from (0, 0) (:336-342, :347-348).
// The two streams of a lane as READERS over the packed reads (tile64: pf_read_dword): consecutive stream dwords are
// consecutive eight-code windows of the read at ONE bit offset, so every raw dword is loaded once and funnel-shifted
// against its neighbour.  Round 5: the band pass reads its streams straight from the batch (the L2), one dword of each
// every eight rows and one step ahead of its use, instead of staging whole streams in LDS first -- 21 KB per wave for
// 2 x 250-base pairs, which held the wide band kernels at 1.5 waves per SIMD.
struct PfQueryStream {                              // dword t = the codes at positions s0 + 8 t .. (0 outside the read)
    const uint32_t *p;
    int ndw, z;
    uint32_t sh, prev;
    ATR_DEV_MEMBER void init(const uint32_t *q, int ndw_, int s0) {
        p = q; ndw = ndw_; z = s0 >> 3; sh = 4u * (uint32_t)(s0 & 7);
        prev = pf_read_dword(p, ndw, z);
    }
    ATR_DEV_MEMBER uint32_t next() {
        const uint32_t a = pf_read_dword(p, ndw, ++z);
        const uint32_t out = sh ? ((prev >> sh) | (a << (32u - sh))) : prev;
        prev = a;
        return out;
    }
};
struct PfRefStream {                                // dword t = the reference codes of rows row_first + 8 t ..
    const uint32_t *p;
    int ndw, z;
    uint32_t sh, carry;
    bool revcomp;
    ATR_DEV_MEMBER void init(const uint32_t *r, int ndw_, int m, bool rc, int row_first) {
        p = r; ndw = ndw_; revcomp = rc;
        // forward: positions row_first - 1 + 8 t; reverse complement: bitrev(codes at m - row_first - 7 - 8 t ..), the raw
        // dwords going DOWN (pf_ref_codes8)
        const int s0 = rc ? m - row_first - 7 : row_first - 1;
        z = s0 >> 3; sh = 4u * (uint32_t)(s0 & 7);
        carry = pf_read_dword(p, ndw, rc ? z + 1 : z);
    }
    ATR_DEV_MEMBER uint32_t next() {
        if (!revcomp) {
            const uint32_t a = pf_read_dword(p, ndw, ++z);
            const uint32_t out = sh ? ((carry >> sh) | (a << (32u - sh))) : carry;
            carry = a;
            return out;
        }
        const uint32_t a = pf_read_dword(p, ndw, z--);
        const uint32_t out = pf_bitrev32(sh ? ((a >> sh) | (carry << (32u - sh))) : a);
        carry = a;
        return out;
    }
};

template <int WB>
ATR_DEV void pf_band_sweep(const PfBandLane &L, int nrows, PfRefStream &rs, PfQueryStream &qs,
                           const PairParams &p, const int16_t *thr, const int16_t *g_ap, uint32_t rec[4]) {
    constexpr int NQ = WB / 8;
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    const uint32_t insw = COST1 + PRIO_INS, delw = COST1 + PRIO_DEL;
    const int m = L.m, n = L.n;
    uint32_t cell[WB];
    {   // the row before the first (row i0 = row_first - 1)
        const int i0 = L.row_first - 1;
#pragma unroll
        for (int c = 0; c < WB; ++c) {
            const int j = i0 + L.d_lo + c;
            uint32_t w = PF_INF;
            if (j > 0) {
                if (i0 == 0)                          // row 0: cost 0 or j, origin j (:385-388)
                    w = sq ? ((uint32_t)PAIRS_ORG_BIAS + (uint32_t)j)
                           : ((uint32_t)PAIRS_ORG_BIAS | ((uint32_t)atr_min(j, INIT_COST_CAP) << CSH));
            } else if (sr) {
                if (j - i0 >= -m) w = (uint32_t)(PAIRS_ORG_BIAS + j - i0);       // cost 0, origin j - i0
            } else if (j == 0) {
                w = (uint32_t)PAIRS_ORG_BIAS | ((uint32_t)atr_min(i0, INIT_COST_CAP) << CSH);   // (i0, 0): i0 insertions
            }
            cell[c] = w;
        }
    }
    Best bl, bm;                                     // last-column candidates / row-m candidates
    bl.key = COST_FIELD_MAX - (m + n);
    bl.word = (uint32_t)(m + n) << CSH;
    bl.ref_stop = m; bl.query_stop = n; bl.matches = 0;
    bm = bl;
    const bool er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0;
    const int gap_n = (int)g_ap[n];
    const int own_rows = L.live ? L.row_last - L.row_first + 1 : 0;
    uint32_t qw[NQ + 1];
#pragma unroll
    for (int t = 0; t <= NQ; ++t) qw[t] = qs.next();
    uint32_t rw = rs.next();
    uint32_t qnext = qs.next(), rnext = rs.next();          // the dwords of the next eight rows: requested a step ahead
#pragma unroll 1
    for (int t = 0; t < nrows; ++t) {
        if (t < own_rows) {
            const int i = L.row_first + t;
            const uint32_t code = rw & 15u;
            uint32_t rrep = code | (code << 4);
            rrep |= rrep << 8;
            rrep |= rrep << 16;
            uint32_t mis[NQ];
#pragma unroll
            for (int g = 0; g < NQ; ++g) mis[g] = pf_nibble_any(qw[g] ^ rrep) & pf_nibble_any(qw[g]);   // differs, and is a base
            uint32_t left = PF_INF;
#pragma unroll
            for (int c = 0; c < WB; ++c) {
                const uint32_t bit = atr_bfe1(mis[c >> 3], 4 * (c & 7) + 3);
                const uint32_t cd = atr_mad24(bit, COST1 + MATCH1, cell[c]);
                const uint32_t up = c + 1 < WB ? cell[c + 1 < WB ? c + 1 : 0] : PF_INF;
                const uint32_t nw = atr_minu(atr_minu(cd, left + delw), up + insw) & ~PRIO_MASK;
                cell[c] = nw;
                left = nw;
                if ((c & 7) == 7) pf_sched_fence();
            }
        }
        // next row: slide the window by one base, next reference code
#pragma unroll
        for (int g = 0; g < NQ; ++g) qw[g] = (qw[g] >> 4) | (qw[g + 1] << 28);
        qw[NQ] >>= 4;
        rw >>= 4;
        if ((t & 7) == 7) {
            qw[NQ] = qnext;
            rw = rnext;
            qnext = qs.next();
            rnext = rs.next();
        }
    }
    // Last-column candidates (:461-474, rows in increasing order) -- AFTER the sweep: a register that has passed column n
    // keeps what it held there.  Behind the read every code is 0 = "matches every row", so the diagonal step leaves the
    // word as it is (no mismatch: no cost, no payload), and it wins the min3: the neighbours of (i + s, n + s) are a real
    // last-column cell and other cells that slid out of the matrix the same way, i.e. D(i', n) + |i - i'| for some rows
    // i' -- never below D(i, n) where that is exact (costs are 1-Lipschitz along a column; band values are upper
    // bounds) -- and a tie goes to the diagonal step (priority 0).  So cell[c] IS the cell (n - d_lo - c, n) for every
    // threat, whose band holds its optimal paths; any other cell's word may have been lowered by a neighbour, never below
    // its true cost, and fails the threat test again.  (Rounds 2-5 picked the cell out of the band after every row: a
    // select tree of WB - 1 v_cndmask, 3.5 % of the 2 x 150 call and 8 % of the 2 x 250 one.)
    if (L.live && L.scan_last && L.cand_first != 0) {
#pragma unroll
        for (int c = WB - 1; c >= 0; --c) {
            const int i = n - L.d_lo - c;
            if (i >= L.cand_first && i >= L.row_first && i <= L.row_last && (er || i == m)) {
                const uint32_t w = cell[c];
                if (pf_is_threat(i, n, (int)(w >> CSH), (int)thr[i], gap_n, p.min_overlap, L.mlb))
                    consider<true, PAIRS_ORG_BIAS>(bl, w, i, n, p.min_overlap, thr, 1);
            }
        }
    }
    // row m (the last row of the lanes that have row-m threats): its cells in column order (:433-455)
    if (L.live && L.row_last == m) {
        const int km = (int)thr[m];
#pragma unroll
        for (int c = 0; c < WB; ++c) {
            const int j = m + L.d_lo + c;
            if (j >= 1 && j <= L.n_sweep && pf_is_threat(m, j, (int)(cell[c] >> CSH), km, (int)g_ap[j], p.min_overlap, L.mlb))
                consider<true, PAIRS_ORG_BIAS>(bm, cell[c], m, j, p.min_overlap, thr, 1);
        }
    }
    if (bm.key >= bl.key) bl = bm;                   // the reference sees the row-m cells first: they keep ties
    const int cost = (int)(bl.word >> CSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (cost != m + n) {
        const int origin = (int)(bl.word & ORG_MASK) - PAIRS_ORG_BIAS;
        if (origin >= 0) querystart = origin; else refstart = -origin;
        refstop = bl.ref_stop; querystop = bl.query_stop;
        matches = bl.matches; errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

// Host side: does the fast pipeline apply, and its tables.
inline bool pairs_fast_applies(double e, int flags, int wildcard_ref, int wildcard_query, int indel_cost, int ref_max_len,
                               int query_max_len) {
    if (!(flags & ATR_STOP_WITHIN_SEQ2) || wildcard_ref || wildcard_query || indel_cost != 1) return false;
    if (ref_max_len < 1 || query_max_len < 1 || ref_max_len > PAIRS_MAX_LEN || query_max_len > PAIRS_MAX_LEN) return false;
    const double kd = e * (double)ref_max_len;
    return kd >= 0.0 && kd < (double)PF_MAX_K && e < 1.0;
}

inline void pairs_fast_tables(double e, PairFastParams &fp) {
    for (int x = 0; x < PAIRS_MAX_LEN + 2; ++x) {
        int gap = -1, gas = -1;
        for (int c = 0; c <= PF_MAX_K + 1; ++c) {
            if ((double)c <= std::floor((double)(x + c) * e)) gap = c; else break;      // downward closed in c (e < 1)
        }
        for (int c = 0; c <= x && c <= PF_MAX_K + 1; ++c) {
            if ((double)c <= std::floor((double)(x - c) * e)) gas = c; else break;
        }
        fp.g_ap[x] = (int16_t)gap;
        fp.g_as[x] = (int16_t)gas;
    }
}

}  // namespace atr
#endif
