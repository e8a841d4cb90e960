// pairs_long_core.hpp -- Aligner.locate for pairs whose reference or query is longer than the packed 32-bit cell
// word reaches (pairs_core.hpp: 320 bases): the reference has no length limit (_align.pyx:266-291).
//
// Same recurrence, tie order and candidate rules as locate_core.hpp / pairs_core.hpp, with a 64-bit cell word
//     [63:48] cost | [47:46] priority | [45:31] matches | [30:0] origin + 2^30
// and the DP column of a pair in GLOBAL memory (cells of row i of the 64 pairs of a wave side by side: every
// access of a wave is one coalesced line).  A fallback, not a throughput path: one pair per lane, the whole
// (windowed, :314-321) matrix.  Costs above k = int(e m) are clamped to k + 1: such a cell is never a candidate
// (every threshold is at most k) and never decides a cell of cost <= k (costs only grow along a path, and the
// smaller cost wins), so what it holds does not matter -- which keeps the 16-bit cost field sufficient for any m.
// The candidate test is the reference's own double comparison `cost <= length * max_error_rate` (:447, :468).
//
// Compiled for gfx950 and, with -DATR_HOST_EMU, for the CPU test emulation.
#ifndef ATR_PAIRS_LONG_CORE_HPP
#define ATR_PAIRS_LONG_CORE_HPP

#include "pairs_core.hpp"

namespace atr {

constexpr int LCSH = 48, LPSH = 46, LMSH = 31;
constexpr uint64_t L_ORG_BIAS = 1ull << 30, L_ORG_MASK = (1ull << 31) - 1ull, L_MAT_MASK = (1ull << 15) - 1ull;
constexpr uint64_t L_COST1 = 1ull << LCSH, L_MATCH1 = 1ull << LMSH;
constexpr uint64_t L_PRIO_INS = 1ull << LPSH, L_PRIO_DEL = 2ull << LPSH, L_PRIO_MASK = 3ull << LPSH;
constexpr int PAIRS_LONG_MAX_K = 32000;              // k + 1 + one indel step must stay inside 16 bits

struct PairLongParams {
    double e;
    int flags, min_overlap, indel_cost, and_mode;
};

struct BestLong {
    long long key;                                   // (matches << 20) | (2^20 - 1 - cost): more matches, then fewer errors
    uint64_t word;
    int ref_stop, query_stop;
    bool found;
};

ATR_DEV uint64_t long_word(int cost, long long origin) { return ((uint64_t)cost << LCSH) | (uint64_t)(origin + (long long)L_ORG_BIAS); }

// Candidate test of _align.pyx:440-455 / :464-474
ATR_DEV void consider_long(BestLong &b, uint64_t w, int ref_stop, int query_stop, int min_overlap, double e) {
    const int cost = (int)(w >> LCSH);
    const long long origin = (long long)(w & L_ORG_MASK) - (long long)L_ORG_BIAS;
    const int length = ref_stop + (int)(origin < 0 ? origin : 0);
    if (length >= min_overlap && (double)cost <= (double)length * e) {
        const int matches = (int)((w >> LMSH) & L_MAT_MASK);
        const long long key = ((long long)matches << 20) | (long long)(0xFFFFF - cost);
        if (key > b.key) { b.key = key; b.word = w; b.ref_stop = ref_stop; b.query_stop = query_stop; b.found = true; }
    }
}

// col: m + 1 cells (stride cs); refc: the reference's m codes (stride rs), filled here; rp / qp: the lane's packed
// reference / query (tile64, chunk 0).  rec: the record as int16 x 8, refstop = -1 for None.
template <bool AND_MODE>
ATR_DEV void locate_pair_long(uint64_t *col, size_t cs, uint8_t *refc, size_t rs, const uint32_t *rp, int m, bool revcomp,
                              const uint32_t *qp, int n, const PairLongParams &p, uint32_t rec[4]) {
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0;
    const bool er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0, eq = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
    int k = (int)(p.e * m);                                            // _align.pyx:312
    if (k < 0) k = -1;
    int indel = p.indel_cost > k ? k + 1 : p.indel_cost;              // beyond k + 1 every indel is unaffordable alike
    if (indel < 1) indel = 1;
    const int cap = k + 1;                                             // clamped cost of every cell above k
    const uint64_t insw = (uint64_t)indel * L_COST1 + L_PRIO_INS, delw = (uint64_t)indel * L_COST1 + L_PRIO_DEL;
    const int max_n = sq ? n : atr_min(n, m + k);                      // :314-321
    const int min_n = eq ? 0 : atr_max(0, n - m - k);
    for (int i = 0; i < m; ++i) {
        uint32_t c = packed_code(rp, revcomp ? m - 1 - i : i);
        if (revcomp) c = bitrev4(c);
        refc[(size_t)i * rs] = (uint8_t)c;
    }
    for (int i = 0; i <= m; ++i) {                                     // :333-352
        long long cost, origin;
        if (!sr && !sq)      { cost = (long long)atr_max(i, min_n) * indel; origin = 0; }
        else if (sr && !sq)  { cost = (long long)min_n * indel;             origin = atr_min(0, min_n - i); }
        else if (!sr && sq)  { cost = (long long)i * indel;                 origin = atr_max(0, min_n - i); }
        else                 { cost = (long long)atr_min(i, min_n) * indel; origin = min_n - i; }
        col[(size_t)i * cs] = long_word((int)(cost > cap ? cap : cost), origin);
    }
    BestLong best;
    best.key = (long long)(0xFFFFF - (m + n));                         // (matches 0, cost m + n): :358-363
    best.word = 0; best.ref_stop = m; best.query_stop = n; best.found = false;
    for (int j = min_n + 1; j <= max_n; ++j) {
        const uint32_t q = packed_code(qp, j - 1);
        long long c0 = (long long)j * indel;
        const uint64_t row0 = sq ? long_word(0, j) : long_word((int)(c0 > cap ? cap : c0), 0);   // :385-388
        uint64_t old_prev = col[0], new_prev = row0;
        col[0] = row0;
        for (int i = 1; i <= m; ++i) {
            const uint64_t old = col[(size_t)i * cs];
            const uint32_t r = refc[(size_t)(i - 1) * rs];
            const bool mis = AND_MODE ? (r & q) == 0u : r != q;                          // :390-393
            const uint64_t cd = mis ? old_prev + L_COST1 : old_prev + L_MATCH1;          // :394-404
            const uint64_t cl = old + delw, cu = new_prev + insw;                        // :405-419
            uint64_t nw = cd < cl ? cd : cl;
            nw = (nw < cu ? nw : cu) & ~L_PRIO_MASK;
            if ((int)(nw >> LCSH) > cap) nw = (nw & (L_COST1 - 1ull)) | ((uint64_t)cap << LCSH);
            col[(size_t)i * cs] = nw;
            old_prev = old;
            new_prev = nw;
        }
        // row-m candidate: looked at only when the band reached row m, i.e. cost <= k (:433-455)
        if (eq && (int)(new_prev >> LCSH) <= k) consider_long(best, new_prev, m, j, p.min_overlap, p.e);
    }
    if (max_n == n) {                                                   // :461-474
        for (int i = er ? 0 : m; i <= m; ++i) {
            const uint64_t w = col[(size_t)i * cs];
            if ((int)(w >> LCSH) <= k) consider_long(best, w, i, n, p.min_overlap, p.e);
        }
    }
    const int cost = (int)(best.word >> LCSH);
    int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
    if (best.found && cost != m + n) {                                  // :476-480
        const long long origin = (long long)(best.word & L_ORG_MASK) - (long long)L_ORG_BIAS;
        if (origin >= 0) querystart = (int)origin; else refstart = (int)(-origin);
        refstop = best.ref_stop; querystop = best.query_stop;
        matches = (int)((best.word >> LMSH) & L_MAT_MASK); errors = cost;
    }
    rec[0] = (uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16);
    rec[1] = (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16);
    rec[2] = (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16);
    rec[3] = 0;
}

}  // namespace atr
#endif
