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

}  // namespace atr
#endif
