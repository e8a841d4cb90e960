// linked_core.hpp -- per-lane arithmetic of the fused linked-adapter pipeline.
//
// The reference matches a set of linked adapters (`-a ^FRONT...BACK`) read by read:
// AdapterCutter._best_match (atropos/commands/trim/modifiers.py:107-122) calls
// LinkedAdapter.match_to (atropos/adapters/__init__.py:671-690) for every adapter, which is
//     front = Adapter(FRONT, PREFIX).match_to(read)              (anchored 5': flags 8)
//     back  = Adapter(BACK, BACK).match_to(read[front.rstop:])   (regular 3': flags 14) if front
// with Adapter.match_to's exact-match shortcut and acceptance test around each alignment
// (:338-400).  Here ONE pass over the packed batch does all of it per read (one read per lane):
//
//   front stage   every 5' adapter against the first m + k bases: literal compare (the shortcut),
//                 then an exact-piece test (the adapter cut into k + 1 pieces: an alignment with at
//                 most k errors leaves one of them intact, at most k bases off its place); only
//                 (read, adapter) pairs that pass it without a literal occurrence run the exact
//                 banded DP (filter_core.hpp, band_locate_prefix) -- collected over several tiles
//                 and run densely, each lane against its OWN (read, adapter) task.
//   back stage    the read keeps the lane; the 3' adapter of the front that matched is swept with
//                 the bit-parallel pre-pass of filter_core.hpp, match masks and thresholds fetched
//                 from the lane's own adapter block in LDS.  read[front.rstop:] is not re-packed:
//                 the bases before s = front.rstop are fed to the sweep as "match nothing", which
//                 leaves the initial column of a 3' adapter (cost i in row i) unchanged, so column
//                 s plays the part of column 0 and all coordinates are re-based by s at the end.
//   unresolved reads go to the banded / windowed DP kernels (locate_fast.hpp), binned per adapter.
//
// Compiled twice like the other *_core.hpp files (hipcc; g++ -DATR_HOST_EMU for tests/emu).
#ifndef ATR_LINKED_CORE_HPP
#define ATR_LINKED_CORE_HPP

#include "filter_core.hpp"

namespace atr {

constexpr int LINKED_MAX = 4;                       // linked adapters per set
constexpr int FRONT_MAX_M = 32;                     // 5' part: m + int(e*m) <= 32 -- chunk 0 holds every column it can touch
constexpr int FRONT_STREAM = 2 + FRONT_MAX_M / 8 + 1;  // stream dwords band_rows reads for m <= 32 (two for the window + one per eight rows)

// One anchored 5' adapter.  Plain data: uploaded once per set, copied to LDS by every workgroup.
constexpr int FRONT_MAX_PIECES = 8;                 // k + 1 <= 8 (2k + 1 <= BAND_W)
struct FrontParams {
    uint32_t pex_code[FRONT_MAX_PIECES], pex_mask[FRONT_MAX_PIECES];   // the k + 1 pieces (<= 8 bases each) as nibbles + masks
    int32_t pex_off[FRONT_MAX_PIECES];              // adapter offset of each piece
    uint32_t code[4], code_mask[4];                 // the adapter as packed nibbles + the mask of its m nibbles (literal compare)
    uint32_t rrep[FRONT_MAX_M];                     // code of row i in all eight nibbles (banded DP)
    int16_t thr[FRONT_MAX_M + 2];                   // floor(L * e): the DP's candidate test (_align.pyx:447, :468)
    int16_t pf_thr[FRONT_MAX_M + 2];                // largest c with c / L <= e in double DIVISION: match_to's own test (:386-398)
    int32_t m, k, min_overlap, indel;               // indel: effective cost min(indel_cost, k + 1)
    int32_t noindel, accept_full, group, npieces;   // accept_full: the literal shortcut applies (:351-367)
};

// One regular 3' adapter as the pre-pass sees it (FilterParams with the thresholds in memory).
struct BackParams {
    uint32_t peq[16][2];                            // FilterParams::peq as (lo, hi) words
    int32_t thr_row[FILTER_MAX_M + 1];              // FilterParams::thr_row
    int16_t pf_thr[FILTER_MAX_M + 2];
    uint32_t tail;                                  // FilterParams::tail (NARROW mode)
    int32_t rows, m, k, min_overlap, indel, accept_full, reserved;
    uint32_t cert[FILTER_CERT_T + 1];               // FilterParams::cert
};

struct LinkedParams {
    int32_t n, and_mode, wide, ngroups;             // wide: two-word bit-vectors in the back sweep
    uint32_t group_mask[LINKED_MAX];                // adapters of DP group g (same length, thresholds, costs)
    int32_t group_first[LINKED_MAX];                // a member of the group (its wave-uniform parameters)
    // excl[a]: the 5' adapters that cannot match a read that STARTS WITH adapter a verbatim (their edit distance to
    // every prefix such a read can offer exceeds their k; literal comparison only) -- linked_host.hpp.  A read
    // with a literal occurrence whose mask covers all the others is through with the 5' stage.
    uint32_t excl[LINKED_MAX];
    int32_t pex_shared, reserved[3];                // every 5' part is cut into the same pieces (count, offsets, k): the read's
                                                    // windows are cut once for all adapters (front_pex_candidates_shared)
    FrontParams f[LINKED_MAX];
    BackParams b[LINKED_MAX];
};

// Random-match-probability filter of Adapter.match_to (optional, device tables)
struct LinkedRmp {
    const double *front[LINKED_MAX];
    const double *back[LINKED_MAX];
    double front_max[LINKED_MAX], back_max[LINKED_MAX];
    int32_t front_ld[LINKED_MAX], back_ld[LINKED_MAX];
};

// What the band / window kernels need to finish a record of ONE 3' adapter (wave-uniform).
struct LinkedPost {
    int16_t pf_thr[FILTER_MAX_M + 2];
    int32_t m, min_overlap, accept_full, rmp_ld;
    const double *rmp;
    double max_rmp;
};

// per-lane view of a BackParams block (filter_decide's parameter type)
struct LaneFilterParams {
    int rows, and_mode;
    uint32_t tail;
    const int32_t *thr_row;
    const uint32_t *cert;                           // FilterParams::cert of the lane's adapter
    uint64_t cert_sub = 0ull;                       // (the substitution certificate wants a diagonal view of the read: not here)
    uint32_t tailx[3] = {0u, 0u, 0u};               // (extended NARROW mode: not in the linked pipeline)
};

// ---- Adapter.match_to's acceptance test (adapters/__init__.py:386-398) ---------------------
// size >= min_overlap, errors / size <= max_error_rate (as a per-size table built with the double
// division), match_probability(matches, size) <= max_rmp; the literal shortcut (:351-367) returns a
// full-length zero-error occurrence without any test.
ATR_DEV bool linked_accept(int size, int matches, int errors, int m, int min_overlap, const int16_t *pf_thr,
                           bool accept_full, const double *rmp, int rmp_ld, double max_rmp) {
    if (accept_full && matches == m && errors == 0 && size == m) return true;
    if (size < min_overlap || size < 1 || errors > (int)pf_thr[atr_min(size, m)]) return false;
    if (rmp) return rmp[(size_t)atr_min(size, rmp_ld - 1) * rmp_ld + atr_min(atr_max(matches, 0), rmp_ld - 1)] <= max_rmp;
    return true;
}

ATR_DEV void rec_none(uint32_t rec[4]) { rec[0] = 0xFFFF0000u; rec[1] = rec[2] = rec[3] = 0u; }
ATR_DEV bool rec_found(const uint32_t rec[4]) { return (rec[0] >> 31) == 0u; }

// Acceptance test + re-basing of a 3' record that was computed in whole-read coordinates.
ATR_DEV void linked_finish(uint32_t rec[4], int s, int m, int min_overlap, const int16_t *pf_thr, bool accept_full,
                           const double *rmp, int rmp_ld, double max_rmp) {
    if (!rec_found(rec)) { rec_none(rec); return; }
    const int refstart = (int)(rec[0] & 0xFFFFu), refstop = (int)(rec[0] >> 16);
    const int matches = (int)(rec[2] & 0xFFFFu), errors = (int)(rec[2] >> 16);
    if (!linked_accept(refstop - refstart, matches, errors, m, min_overlap, pf_thr, accept_full, rmp, rmp_ld, max_rmp)) {
        rec_none(rec);
        return;
    }
    const uint32_t qs = (rec[1] & 0xFFFFu) - (uint32_t)s, qe = (rec[1] >> 16) - (uint32_t)s;
    rec[1] = (qs & 0xFFFFu) | (qe << 16);
}

// ---- front stage ---------------------------------------------------------------------------
// Literal occurrence at the read start (str.startswith on the upper-cased read, :353-355): the
// packed codes of the first m bases equal the adapter's.  Bases past the read end are code 0 and
// equal no adapter base, so a read shorter than m never passes.
ATR_DEV bool front_exact(const uint32_t *code, const uint32_t *code_mask, const uint32_t w[4]) {
    uint32_t x = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) x |= (w[d] ^ code[d]) & code_mask[d];
    return x == 0u;
}

// Exact-piece filter (Navarro / Baeza-Yates partitioning): the alignment of an anchored 5' adapter
// starts at (0, 0) and may hold at most k edit operations (every mismatch, insertion or deletion
// costs at least 1); cut into k + 1 pieces, one piece is free of them and therefore occurs
// verbatim in the read, displaced from its adapter offset by the insertions minus the deletions
// before it -- at most k either way.  Necessary for ANY acceptable alignment, whatever the indel
// cost; what passes goes to the exact DP.  t = read offset (0-based) of the eight-base window.
ATR_DEV uint32_t chunk_window(const uint32_t w[4], int t) {                  // t wave-uniform, 0 .. 31
    const uint32_t sh = 4u * (uint32_t)(t & 7);
    uint32_t lo, hi;
    switch (t >> 3) {
        case 0: lo = w[0]; hi = w[1]; break;
        case 1: lo = w[1]; hi = w[2]; break;
        case 2: lo = w[2]; hi = w[3]; break;
        default: lo = w[3]; hi = 0u; break;
    }
#ifdef ATR_HOST_EMU
    return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
#else
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#endif
}
template <bool AND_MODE>
ATR_DEV bool front_pex_candidate(const uint32_t *pex_code, const uint32_t *pex_mask, const int32_t *pex_off, int npieces,
                                 int k, const uint32_t w[4]) {
    uint32_t best = ~0u;                                                     // 0 <=> some piece occurs
    for (int p = 0; p < npieces; ++p) {
        const uint32_t code = pex_code[p], mask = pex_mask[p];
        const int off = pex_off[p];
        for (int d = -k; d <= k; ++d) {
            const int t = off + d;
            if (t < 0 || t > 31) continue;
            const uint32_t win = chunk_window(w, t);
            // literal: every base equal; wildcards: every position shares a bit
            const uint32_t bad = AND_MODE ? (~nibble_any(win & code)) & (mask & 0x88888888u) : (win ^ code) & mask;
            best = atr_minu(best, bad);
        }
    }
    return best == 0u;
}

// The same test for all adapters of a set whose pieces sit at the same offsets (equal m and k): a window of the read
// is cut once and compared with every adapter's piece.  ok[a] = some piece of adapter a occurs.
template <bool AND_MODE>
ATR_DEV void front_pex_candidates_shared(const FrontParams *f, int nad, const uint32_t w[4], bool ok[LINKED_MAX]) {
    uint32_t best[LINKED_MAX];
#pragma unroll
    for (int a = 0; a < LINKED_MAX; ++a) best[a] = ~0u;
    const int npieces = f[0].npieces, k = f[0].k;
    for (int p = 0; p < npieces; ++p) {
        const int off = f[0].pex_off[p];
        for (int d = -k; d <= k; ++d) {
            const int t = off + d;
            if (t < 0 || t > 31) continue;
            const uint32_t win = chunk_window(w, t);
#pragma unroll
            for (int a = 0; a < LINKED_MAX; ++a) {
                if (a >= nad) continue;
                const uint32_t code = f[a].pex_code[p], mask = f[a].pex_mask[p];
                const uint32_t bad = AND_MODE ? (~nibble_any(win & code)) & (mask & 0x88888888u) : (win ^ code) & mask;
                best[a] = atr_minu(best[a], bad);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < LINKED_MAX; ++a) ok[a] = a < nad && best[a] == 0u;
}

// Result of the 5' stage of one read in one word (anchored: refstart = querystart = 0, refstop = m):
//   [31:24] adapter index   [23:16] querystop   [15:8] matches   [7:0] errors;  all ones: no match.
// The smallest word belongs to the first matching adapter.
constexpr uint32_t FRONT_NONE = ~0u;
ATR_DEV uint32_t front_word(int a, int querystop, int matches, int errors) {
    return ((uint32_t)a << 24) | ((uint32_t)querystop << 16) | ((uint32_t)matches << 8) | (uint32_t)errors;
}
ATR_DEV uint32_t front_word_of(int a, const uint32_t rec[4]) {
    return front_word(a, (int)(rec[1] >> 16), (int)(rec[2] & 0xFFFFu), (int)(rec[2] >> 16));
}
ATR_DEV void front_word_record(uint32_t word, int m, uint32_t rec[4]) {
    if (word == FRONT_NONE) { rec_none(rec); return; }
    rec[0] = (uint32_t)m << 16;
    rec[1] = ((word >> 16) & 0xFFu) << 16;
    rec[2] = ((word >> 8) & 0xFFu) | ((word & 0xFFu) << 16);
    rec[3] = 0u;
}

// The read's first bases staged for the banded DP (band_stage with dlo = -k, from the first chunk
// only: m + k <= 32): stream dword z holds bases 1 - k + 8z .. ; ns stride nss, FRONT_STREAM entries.
ATR_DEV void front_stage(const uint32_t w[4], int k, uint32_t *ns, int nss) {
    // bases -k+1 .. : an eight-base window that starts k bases before the read
    const uint32_t ext[6] = {0u, w[0], w[1], w[2], w[3], 0u};
    const uint32_t sh = 4u * (uint32_t)((8 - k) & 7);
    const int first = k > 0 ? 0 : 1;                                         // k = 0: the stream starts at base 1
#pragma unroll
    for (int z = 0; z < FRONT_STREAM; ++z) {
        uint32_t v = 0u;
        if (z + first + 1 < 6) {
            const uint32_t lo = ext[z + first], hi = ext[z + first + 1];
            v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        }
        ns[(size_t)z * nss] = v;
    }
}

// wave-uniform parameters of a front adapter's DP group
ATR_DEV Uniform front_uniform(int m, int k, int indel, int min_overlap) {
    Uniform u;
    u.m = m; u.k = k; u.p0 = 0; u.first_p = m; u.indel = indel; u.min_overlap = min_overlap;
    u.sr = false; u.sq = false; u.er = false; u.eq = true;                 // flags 8: STOP_WITHIN_SEQ2 only
    u.insw = (uint32_t)indel * COST1 + PRIO_INS;
    u.delw = (uint32_t)indel * COST1 + PRIO_DEL;
    u.klimit = (uint32_t)(k + 1) << CSH;
    return u;
}

// A front record passes Adapter.match_to's test?
ATR_DEV bool front_accept(const uint32_t rec[4], int m, int min_overlap, const int16_t *pf_thr, bool accept_full,
                          const double *rmp, int rmp_ld, double max_rmp) {
    if (!rec_found(rec)) return false;
    const int refstart = (int)(rec[0] & 0xFFFFu), refstop = (int)(rec[0] >> 16);
    return linked_accept(refstop - refstart, (int)(rec[2] & 0xFFFFu), (int)(rec[2] >> 16), m, min_overlap, pf_thr,
                         accept_full, rmp, rmp_ld, max_rmp);
}
ATR_DEV void front_exact_record(uint32_t rec[4], int m) {                   // Match(0, m, 0, m, m, 0)  (:355)
    rec[0] = (uint32_t)m << 16; rec[1] = (uint32_t)m << 16; rec[2] = (uint32_t)m; rec[3] = 0u;
}

// ---- back stage ----------------------------------------------------------------------------
// Bases before s read as code 0 ("match nothing"): dword z (bases 8z .. 8z + 7, 0-based) of a read
// whose alignment starts at base s.
ATR_DEV uint32_t start_mask(int z, int s) {
    const int lo = s - 8 * z;                                                // nibbles of this dword before s
    return lo <= 0 ? ~0u : lo >= 8 ? 0u : ~0u << (4 * lo);
}

// scatter bin of the fused pipeline: FILTER_BINS bins per adapter, adapter-major
ATR_DEV int linked_bin(uint32_t ww, int which, int m, bool by_rows) { return which * FILTER_BINS + window_bin(ww, m, by_rows); }

}  // namespace atr
#endif
