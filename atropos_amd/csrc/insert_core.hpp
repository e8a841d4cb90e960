// insert_core.hpp -- per-lane arithmetic of the batched InsertAligner.match_insert
// kernel (reference: atropos/align/__init__.py:250-377 on top of
// MultiAligner.locate, atropos/align/_align.pyx:593-783).
//
// Compiled twice like locate_core.hpp: by hipcc for gfx950 and by g++
// (-DATR_HOST_EMU) for the CPU lock-step emulation in tests/emu.
//
// What the reference does per pair, and how it maps here
// -------------------------------------------------------
//  1. truncate both reads to L = min(len1, len2); rc2 = reverse_complement(read2);
//  2. MultiAligner(frac, START_WITHIN_SEQ1|STOP_WITHIN_SEQ2, min_overlap).locate(rc2, read1):
//     a no-indel DP.  With these flags and m == n == L the cell (m, j) is simply
//         cost_j = Hamming(read1[0:j], rc2[L-j:L])
//     and a hit (L-j, L, 0, j, j-cost_j, cost_j) is emitted, in increasing j, iff
//         cost_j <= int(frac*L), j >= min_overlap, cost_j <= floor(j*frac);
//     a perfect full overlap (j == L, cost 0) makes the result that single hit; the
//     scan stops after 100 hits; the full-length hit is emitted twice (harmless:
//     same tuple, same probability, the first copy wins every tie);
//  3. keep hits with rmp(matches, size=j) <= insert_max_rmp, try them in ascending
//     probability (stable) until _match() succeeds: equivalently, of the hits whose
//     _match() succeeds take the one with the smallest probability, earliest j on ties;
//  4. _match(): if the overhang L-j is shorter than min_adapter_overlap, succeed with
//     no adapter matches; else Hamming-compare both overhangs with their adapters
//     (compare_prefixes, read as "ref"), reject if BOTH exceed round(alen*frac)
//     mismatches or if the product of the two adapter RMPs exceeds adapter_max_rmp
//     (only when min(alen1, alen2) > adapter_check_cutoff).
//
// On the device both reads are packed as BIT PLANES of their 4-bit DNA15 codes (plane64:
// word p of a 16-byte chunk, bit b = bit p of the code of base 32c + b; atr_pack_planes).
// Complement == bit reversal of the nibble == plane p <-> plane 3-p, reverse == bit reversal
// of the multiword plane, so with R = bitrev(complement planes of read 2) over 32*W bits
//     Rv_j = R >> (32*W - j)   holds rc2[L-j:L] at bits 0..j-1, for every L >= j,
// and  cost_j = popcount( OR_p (A_p xor Rv_j,p) & lowmask(j) ),  32 bases per boolean op.
// The shift 32*W - j = 32*q + s is split into a bit shift s (run-time loop: 4*W
// v_alignbit_b32 per s) and a word shift q (compile-time unrolled: static register indices),
// so a pair costs about 32 * (4W + 6 * W(W+1)/2) VALU ops instead of one shift and one
// nibble-compare of every live dword for every j.  The offsets are therefore NOT visited in
// increasing j; the reference's order-dependent rules are restated order-free (the hit with
// the smallest probability, earliest j on ties; a perfect full overlap is the only hit) and
// the one rule that cannot be -- "stop after 100 hits" -- is handled by re-running the rare
// pair with more than 100 hits in increasing j (ORDERED sweep below).
#ifndef ATR_INSERT_CORE_HPP
#define ATR_INSERT_CORE_HPP

#include <stdint.h>
#include "atropos_hip.h"

#ifdef ATR_HOST_EMU
#ifndef ATR_DEV
#define ATR_DEV static inline
#endif
#ifndef ATR_DEV_MEMBER
#define ATR_DEV_MEMBER inline
#endif
static inline uint32_t atr_bfrev(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
static inline int atr_popc(uint32_t v) { return __builtin_popcount(v); }
static inline int atr_imin(int a, int b) { return a < b ? a : b; }
static inline int atr_imax(int a, int b) { return a > b ? a : b; }
static inline uint32_t atr_funnel(uint32_t hi, uint32_t lo, int s) { return s ? ((lo >> s) | (hi << (32 - s))) : lo; }   // s in 0..31
static inline uint32_t atr_or_xor(uint32_t x, uint32_t a, uint32_t b) { return x | (a ^ b); }
#else
#ifndef ATR_DEV
#define ATR_DEV __device__ __forceinline__
#endif
#ifndef ATR_DEV_MEMBER
#define ATR_DEV_MEMBER __device__ __forceinline__
#endif
#define atr_bfrev __brev
#define atr_popc __popc
#define atr_imin min
#define atr_imax max
// ({hi, lo} >> s)[31:0], s in 0..31: ONE v_alignbit_b32 (s == 0 yields lo; written as C the
// shift-by-32 case costs a compare and a select per word)
static __device__ __forceinline__ uint32_t atr_funnel(uint32_t hi, uint32_t lo, int s) {
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);
}
// x | (a ^ b) as one v_bitop3_b32 (truth table 0xF0 | (0xCC ^ 0xAA))
static __device__ __forceinline__ uint32_t atr_or_xor(uint32_t x, uint32_t a, uint32_t b) {
    return __builtin_amdgcn_bitop3_b32(x, a, b, 0xF6);
}
#endif

namespace atr {

constexpr int INS_MAX_ADAPTER = 128;                // adapter length handled by the insert kernel (two 64-base halves)
constexpr int INS_MAX_LEN = 320;                    // read length handled by the insert kernel
constexpr int INS_MAX_MATCHES = 100;                // MultiAligner.locate(max_matches=100)
constexpr int INS_CAND = 4;                         // recorded hits per pair in the unordered sweep (more: ordered redo)

// adapter compare modes (compare_prefixes(read_overhang, adapter, wildcard_ref=adapter_wildcards,
// wildcard_query=read_wildcards), align/__init__.py:285-288)
enum { INS_CMP_EQ = 0,        // neither flag: byte equality
       INS_CMP_AND = 1,       // read codes as they are (IUPAC), adapter translated on the host
       INS_CMP_AND_READ_ACGT = 2 };   // read_wildcards only: read bases that are not A/C/G/T become 0

struct InsertParams {
    uint32_t a1[4][4], a2[4][4];                    // adapter code PLANES, 128 bases (table chosen by the host per mode)
    int16_t mm_by_alen[INS_MAX_ADAPTER + 1];        // round(alen * max_adapter_mismatch_frac)
    int16_t thr_ins[INS_MAX_LEN + 1];               // floor(j * max_insert_mismatch_frac)
    int16_t k_by_len[INS_MAX_LEN + 1];              // int(max_insert_mismatch_frac * L)
    int32_t thr_hit[INS_MAX_LEN + 1];               // thr_ins[j] for j >= min_insert_overlap, else -1; 32-bit so that the
                                                    // sweep fetches its (wave-uniform) entry with a scalar load
    int alen1, alen2, cmp_mode;
    int long_adapters;                              // an adapter has more than 64 bases: the overhangs are compared in two halves
                                                    // (the launcher then picks the kernels built with InsertParamsLong)
    static constexpr bool kLongAdapters = false;
    static constexpr bool kCasedReads = false;
    int min_insert_overlap, min_adapter_overlap, adapter_check_cutoff;
    int min_hit_j;                                  // smallest j whose PERFECT overlap passes insert_max_rmp (host)
    int rmp_ld;
    double insert_max_rmp, adapter_max_rmp;
    const double *rmp_insert, *rmp_adapter;         // device pointers, [size][matches], ld = rmp_ld
};

// Same bytes; the type tells the compare code at compile time to look at the second 64 adapter bases too (the
// extra path costs the common kernels registers they do not have: C3 1.04 -> 1.27 ms when it was a run-time test).
struct InsertParamsLong : InsertParams { static constexpr bool kLongAdapters = true; };
// Soft-masked reads: both reads are packed with the CASE-SENSITIVE table (lower-case a/t, c/g, n on the codes 3/12,
// 5/10, 6 -- closed under the complement, distinct from their upper-case letters, as the insert compare needs:
// MultiAligner compares characters, _align.pyx:690), and the adapter compares that translate the read
// (compare_prefixes with a wildcard flag folds the case, _align.pyx:31-86) fold those codes back first.
struct InsertParamsCased : InsertParams { static constexpr bool kCasedReads = true; };
struct InsertParamsLongCased : InsertParams { static constexpr bool kLongAdapters = true; static constexpr bool kCasedReads = true; };

ATR_DEV uint32_t low_mask(int nb) {                 // nb low bits set, nb in (-inf, 32]
    return nb >= 32 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
}

// word `idx` of a W-word plane, 0 beyond it (idx is a compile-time constant after unrolling)
template <int W>
ATR_DEV uint32_t word_or_zero(const uint32_t (&v)[W], int idx) { return (idx >= 0 && idx < W) ? v[(idx >= 0 && idx < W) ? idx : 0] : 0u; }

template <int W>
struct PairState {
    uint32_t a[4][W];                                // read 1, code planes
    uint32_t r[4][W];                                // bit-reversed complement planes of read 2 (R above)
    int L, len1, len2, k;
    int nhits;
    double best_prob;
    int best_j, best_cost, best_e1, best_e2;        // best_e1 < 0: no adapter matches attached
    bool has_best;
    // unordered sweep: hits that can pass the insert RMP filter are only recorded during the
    // sweep (j << 16 | cost) and evaluated afterwards, all lanes together
    // (four scalars, not an array: `if (ncand == c) cand[c] = v` over an array is rewritten by the compiler into
    // cand[ncand] = v, a dynamically indexed store that moves the WHOLE PairState into scratch memory -- 176 bytes
    // written per pair before the sweep starts, the "spill" traffic the round-2 counters showed)
    uint32_t cand0, cand1, cand2, cand3;
    int ncand;
};
static_assert(INS_CAND == 4, "PairState holds four recorded hits");
template <int W>
ATR_DEV uint32_t cand_get(const PairState<W> &P, int c) { return c == 0 ? P.cand0 : c == 1 ? P.cand1 : c == 2 ? P.cand2 : P.cand3; }

// b1, b2: the plane64 chunks of the two reads (word p of chunk c at [4*c + p]).
template <int W, class IP>
ATR_DEV void pair_init(PairState<W> &P, const IP &ip, int len1, int len2, const uint32_t *b1, const uint32_t *b2) {
    P.len1 = len1; P.len2 = len2;
    P.L = atr_imin(len1, len2);                                       // align/__init__.py:259-265
    P.k = ip.k_by_len[atr_imin(P.L, INS_MAX_LEN)];
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            P.a[p][w] = b1[4 * w + p];
            P.r[p][w] = atr_bfrev(b2[4 * (W - 1 - w) + (3 - p)]);     // complement: plane 3-p; reverse: bit reversal
        }
}

template <int W>
ATR_DEV void pair_reset(PairState<W> &P) {
    P.nhits = 0; P.has_best = false; P.best_prob = 0.0;
    P.best_j = P.best_cost = 0; P.best_e1 = P.best_e2 = -1;
    P.ncand = 0;
    P.cand0 = P.cand1 = P.cand2 = P.cand3 = 0u;
}

// Case-sensitive read codes -> the codes of their upper-case letters (3 -> 1, 12 -> 8, 5 -> 2, 10 -> 4, 6 -> 15),
// 32 bases per word on the four planes.
ATR_DEV void fold_case_planes(uint32_t &b0, uint32_t &b1, uint32_t &b2, uint32_t &b3) {
    const uint32_t la = ~b3 & ~b2 & b1 & b0, lt = b3 & b2 & ~b1 & ~b0, lc = ~b3 & b2 & ~b1 & b0;
    const uint32_t lg = b3 & ~b2 & b1 & ~b0, ln = ~b3 & b2 & b1 & ~b0;
    const uint32_t o0 = (b0 & ~lc) | ln;
    const uint32_t o1 = (b1 & ~la & ~lg) | lc | ln;
    const uint32_t o2 = (b2 & ~lt & ~lc) | lg | ln;
    const uint32_t o3 = (b3 & ~lg) | ln;
    b0 = o0; b1 = o1; b2 = o2; b3 = o3;
}

// Mismatches of 64 bases of read overhang (planes ov[p][0..1]) against the bases 64 * half .. of an adapter
// (alen = the adapter bases compared in all; the second half only exists for adapters of more than 64 bases).
template <bool CASED = false>
ATR_DEV int overhang_mismatches(const uint32_t (&ov)[4][2], const uint32_t (&adp)[4][4], int alen, int mode, int half = 0) {
    int mism = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        uint32_t x0 = ov[0][t], x1 = ov[1][t], x2 = ov[2][t], x3 = ov[3][t];
        if (CASED && mode != INS_CMP_EQ) fold_case_planes(x0, x1, x2, x3);        // the translate tables fold the case
        const uint32_t ad[4][1] = {{half ? adp[0][2 + t] : adp[0][t]}, {half ? adp[1][2 + t] : adp[1][t]},
                                   {half ? adp[2][2 + t] : adp[2][t]}, {half ? adp[3][2 + t] : adp[3][t]}};
        uint32_t bad;
        if (mode == INS_CMP_EQ) {
            bad = (x0 ^ ad[0][0]) | (x1 ^ ad[1][0]) | (x2 ^ ad[2][0]) | (x3 ^ ad[3][0]);
        } else {
            if (mode == INS_CMP_AND_READ_ACGT) {
                // keep only A/C/G/T codes (exactly one plane bit), zero everything else: the
                // _acgt_table view of an IUPAC-coded read
                const uint32_t one = ((x0 ^ x1) ^ (x2 ^ x3)) & ~((x0 & x1) | (x2 & x3));
                x0 &= one; x1 &= one; x2 &= one; x3 &= one;
            }
            bad = ~((x0 & ad[0][0]) | (x1 & ad[1][0]) | (x2 & ad[2][0]) | (x3 & ad[3][0]));
        }
        mism += atr_popc(bad & low_mask(alen - 64 * half - 32 * t));
    }
    return mism;
}

// The 64 bases from base j = 32*WI + sh on of a W-word plane set (WI compile time, sh run time).
template <int W, int WI>
ATR_DEV void planes_from(const uint32_t (&v)[4][W], int sh, uint32_t (&ov)[4][2]) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t lo = word_or_zero<W>(v[p], WI + t), hi = word_or_zero<W>(v[p], WI + t + 1);
            ov[p][t] = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        }
}

// Everything that happens to a hit of MultiAligner.locate (overlap j, `cost` mismatches):
// insert RMP filter and InsertAligner._match.  WI = j >> 5 as a compile-time constant.
// Returns whether the hit survives; prob, e1, e2 describe it.
template <int W, int WI, class IP>
ATR_DEV bool evaluate_hit(const PairState<W> &P, const IP &ip, int j, int cost, double &prob, int &e1, int &e2) {
    const int matches = j - cost;
    prob = ip.rmp_insert[(size_t)j * ip.rmp_ld + matches];            // align/__init__.py:359
    e1 = e2 = -1;
    if (!(prob <= ip.insert_max_rmp)) return false;
    const int offset = P.L - j;
    if (offset >= ip.min_adapter_overlap) {                           // align/__init__.py:270-276
        const int al1 = atr_imin(offset, ip.alen1), al2 = atr_imin(offset, ip.alen2);
        uint32_t ov[4][2];
        planes_from<W, WI>(P.a, j & 31, ov);
        e1 = overhang_mismatches<IP::kCasedReads>(ov, ip.a1, al1, ip.cmp_mode);
        if (IP::kLongAdapters) {                                      // compile time: adapters of more than 64 bases
            planes_from<W, WI + 2>(P.a, j & 31, ov);
            e1 += overhang_mismatches<IP::kCasedReads>(ov, ip.a1, al1, ip.cmp_mode, 1);
        }
        // read 2 in natural order: plane p word w = bitrev(R[3-p][W-1-w])
        uint32_t b2[4][W];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int w = 0; w < W; ++w) b2[p][w] = atr_bfrev(P.r[3 - p][W - 1 - w]);
        planes_from<W, WI>(b2, j & 31, ov);
        e2 = overhang_mismatches<IP::kCasedReads>(ov, ip.a2, al2, ip.cmp_mode);
        if (IP::kLongAdapters) {
            planes_from<W, WI + 2>(b2, j & 31, ov);
            e2 += overhang_mismatches<IP::kCasedReads>(ov, ip.a2, al2, ip.cmp_mode, 1);
        }
        if (e1 > (int)ip.mm_by_alen[al1] && e2 > (int)ip.mm_by_alen[al2]) return false;   // :297-300
        if (atr_imin(al1, al2) > ip.adapter_check_cutoff) {                               // :302-306
            const double p1 = ip.rmp_adapter[(size_t)al1 * ip.rmp_ld + (al1 - e1)];
            const double p2 = ip.rmp_adapter[(size_t)al2 * ip.rmp_ld + (al2 - e2)];
            if (p1 * p2 > ip.adapter_max_rmp) return false;
        }
    }
    return true;
}

// Overlap length j = 32*(W - Q) - s with `cost` mismatches, for one lane, in the reference's
// order (increasing j).
template <int W, int Q, class IP>
ATR_DEV void pair_hit_ordered(PairState<W> &P, const IP &ip, int s, int j, int cost) {
    if (j > P.L || P.nhits >= INS_MAX_MATCHES) return;
    // the hit test of MultiAligner.locate (_align.pyx:713-745)
    if (cost > P.k || j < ip.min_insert_overlap || cost > (int)ip.thr_ins[j]) return;
    P.nhits += 1;
    if (cost == 0 && j == P.L) P.has_best = false;      // exact full overlap: the only hit (:737-741, :767-768)
    double prob;
    int e1, e2;
    bool ok;
    if (s == 0) ok = evaluate_hit<W, W - Q>(P, ip, j, cost, prob, e1, e2);      // j >> 5 == W - Q
    else ok = evaluate_hit<W, W - Q - 1>(P, ip, j, cost, prob, e1, e2);         // j >> 5 == W - Q - 1
    if (!ok) return;
    if (P.has_best && !(prob < P.best_prob)) return;            // stable ascending-probability order
    P.has_best = true; P.best_prob = prob; P.best_j = j; P.best_cost = cost; P.best_e1 = e1; P.best_e2 = e2;
}

// The same hit in the unordered sweep: count it; if its overlap is long enough to pass the
// insert RMP filter at all, remember it for the evaluation pass.
template <int W, class IP>
ATR_DEV void pair_hit_record(PairState<W> &P, const IP &ip, int j, int cost, int limit) {
    if (j > P.L || cost > limit) return;                // the common exit; limit = ip.thr_hit[j], in an SGPR
    if (cost > P.k) return;
    P.nhits += 1;
    if (j < ip.min_hit_j) return;                       // rmp(matches <= j, j) >= rmp(j, j) > insert_max_rmp
    // slot ncand, written in place under the lane mask (a rotation of the slots would cost register
    // copies on every offset, hit or not); the evaluation pass does not depend on the slot order
    const uint32_t v = ((uint32_t)j << 16) | (uint32_t)cost;
    const int at = P.ncand;
    P.cand0 = at == 0 ? v : P.cand0;
    P.cand1 = at == 1 ? v : P.cand1;
    P.cand2 = at == 2 ? v : P.cand2;
    P.cand3 = at == 3 ? v : P.cand3;
    P.ncand += 1;
}

// 64 bases of a plane64-packed read from base j on, fetched from memory (per-lane j):
// g = the lane's chunk 0, chunk stride `cstride` dwords, nchunks chunks.
ATR_DEV void planes_from_memory(const uint32_t *g, int cstride, int nchunks, int j, uint32_t (&ov)[4][2]) {
    const int wi = j >> 5, sh = j & 31;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t w3[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) w3[t] = (wi + t < nchunks) ? g[(size_t)(wi + t) * cstride + p] : 0u;
#pragma unroll
        for (int t = 0; t < 2; ++t) ov[p][t] = sh ? ((w3[t] >> sh) | (w3[t + 1] << (32 - sh))) : w3[t];
    }
}

// Evaluation of one recorded hit (insert RMP filter + InsertAligner._match) with a per-lane j.
template <int W, class IP>
ATR_DEV bool evaluate_candidate(const PairState<W> &P, const IP &ip, int j, int cost, const uint32_t *g1,
                                const uint32_t *g2, int cstride, double &prob, int &e1, int &e2) {
    prob = ip.rmp_insert[(size_t)j * ip.rmp_ld + (j - cost)];         // align/__init__.py:359
    e1 = e2 = -1;
    if (!(prob <= ip.insert_max_rmp)) return false;
    const int offset = P.L - j;
    if (offset >= ip.min_adapter_overlap) {                           // align/__init__.py:270-276
        const int al1 = atr_imin(offset, ip.alen1), al2 = atr_imin(offset, ip.alen2);
        uint32_t ov[4][2];
        planes_from_memory(g1, cstride, W, j, ov);
        e1 = overhang_mismatches<IP::kCasedReads>(ov, ip.a1, al1, ip.cmp_mode);
        planes_from_memory(g2, cstride, W, j, ov);
        e2 = overhang_mismatches<IP::kCasedReads>(ov, ip.a2, al2, ip.cmp_mode);
        if (IP::kLongAdapters) {                                      // compile time: adapters of more than 64 bases
            planes_from_memory(g1, cstride, W, j + 64, ov);
            e1 += overhang_mismatches<IP::kCasedReads>(ov, ip.a1, al1, ip.cmp_mode, 1);
            planes_from_memory(g2, cstride, W, j + 64, ov);
            e2 += overhang_mismatches<IP::kCasedReads>(ov, ip.a2, al2, ip.cmp_mode, 1);
        }
        if (e1 > (int)ip.mm_by_alen[al1] && e2 > (int)ip.mm_by_alen[al2]) return false;   // :297-300
        if (atr_imin(al1, al2) > ip.adapter_check_cutoff) {                               // :302-306
            const double p1 = ip.rmp_adapter[(size_t)al1 * ip.rmp_ld + (al1 - e1)];
            const double p2 = ip.rmp_adapter[(size_t)al2 * ip.rmp_ld + (al2 - e2)];
            if (p1 * p2 > ip.adapter_max_rmp) return false;
        }
    }
    return true;
}

// mismatches of read-1 word w against Rs word w + Q (Rs = R >> s); the top word keeps only its
// 32 - s valid bits.  Four ops per 32 bases.
template <int W, int Q>
ATR_DEV int word_cost(const PairState<W> &P, const uint32_t (&rs)[4][W], uint32_t topmask, int w) {
    uint32_t m = P.a[0][w] ^ rs[0][w + Q];
    m = atr_or_xor(m, P.a[1][w], rs[1][w + Q]);
    m = atr_or_xor(m, P.a[2][w], rs[2][w + Q]);
    m = atr_or_xor(m, P.a[3][w], rs[3][w + Q]);
    if (w + Q == W - 1) m &= topmask;
    return atr_popc(m);
}

// cost of overlap j = 32*(W-Q) - s: read-1 words 0 .. W-1-Q against Rs words Q .. W-1.
template <int W, int Q>
ATR_DEV int overlap_cost(const PairState<W> &P, const uint32_t (&rs)[4][W], uint32_t topmask) {
    int cost = 0;
#pragma unroll
    for (int w = 0; w + Q < W; ++w) cost += word_cost<W, Q>(P, rs, topmask, w);
    return cost;
}

// The same for the unordered sweep, which only needs the cost of the offsets that can be hits
// (cost <= limit, the wave-uniform threshold of this overlap length): two words of a random
// overlap already hold ~48 mismatches, so the remaining words are compared only if some lane of
// the wave is still below the limit after two.  A lane that returns early returns a partial
// cost above the limit, which the hit test rejects like the full one.
constexpr int INS_PROBE_WORDS = 2;
template <int W, int Q>
ATR_DEV int overlap_cost_limited(const PairState<W> &P, const uint32_t (&rs)[4][W], uint32_t topmask, int limit) {
    int cost = 0;
#pragma unroll
    for (int w = 0; w + Q < W && w < INS_PROBE_WORDS; ++w) cost += word_cost<W, Q>(P, rs, topmask, w);
    if constexpr (W - Q > INS_PROBE_WORDS) {
#ifdef ATR_HOST_EMU
        if (cost > limit) return cost;
#else
        if (!__any(cost <= limit)) return cost;                     // wave-uniform
#endif
#pragma unroll
        for (int w = INS_PROBE_WORDS; w + Q < W; ++w) cost += word_cost<W, Q>(P, rs, topmask, w);
    }
    return cost;
}

template <int W, int Q, bool ORDERED>
struct WordShift {
    // unordered sweep: one bit shift s, all word shifts
    template <class IP>
    static ATR_DEV_MEMBER void all(PairState<W> &P, const IP &ip, int s, const uint32_t (&rs)[4][W],
                                   uint32_t topmask, int jmax) {
        const int j = 32 * (W - Q) - s;
        if (j >= 1 && j <= jmax) {
            const int limit = ip.thr_hit[atr_imin(j, INS_MAX_LEN)];       // wave-uniform: a scalar load
            pair_hit_record<W>(P, ip, j, overlap_cost_limited<W, Q>(P, rs, topmask, limit), limit);
        }
        WordShift<W, Q + 1, ORDERED>::all(P, ip, s, rs, topmask, jmax);
    }
    // ordered sweep: word shifts from W-1 down (j ascending), all bit shifts from 31 down inside
    template <class IP>
    static ATR_DEV_MEMBER void descending(PairState<W> &P, const IP &ip, int jmax) {
        if (32 * (W - Q) - 31 <= jmax) {                              // wave-uniform
#ifndef ATR_HOST_EMU
#pragma unroll 1
#endif
            for (int s = 31; s >= 0; --s) {
                const int j = 32 * (W - Q) - s;
                if (j > jmax) break;
                uint32_t rs[4][W];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const uint32_t lo = P.r[p][w], hi = (w + 1 < W) ? P.r[p][(w + 1 < W) ? w + 1 : 0] : 0u;
                        rs[p][w] = (w >= Q) ? atr_funnel(hi, lo, s) : 0u;
                    }
                pair_hit_ordered<W, Q>(P, ip, s, j, overlap_cost<W, Q>(P, rs, 0xFFFFFFFFu >> s));
            }
        }
        if constexpr (Q > 0) WordShift<W, Q - 1, ORDERED>::descending(P, ip, jmax);
    }
};
template <int W, bool ORDERED>
struct WordShift<W, W, ORDERED> {
    static ATR_DEV_MEMBER void all(PairState<W> &, const InsertParams &, int, const uint32_t (&)[4][W], uint32_t, int) {}
    static ATR_DEV_MEMBER void descending(PairState<W> &, const InsertParams &, int) {}
};

// The unordered sweep of one lane; jmax = the wave-uniform upper bound of the overlap length.
// g1, g2: the lane's plane64 chunks in memory (chunk stride cstride dwords), for the evaluation pass.
template <int W, class IP>
ATR_DEV void sweep_unordered(PairState<W> &P, const IP &ip, int jmax, const uint32_t *g1, const uint32_t *g2,
                             int cstride) {
    pair_reset<W>(P);
#ifndef ATR_HOST_EMU
#pragma unroll 1
#endif
    for (int s = 0; s < 32; ++s) {
        if (32 * W - s < 1) break;
        uint32_t rs[4][W];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t lo = P.r[p][w], hi = (w + 1 < W) ? P.r[p][(w + 1 < W) ? w + 1 : 0] : 0u;
                rs[p][w] = atr_funnel(hi, lo, s);
            }
        WordShift<W, 0, false>::all(P, ip, s, rs, 0xFFFFFFFFu >> s, jmax);
    }
    // evaluation pass: every lane looks at its c-th recorded hit.  Order-free restatement of
    // "try the hits in ascending probability (stable)": smallest probability, earliest j on ties;
    // a perfect full overlap is the only hit (:737-741).
    bool exact = false;
    for (int c = 0; c < INS_CAND; ++c) {
        if (c >= P.ncand) continue;
        const uint32_t cv = cand_get<W>(P, c);
        const int j = (int)(cv >> 16), cost = (int)(cv & 0xFFFFu);
        double prob;
        int e1, e2;
        const bool ok = evaluate_candidate<W>(P, ip, j, cost, g1, g2, cstride, prob, e1, e2);
        const bool full = (cost == 0 && j == P.L);
        if (exact && !full) continue;
        if (full) { exact = true; P.has_best = false; }
        if (!ok) continue;
        if (!full && P.has_best && !(prob < P.best_prob || (prob == P.best_prob && j < P.best_j))) continue;
        P.has_best = true; P.best_prob = prob; P.best_j = j; P.best_cost = cost; P.best_e1 = e1; P.best_e2 = e2;
    }
}

// ---- probed sweep: the unordered sweep in two passes ------------------------------------------------------
// The unordered sweep above compares all four planes of every (bit shift, word shift) step as soon as ONE lane of
// the wave is still below its limit after two words -- and with 64 pairs per wave some lane's true overlap sits in
// 40 % of the steps.  Here pass 1 only computes a LOWER BOUND of every overlap's cost -- the positions where planes
// 0 and 1 differ (two ops per 32 bases instead of four; 5/8 of random bases differ there, 3/4 in all four planes)
// over the first two to four words (ins_probe_words) -- and notes the overlap lengths the bound cannot reject in a per-lane list
// in LDS; pass 2 computes the exact cost of the listed lengths, each lane its own (the planes of read 2 sit in
// LDS, indexed per lane), and hands them to the same pair_hit_record.  A true hit always passes the bound, so the
// hits -- and with them everything after -- are those of the unordered sweep.  Overlaps of at most 32 bases (one
// word) are costed exactly in pass 1: the many chance hits of a few bases need no second look.
// rl: the lane's R planes in LDS, rl[(p * W + w) * rls] (no padding word: the funnel's high word above the last one is
// selected to 0 -- 1 KB per wave, what stood between 250-base pairs and a fourth block per CU); cl: its candidate
// list, cl[c * cls].
// Words the bound looks at, by the words of the overlap: 32 random bases differ in planes 0 / 1 at 20 +- 2.7
// positions, the limit of an overlap of n words is 6.4 n (frac 0.2): the bound must clear it by a few sigma, or
// every long overlap lands in the list (2 words at 2 x 250 bp: list overflow, ordered redo, 0.41 -> 1.5 ms).
constexpr int ins_probe_words(int nw) { return nw <= 3 ? 2 : nw <= 6 ? 3 : 4; }
constexpr int INS_LIST_CAP = 16;                    // listed overlap lengths per lane; more: ordered redo
// (eight-chunk reads list 12: with 8 KB of planes per wave that is the fourth block of a CU's 160 KB; five-chunk
// reads list 8: the sixth)
constexpr int ins_list_cap(int w) { return w == 8 ? 12 : w == 5 ? 8 : INS_LIST_CAP; }

template <int W>
ATR_DEV void planes_to_lds(const PairState<W> &P, uint32_t *rl, int rls) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int w = 0; w < W; ++w) rl[(size_t)(p * W + w) * rls] = P.r[p][w];
    }
}
template <int W>
ATR_DEV void planes_from_lds(PairState<W> &P, const uint32_t *rl, int rls) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int w = 0; w < W; ++w) P.r[p][w] = rl[(size_t)(p * W + w) * rls];
}

template <int W, int Q>
struct ProbeShift {
    template <class IP>
    static ATR_DEV_MEMBER void all(PairState<W> &P, const IP &ip, int s, const uint32_t (&rs0)[W], const uint32_t (&rs1)[W],
                                   uint32_t rt2, uint32_t rt3, uint32_t topmask, int jmax, uint16_t *cl, int cls, int &nlist) {
        const int j = 32 * (W - Q) - s;
        if (j >= 1 && j <= jmax) {                                        // wave-uniform
            const int limit = ip.thr_hit[atr_imin(j, INS_MAX_LEN)];       // wave-uniform: a scalar load
            if constexpr (Q == W - 1) {
                // one word: the exact cost, as the unordered sweep
                uint32_t m = P.a[0][0] ^ rs0[W - 1];
                m = atr_or_xor(m, P.a[1][0], rs1[W - 1]);
                m = atr_or_xor(m, P.a[2][0], rt2);
                m = atr_or_xor(m, P.a[3][0], rt3);
                pair_hit_record<W>(P, ip, j, atr_popc(m & topmask), limit);
            } else {
                int lb = 0;
#pragma unroll
                for (int w = 0; w + Q < W && w < ins_probe_words(W - Q); ++w) {
                    uint32_t m = atr_or_xor(P.a[0][w] ^ rs0[w + Q], P.a[1][w], rs1[w + Q]);
                    if (w + Q == W - 1) m &= topmask;
                    lb += atr_popc(m);
                }
                if (lb <= limit && j <= P.L) {
                    if (nlist < ins_list_cap(W)) cl[(size_t)nlist * cls] = (uint16_t)j;
                    nlist += 1;
                }
            }
        }
        if constexpr (Q + 1 < W) ProbeShift<W, Q + 1>::all(P, ip, s, rs0, rs1, rt2, rt3, topmask, jmax, cl, cls, nlist);
    }
};

// exact cost of overlap j for one lane, read 2's planes from LDS (per-lane word shift)
template <int W>
ATR_DEV int overlap_cost_lds(const PairState<W> &P, const uint32_t *rl, int rls, int j) {
    const int sh = 32 * W - j, q = sh >> 5, s = sh & 31;
    const uint32_t topmask = 0xFFFFFFFFu >> s;
    int cost = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (w + q < W) {
            uint32_t m = 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t lo = rl[(size_t)(p * W + w + q) * rls];
                const uint32_t hi = w + q + 1 < W ? rl[(size_t)(p * W + w + q + 1) * rls] : 0u;
                m = atr_or_xor(m, P.a[p][w], atr_funnel(hi, lo, s));
            }
            if (w + q == W - 1) m &= topmask;
            cost += atr_popc(m);
        }
    }
    return cost;
}

// The positions of read 1 that DISAGREE with what faces them in an insert match of overlap j (read1[t] faces the
// complement of read2[j - 1 - t], t < j): word w of the mask overlap_cost_lds counts.  What the error correction
// visits (commands/trim/modifiers.py:272-300; misc_core.hpp facing_mismatches computes the same words from read 2's
// raw planes) -- here from the planes the match was found on, so that the fused kernel does not stream them twice.
template <int W>
ATR_DEV void insert_overlap_mismatches(const PairState<W> &P, const uint32_t *rl, int rls, int j, uint32_t (&mism)[W]) {
    const int sh = 32 * W - j, q = sh >> 5, s = sh & 31;
    const uint32_t topmask = 0xFFFFFFFFu >> s;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint32_t m = 0u;
        if (j >= 1 && w + q < W) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t lo = rl[(size_t)(p * W + w + q) * rls];
                const uint32_t hi = w + q + 1 < W ? rl[(size_t)(p * W + w + q + 1) * rls] : 0u;
                m = atr_or_xor(m, P.a[p][w], atr_funnel(hi, lo, s));
            }
            if (w + q == W - 1) m &= topmask;
        }
        mism[w] = m;
    }
}

// nmax_of(n): the wave's largest n (the kernel: a cross-lane maximum; the emulation: n itself)
template <int W, class IP, class NMAX>
ATR_DEV void sweep_probed(PairState<W> &P, const IP &ip, int jmax, const uint32_t *g1, const uint32_t *g2, int cstride,
                          const uint32_t *rl, int rls, uint16_t *cl, int cls, const int32_t *thr_hit, NMAX nmax_of) {
    pair_reset<W>(P);
    int nlist = 0;
#ifndef ATR_HOST_EMU
#pragma unroll 1
#endif
    for (int s = 0; s < 32; ++s) {
        if (32 * W - s < 1) break;
        uint32_t rs0[W], rs1[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            rs0[w] = atr_funnel((w + 1 < W) ? P.r[0][(w + 1 < W) ? w + 1 : 0] : 0u, P.r[0][w], s);
            rs1[w] = atr_funnel((w + 1 < W) ? P.r[1][(w + 1 < W) ? w + 1 : 0] : 0u, P.r[1][w], s);
        }
        const uint32_t rt2 = P.r[2][W - 1] >> s, rt3 = P.r[3][W - 1] >> s;
        ProbeShift<W, 0>::all(P, ip, s, rs0, rs1, rt2, rt3, 0xFFFFFFFFu >> s, jmax, cl, cls, nlist);
    }
    // pass 2: the exact cost of the listed overlap lengths
    const int nmax = nmax_of(atr_imin(nlist, ins_list_cap(W)));
    for (int c = 0; c < nmax; ++c) {
        if (c < nlist && c < ins_list_cap(W)) {
            const int j = (int)cl[(size_t)c * cls];
            pair_hit_record<W>(P, ip, j, overlap_cost_lds<W>(P, rl, rls, j), thr_hit[atr_imin(j, INS_MAX_LEN)]);
        }
    }
    if (nlist > ins_list_cap(W)) P.ncand = INS_CAND + 1;                     // list overflow: the ordered sweep decides
    // evaluation pass, as in sweep_unordered
    bool exact = false;
    for (int c = 0; c < INS_CAND; ++c) {
        if (c >= P.ncand) continue;
        const uint32_t cv = cand_get<W>(P, c);
        const int j = (int)(cv >> 16), cost = (int)(cv & 0xFFFFu);
        double prob;
        int e1, e2;
        const bool ok = evaluate_candidate<W>(P, ip, j, cost, g1, g2, cstride, prob, e1, e2);
        const bool full = (cost == 0 && j == P.L);
        if (exact && !full) continue;
        if (full) { exact = true; P.has_best = false; }
        if (!ok) continue;
        if (!full && P.has_best && !(prob < P.best_prob || (prob == P.best_prob && j < P.best_j))) continue;
        P.has_best = true; P.best_prob = prob; P.best_j = j; P.best_cost = cost; P.best_e1 = e1; P.best_e2 = e2;
    }
}

// Whether the unordered sweep's answer stands: at most 100 hits (the reference stops after
// 100, in increasing j) and no more recorded hits than there are slots.
template <int W>
ATR_DEV bool unordered_is_exact(const PairState<W> &P) { return P.nhits <= INS_MAX_MATCHES && P.ncand <= INS_CAND; }

// The reference's own order (increasing j, at most 100 hits): only for the rare pairs whose
// unordered sweep does not stand (low-complexity reads).
template <int W, class IP>
ATR_DEV void sweep_ordered(PairState<W> &P, const IP &ip, int jmax) {
    pair_reset<W>(P);
    WordShift<W, W - 1, true>::descending(P, ip, jmax);
}

// Three 16-byte records per pair: the insert match, Match 1, Match 2
// (refstop / astop == -1: absent).
template <int W, class IP>
ATR_DEV void pair_result(const PairState<W> &P, const IP &ip, uint32_t rec[12]) {
    int v[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) v[i] = 0;
    v[1] = v[7] = v[13] = -1;
    if (P.has_best) {
        const int j = P.best_j, offset = P.L - j;
        v[0] = offset; v[1] = P.L; v[2] = 0; v[3] = j; v[4] = j - P.best_cost; v[5] = P.best_cost;
        if (P.best_e1 >= 0) {
            const int mism = atr_imin(P.best_e1, P.best_e2);                          // :308
            const int al1 = atr_imin(atr_imin(offset, ip.alen1), P.len1 - j);         // _create_match, :310-314
            const int al2 = atr_imin(atr_imin(offset, ip.alen2), P.len2 - j);
            const int m1 = atr_imin(al1, mism), m2 = atr_imin(al2, mism);
            v[6] = 0; v[7] = al1; v[8] = j; v[9] = P.len1; v[10] = al1 - m1; v[11] = m1;
            v[12] = 0; v[13] = al2; v[14] = j; v[15] = P.len2; v[16] = al2 - m2; v[17] = m2;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        rec[4 * t + 0] = (uint32_t)(v[6 * t + 0] & 0xFFFF) | ((uint32_t)(v[6 * t + 1] & 0xFFFF) << 16);
        rec[4 * t + 1] = (uint32_t)(v[6 * t + 2] & 0xFFFF) | ((uint32_t)(v[6 * t + 3] & 0xFFFF) << 16);
        rec[4 * t + 2] = (uint32_t)(v[6 * t + 4] & 0xFFFF) | ((uint32_t)(v[6 * t + 5] & 0xFFFF) << 16);
        rec[4 * t + 3] = 0;
    }
}

}  // namespace atr
#endif
