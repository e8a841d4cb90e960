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
// On the device both reads are 4-bit packed with the DNA15 table (upper-case IUPAC
// letters -> bit codes).  Complement == bit reversal of a nibble and reverse ==
// nibble order reversal, so rc2 is v_bfrev_b32 of read2's dwords taken in reverse
// order.  For overlap length j the alignment of rc2 against read1 is a shift by
// (D*8 - j) bases regardless of L, so one shift register X (fed one base per step
// from the top of bfrev(read2)) serves every lane of the wave; mismatches are counted
// 8 bases per dword with xor / nibble-collapse / v_bcnt.  The sweep is unrolled by
// dword block B (compile time) x base r (run time), so every register index is static.
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
#endif

namespace atr {

constexpr int INS_MAX_ADAPTER = 64;                 // adapter length handled by the insert kernel
constexpr int INS_AW = INS_MAX_ADAPTER / 8;         // adapter words
constexpr int INS_MAX_LEN = 256;                    // read length handled by the insert kernel
constexpr int INS_MAX_MATCHES = 100;                // MultiAligner.locate(max_matches=100)

// adapter compare modes (compare_prefixes(read_overhang, adapter, wildcard_ref=adapter_wildcards,
// wildcard_query=read_wildcards), align/__init__.py:285-288)
enum { INS_CMP_EQ = 0,        // neither flag: byte equality
       INS_CMP_AND = 1,       // read codes as they are (IUPAC), adapter translated on the host
       INS_CMP_AND_READ_ACGT = 2 };   // read_wildcards only: read bases that are not A/C/G/T become 0

struct InsertParams {
    uint32_t a1[INS_AW], a2[INS_AW];                // adapter codes (table chosen by the host per mode)
    int16_t mm_by_alen[INS_MAX_ADAPTER + 1];        // round(alen * max_adapter_mismatch_frac)
    int16_t thr_ins[INS_MAX_LEN + 1];               // floor(j * max_insert_mismatch_frac)
    int16_t k_by_len[INS_MAX_LEN + 1];              // int(max_insert_mismatch_frac * L)
    int alen1, alen2, cmp_mode;
    int min_insert_overlap, min_adapter_overlap, adapter_check_cutoff;
    int rmp_ld;
    double insert_max_rmp, adapter_max_rmp;
    const double *rmp_insert, *rmp_adapter;         // device pointers, [size][matches], ld = rmp_ld
};

// nibble != 0  ->  bit 0 of that nibble
ATR_DEV uint32_t nibble_nonzero(uint32_t d) {
    uint32_t t = d | (d >> 2);
    t |= t >> 1;
    return t & 0x11111111u;
}

// one bit per base for the first nb (0..8) bases of a dword
ATR_DEV uint32_t base_mask(int nb) {
    return nb >= 8 ? 0x11111111u : (nb <= 0 ? 0u : (0x11111111u & ((1u << (4 * nb)) - 1u)));
}

// Keep only A/C/G/T codes (one-hot nibbles), zero everything else (the _acgt_table view
// of an IUPAC-coded read).
ATR_DEV uint32_t acgt_only(uint32_t x) {
    // per-nibble popcount == 1  <=>  x != 0 and (x & (x-1)) == 0, done without borrows:
    const uint32_t b0 = x & 0x11111111u, b1 = (x >> 1) & 0x11111111u, b2 = (x >> 2) & 0x11111111u,
                   b3 = (x >> 3) & 0x11111111u;
    const uint32_t sum2 = (b0 & b1) | (b0 & b2) | (b0 & b3) | (b1 & b2) | (b1 & b3) | (b2 & b3);  // >= 2 bits
    const uint32_t one = (b0 | b1 | b2 | b3) & ~sum2;                                             // exactly 1
    return x & (one * 15u);
}

template <int D>
struct PairState {
    uint32_t s1[D], s2[D], x[D];
    int L, len1, len2, k;
    int nhits;
    double best_prob;
    int best_j, best_cost, best_e1, best_e2;        // best_e1 < 0: no adapter matches attached
    bool has_best;
};

template <int D>
ATR_DEV void pair_init(PairState<D> &P, const InsertParams &ip, int len1, int len2) {
    P.len1 = len1; P.len2 = len2;
    P.L = atr_imin(len1, len2);                                       // align/__init__.py:259-265
    P.k = ip.k_by_len[atr_imin(P.L, INS_MAX_LEN)];
    P.nhits = 0; P.has_best = false; P.best_prob = 0.0;
    P.best_j = P.best_cost = 0; P.best_e1 = P.best_e2 = -1;
#pragma unroll
    for (int w = 0; w < D; ++w) P.x[w] = 0;
}

// Mismatches of the read overhang starting at base j = 8*B + r against an adapter.
template <int D, int B>
ATR_DEV int overhang_mismatches(const uint32_t (&s)[D], int r, const uint32_t (&a)[INS_AW], int alen, int mode) {
    int mism = 0;
#pragma unroll
    for (int w = 0; w < INS_AW; ++w) {
        const uint32_t lo = (B + w < D) ? s[(B + w < D) ? B + w : 0] : 0u;
        const uint32_t hi = (B + w + 1 < D) ? s[(B + w + 1 < D) ? B + w + 1 : 0] : 0u;
        uint32_t ov = (r >= 8) ? hi : ((lo >> (4 * r)) | (hi << (32 - 4 * r)));   // r in 1..8
        uint32_t bad;
        if (mode == INS_CMP_EQ) {
            bad = nibble_nonzero(ov ^ a[w]);
        } else {
            if (mode == INS_CMP_AND_READ_ACGT) ov = acgt_only(ov);
            bad = nibble_nonzero(ov & a[w]) ^ 0x11111111u;
        }
        mism += atr_popc(bad & base_mask(alen - 8 * w));
    }
    return mism;
}

// Overlap length j = 8*B + r (r = 1..8).  ycur holds the not-yet-consumed bases of
// bfrev(read2 dword B), next base in its top nibble.
template <int D, int B>
ATR_DEV void pair_step(PairState<D> &P, const InsertParams &ip, int r, uint32_t &ycur) {
    const int j = 8 * B + r;
    // X <<= one base, bringing in the next base of rc2 at the bottom
#pragma unroll
    for (int w = B; w >= 1; --w) P.x[w] = (P.x[w] << 4) | (P.x[w - 1] >> 28);
    P.x[0] = (P.x[0] << 4) | (ycur >> 28);
    ycur <<= 4;
    // Hamming(read1[0:j], rc2[L-j:L]), byte-equality semantics (_align.pyx:690)
    int cost = 0;
#pragma unroll
    for (int w = 0; w <= B; ++w) {
        uint32_t bad = nibble_nonzero(P.x[w] ^ P.s1[w]);
        if (w == B) bad &= base_mask(r);
        cost += atr_popc(bad);
    }
    if (j > P.L || P.nhits >= INS_MAX_MATCHES) return;
    // the hit test of MultiAligner.locate (_align.pyx:713-745)
    if (cost > P.k || j < ip.min_insert_overlap || cost > (int)ip.thr_ins[j]) return;
    P.nhits += 1;
    if (cost == 0 && j == P.L) P.has_best = false;             // exact full overlap: the only hit (:737-741, :767-768)
    const int matches = j - cost;
    const double prob = ip.rmp_insert[(size_t)j * ip.rmp_ld + matches];   // align/__init__.py:359
    if (!(prob <= ip.insert_max_rmp)) return;
    if (P.has_best && !(prob < P.best_prob)) return;            // stable ascending-probability order
    const int offset = P.L - j;
    int e1 = -1, e2 = -1;
    if (offset >= ip.min_adapter_overlap) {                     // align/__init__.py:270-276
        const int al1 = atr_imin(offset, ip.alen1), al2 = atr_imin(offset, ip.alen2);
        e1 = overhang_mismatches<D, B>(P.s1, r, ip.a1, al1, ip.cmp_mode);
        e2 = overhang_mismatches<D, B>(P.s2, r, ip.a2, al2, ip.cmp_mode);
        if (e1 > (int)ip.mm_by_alen[al1] && e2 > (int)ip.mm_by_alen[al2]) return;    // :297-300
        if (atr_imin(al1, al2) > ip.adapter_check_cutoff) {                           // :302-306
            const double p1 = ip.rmp_adapter[(size_t)al1 * ip.rmp_ld + (al1 - e1)];
            const double p2 = ip.rmp_adapter[(size_t)al2 * ip.rmp_ld + (al2 - e2)];
            if (p1 * p2 > ip.adapter_max_rmp) return;
        }
    }
    P.has_best = true; P.best_prob = prob; P.best_j = j; P.best_cost = cost; P.best_e1 = e1; P.best_e2 = e2;
}

// Three 16-byte records per pair: the insert match, Match 1, Match 2
// (refstop / astop == -1: absent).
template <int D>
ATR_DEV void pair_result(const PairState<D> &P, const InsertParams &ip, uint32_t rec[12]) {
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

// The whole sweep for one lane: blocks B = 0..D-1 (compile time), bases r = 1..8.
// jmax is the wave-uniform upper bound of the overlap length (max L over the lanes).
template <int D, int B>
struct InsertSweep {
    static ATR_DEV_MEMBER void run(PairState<D> &P, const InsertParams &ip, int jmax) {
        if (8 * B >= jmax) return;                                    // wave-uniform
        uint32_t ycur = atr_bfrev(P.s2[B]);                           // rc2 dword D-1-B
#ifndef ATR_HOST_EMU
#pragma unroll 1
#endif
        for (int r = 1; r <= 8; ++r) {
            if (8 * B + r > jmax) break;
            pair_step<D, B>(P, ip, r, ycur);
        }
        InsertSweep<D, B + 1>::run(P, ip, jmax);
    }
};
template <int D>
struct InsertSweep<D, D> {
    static ATR_DEV_MEMBER void run(PairState<D> &, const InsertParams &, int) {}
};

}  // namespace atr
#endif
