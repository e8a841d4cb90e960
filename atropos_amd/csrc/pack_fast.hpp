// pack_fast.hpp -- device only: a read's row, staged in LDS, turned into plane64 / tile64 chunks FOUR bases per step.
// Shared by pack_kernel (api.hip: atr_pack_reads / atr_pack_planes, rows of an ASCII matrix) and pack_records_kernel
// (fastq_kernels.hip: atr_pack_records, rows cut out of a FASTQ chunk).  The reference does bytes.translate(table) per
// read (_align.pyx:243-248, :292-297); the byte-by-byte form (locate_core.hpp: pack_word / pack_planes_chunk -- a byte
// read, a table read and eight VALU ops per base) stays for the unstaged kernels and is what these routines fall back
// to, dword by dword, for every byte that is not one of 'A' 'C' 'G' 'T'.
//
// A dword of the row -- read aligned from LDS, funnel-shifted by the row's byte offset -- holds four bases.  With the
// two-bit index (c >> 1) & 3 ('A' 0, 'C' 1, 'T' 2, 'G' 3) v_perm_b32 rebuilds the four letters: equal to the dword <=>
// all four bytes are of the four letters.  tile64: their codes by a second v_perm_b32 from the table's four entries,
// nibbles pushed together.  plane64: the index pairs compacted to a byte, the chunk's 32 pairs de-interleaved into two
// bit planes I0 / I1, the four code planes boolean functions of those (the four codes are wave-uniform).
#ifndef ATR_PACK_FAST_HPP
#define ATR_PACK_FAST_HPP

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace atr {

struct PackTableArg { uint8_t t[256]; };           // a translate table as a kernel argument

struct PackLetters {
    uint32_t cA, cC, cG, cT, lutc;
    bool ok;                                         // all four letters have a code: the fast path applies
};
__device__ __forceinline__ PackLetters pack_letters(const uint8_t *s_tab) {
    PackLetters L;
    L.cA = s_tab['A'] & 15u; L.cC = s_tab['C'] & 15u; L.cG = s_tab['G'] & 15u; L.cT = s_tab['T'] & 15u;
    L.ok = L.cA && L.cC && L.cG && L.cT;
    L.lutc = L.cA | (L.cC << 8) | (L.cT << 16) | (L.cG << 24);        // index order: A C T G
    return L;
}
constexpr uint32_t PACK_LETTERS_BY_INDEX = 0x47544341u;                 // 'A' 'C' 'T' 'G'

// One chunk (32 bases, chunk c) of a row as four bit planes.  sw: LDS dwords; the row starts at byte 4 k + sh of it and
// has n bases; dwords from index `limit` on are not read; lo: the row's dword 8 c (carried from chunk to chunk: the
// caller starts with sw[k]).
//
// Round 6.  Rounds 4 / 5 decided per DWORD whether its four bytes were all of A C G T and took a byte-by-byte path
// otherwise -- a divergent branch that SOME lane of the wave takes in a quarter of the dwords (0.1 % N), so the wave
// paid for both paths most of the time (pack_kernel: 0.61 ms per 10 M x 150 bp, issue bound).  Now the chunk is packed
// as if every base were one of the four letters -- the two index bits of a base are bits 1 and 2 of its byte; a
// v_dot4_u32_u8 with the weights 1 2 4 8 (16 32 64 128 for the odd dword) gathers one bit of four bytes into a nibble,
// two dwords into a byte: 11 VALU ops per dword, no branch -- while `bad` collects perm(letters, index) ^ dword.  Only a
// lane with a byte that is none of the four letters (a chunk in 30) then finds its invalid positions (a third dot
// plane over the nonzero bytes) and walks them one by one through the table: the wave runs as many steps as its worst
// lane has such bases -- one or two.
__device__ __forceinline__ uint4 pack_planes_chunk_fast(const uint32_t *sw, uint32_t k, uint32_t limit, uint32_t sh, int n, int c,
                                                        const PackLetters &L, const uint8_t *s_tab, uint32_t &lo, bool &zero_seen) {
    uint32_t t[8], b0[4] = {0u, 0u, 0u, 0u}, b1[4] = {0u, 0u, 0u, 0u}, bad = 0u;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int j = 32 * c + 4 * g;
        const uint32_t at = k + 8u * (uint32_t)c + (uint32_t)g + 1u;
        const uint32_t hi = (j < n && at < limit) ? sw[at] : 0u;   // (nothing is read behind the read's last dword)
        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
        lo = hi;
        const uint32_t idx = (w >> 1) & 0x03030303u;
        t[g] = __builtin_amdgcn_perm(0u, PACK_LETTERS_BY_INDEX, idx) ^ w;      // nonzero byte <=> not one of the four letters
        bad |= t[g];
        const uint32_t wt = (g & 1) ? 0x80402010u : 0x08040201u;
        b0[g >> 1] = __builtin_amdgcn_udot4(idx & 0x01010101u, wt, b0[g >> 1], false);
        b1[g >> 1] = __builtin_amdgcn_udot4(idx & 0x02020202u, wt, b1[g >> 1], false);   // (twice the byte: shifted back below)
    }
    const uint32_t I0 = b0[0] | (b0[1] << 8) | (b0[2] << 16) | (b0[3] << 24);
    const uint32_t I1 = (b1[0] >> 1) | (b1[1] << 7) | (b1[2] << 15) | (b1[3] << 23);
    const int left = n - 32 * c;                                    // bases of this chunk inside the read
    const uint32_t lm = left >= 32 ? ~0u : left <= 0 ? 0u : (1u << left) - 1u;
    uint32_t inv = L.ok ? 0u : lm, slow[4] = {0u, 0u, 0u, 0u};
    if (L.ok && bad != 0u) {
        uint32_t z[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const uint32_t nzb = ((t[g] | ((t[g] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;   // 1 per nonzero byte
            z[g >> 1] = __builtin_amdgcn_udot4(nzb, (g & 1) ? 0x80402010u : 0x08040201u, z[g >> 1], false);
        }
        inv = (z[0] | (z[1] << 8) | (z[2] << 16) | (z[3] << 24)) & lm;
    }
    const uint32_t V = lm & ~inv;
    if (inv != 0u) {
        const uint8_t *row = (const uint8_t *)sw + 4u * k + sh + 32u * (uint32_t)c;
        uint32_t todo = inv;
        while (todo != 0u) {
            const int i = __builtin_ctz(todo);
            todo &= todo - 1u;
            const uint32_t code = s_tab[row[i]] & 15u;
            zero_seen = zero_seen || code == 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) slow[p] |= ((code >> p) & 1u) << i;
        }
    }
    const uint32_t m0 = ~I1 & ~I0 & V, m1 = ~I1 & I0 & V, m2 = I1 & ~I0 & V, m3 = I1 & I0 & V;   // A C T G
    uint32_t pl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        pl[p] = (((L.cA >> p) & 1u) ? m0 : 0u) | (((L.cC >> p) & 1u) ? m1 : 0u) | (((L.cT >> p) & 1u) ? m2 : 0u) |
                (((L.cG >> p) & 1u) ? m3 : 0u) | slow[p];
    return make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

// dst: the lane's chunk 0 (chunks are 64 uint4 apart).
__device__ __forceinline__ void pack_planes_row_fast(const uint32_t *sw, uint32_t k, uint32_t limit, uint32_t sh, int n, int nchunks,
                                                     const uint8_t *s_tab, uint4 *dst, bool &zero_seen) {
    const PackLetters L = pack_letters(s_tab);
    uint32_t lo = k < limit ? sw[k] : 0u;
    for (int c = 0; c < nchunks; ++c) dst[(size_t)c * 64] = pack_planes_chunk_fast(sw, k, limit, sh, n, c, L, s_tab, lo, zero_seen);
}

// The same into registers (the fused ASCII entry of the two-pass pre-pass, piece_filter.hpp): pl[c] = chunk c.
template <int NCH>
__device__ __forceinline__ void pack_planes_row_regs(const uint32_t *sw, uint32_t k, uint32_t sh, int n, const uint8_t *s_tab,
                                                     uint4 (&pl)[NCH], bool &zero_seen) {
    const PackLetters L = pack_letters(s_tab);
    uint32_t lo = sw[k];
#pragma unroll
    for (int c = 0; c < NCH; ++c) pl[c] = pack_planes_chunk_fast(sw, k, 0x7fffffffu, sh, n, c, L, s_tab, lo, zero_seen);
}

__device__ __forceinline__ void pack_codes_row_fast(const uint32_t *sw, uint32_t k, uint32_t limit, uint32_t sh, int n, int nchunks,
                                                    const uint8_t *s_tab, uint4 *dst, bool &zero_seen) {
    const PackLetters L = pack_letters(s_tab);
    uint32_t lo = k < limit ? sw[k] : 0u;
    for (int c = 0; c < nchunks; ++c) {
        uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int j = 32 * c + 4 * g;
            const uint32_t at = k + 8u * (uint32_t)c + (uint32_t)g + 1u;
            const uint32_t hi = (j < n && at < limit) ? sw[at] : 0u;
            uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
            lo = hi;
            const int left = n - j;
            if (left < 4) w = left <= 0 ? 0u : (w & ((1u << (8 * left)) - 1u));
            const uint32_t idx = (w >> 1) & 0x03030303u;
            uint32_t nib = 0u;                                      // four codes, base b at bits 4b .. 4b + 3
            if (L.ok && __builtin_amdgcn_perm(0u, PACK_LETTERS_BY_INDEX, idx) == w) {
                uint32_t cc = __builtin_amdgcn_perm(0u, L.lutc, idx);
                cc = (cc | (cc >> 4)) & 0x00FF00FFu;
                nib = (cc | (cc >> 8)) & 0xFFFFu;
            } else if (left > 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < left) {
                        const uint32_t code = s_tab[(w >> (8 * b)) & 255u] & 15u;
                        zero_seen = zero_seen || code == 0u;
                        nib |= code << (4 * b);
                    }
                }
            }
            out[g >> 1] |= nib << (16 * (g & 1));
        }
        dst[(size_t)c * 64] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// The 64 rows of tile `tile` (contiguous: 64 row_stride bytes) copied into the wave's LDS stage with coalesced 16-byte
// loads -- the 16-byte aligned window around the region; returns its offset in the window (0 .. 15): row l starts at
// stage + mis + l row_stride.  [ascii, buf_end) is the caller's matrix: a 16-byte piece of the window that sticks out of
// it (before the first row of the batch, after its last) is copied byte by byte, never read as a whole.  Eight pieces per
// lane are requested before the first one is stored (one at a time the copy was a chain of need / 1024 memory round
// trips per wave -- ten for 150-byte rows).  The stage needs 64 row_stride + PACK_STAGE_SLACK bytes.
constexpr int PACK_STAGE_SLACK = 32;                // bytes behind a wave's 64 rows: the aligned window's overhang (< 16) and the dword
                                                    // reads of the four-bases-per-step path at the last row's end (< 8 more)
__device__ __forceinline__ uint32_t pack_stage_tile(uint8_t *stage, const uint8_t *ascii, long long row_stride, long long nreads,
                                                    long long tile, int lane) {
    const uint8_t *src = ascii + tile * 64 * row_stride;
    const uintptr_t mis = (uintptr_t)src & 15;
    const uint8_t *src_al = src - mis;
    const long long rows_here = nreads - tile * 64 < 64 ? nreads - tile * 64 : 64;
    const long long need = mis + rows_here * row_stride;       // bytes of the window that are ours
    const uint8_t *buf_end = ascii + nreads * row_stride;
    for (long long base = 0; base < need; base += 8 * 64 * 16) {
        uint4 buf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long o = base + (long long)u * 64 * 16 + (long long)lane * 16;
            const uint8_t *piece = src_al + o;
            buf[u] = (o < need && piece >= ascii && piece + 16 <= buf_end) ? *(const uint4 *)piece : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long o = base + (long long)u * 64 * 16 + (long long)lane * 16;
            const uint8_t *piece = src_al + o;
            if (o < need) {
                if (piece >= ascii && piece + 16 <= buf_end) {
                    *(uint4 *)(stage + o) = buf[u];
                } else {
                    for (int b = 0; b < 16; ++b) stage[o + b] = (piece + b >= ascii && piece + b < buf_end) ? piece[b] : (uint8_t)0;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);                             // LDS is visible wave-wide after the stores land
    return (uint32_t)mis;
}

}  // namespace atr
#endif
