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

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atr {

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

// sw: LDS dwords; the row starts at byte 4 k + sh of it and has n bases; dwords from index `limit` on are not read.
// dst: the lane's chunk 0 (chunks are 64 uint4 apart).
__device__ __forceinline__ void pack_planes_row_fast(const uint32_t *sw, uint32_t k, uint32_t limit, uint32_t sh, int n, int nchunks,
                                                     const uint8_t *s_tab, uint4 *dst, bool &zero_seen) {
    const PackLetters L = pack_letters(s_tab);
    uint32_t lo = k < limit ? sw[k] : 0u;
    for (int c = 0; c < nchunks; ++c) {
        uint32_t X = 0u, Y = 0u, V = 0u, slow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int j = 32 * c + 4 * g;
            const uint32_t at = k + 8u * (uint32_t)c + (uint32_t)g + 1u;
            const uint32_t hi = (j < n && at < limit) ? sw[at] : 0u;   // (nothing is read behind the read's last dword)
            uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
            lo = hi;
            const int left = n - j;                                 // bases of this dword inside the read
            if (left < 4) w = left <= 0 ? 0u : (w & ((1u << (8 * left)) - 1u));
            const uint32_t idx = (w >> 1) & 0x03030303u;
            if (L.ok && __builtin_amdgcn_perm(0u, PACK_LETTERS_BY_INDEX, idx) == w) {
                uint32_t z = idx | (idx >> 6);
                z |= z >> 12;
                const uint32_t r = z & 0xFFu;                      // the four index pairs, base b at bits 2b, 2b + 1
                if (g < 4) X |= r << (8 * g); else Y |= r << (8 * (g - 4));
                V |= 0xFu << (4 * g);
            } else if (left > 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < left) {
                        const uint32_t code = s_tab[(w >> (8 * b)) & 255u] & 15u;
                        zero_seen = zero_seen || code == 0u;
#pragma unroll
                        for (int p = 0; p < 4; ++p) slow[p] |= ((code >> p) & 1u) << (4 * g + b);
                    }
                }
            }
        }
        // even / odd bits of X (bases 0 .. 15) and Y (16 .. 31) -> I0, I1
        const auto even16 = [](uint32_t x) {
            x &= 0x55555555u;
            x = (x | (x >> 1)) & 0x33333333u;
            x = (x | (x >> 2)) & 0x0F0F0F0Fu;
            x = (x | (x >> 4)) & 0x00FF00FFu;
            return (x | (x >> 8)) & 0xFFFFu;
        };
        const uint32_t I0 = even16(X) | (even16(Y) << 16), I1 = even16(X >> 1) | (even16(Y >> 1) << 16);
        const uint32_t m0 = ~I1 & ~I0 & V, m1 = ~I1 & I0 & V, m2 = I1 & ~I0 & V, m3 = I1 & I0 & V;   // A C T G
        uint32_t pl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
            pl[p] = (((L.cA >> p) & 1u) ? m0 : 0u) | (((L.cC >> p) & 1u) ? m1 : 0u) | (((L.cT >> p) & 1u) ? m2 : 0u) |
                    (((L.cG >> p) & 1u) ? m3 : 0u) | slow[p];
        dst[(size_t)c * 64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
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
    const long long rows_here = min<long long>(64, nreads - tile * 64);
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
