// wave_kernel.hip -- locate_wave_kernel: one read per wavefront, anti-diagonal sweep (wave_core.hpp).
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstring>
#include <type_traits>
#include "aligner_host.hpp"
#include "wave_sweep.hpp"

namespace atr {

// The aligner's parameters and its query translate table (_align.pyx:243-248, :292-297) as ONE kernel argument -- the
// first, so that thresholds and table entries can be fetched from the kernel-argument segment with per-lane indices.
struct WaveParams {
    LocateParams p;
    uint8_t table[256];
};

// packed != nullptr: reads in the tile64 layout.  Otherwise `ascii`: one row of ASCII per read (row stride and base
// address multiples of four; device memory or page-locked host memory -- atr_locate_one reads the caller's read straight
// from its staging buffer), translated here.
template <bool XREP, bool SQ, int R>
__global__ __launch_bounds__(64) void locate_wave_kernel(const WaveParams wp, const uint4 *__restrict__ packed,
                                                         const uint8_t *__restrict__ ascii, long long ascii_stride,
                                                         const int32_t *__restrict__ lens, long long nreads, int nchunks,
                                                         int max_len, uint4 *__restrict__ out) {
    const LocateParams &p = wp.p;
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ __attribute__((aligned(16))) uint32_t s_code[WAVE_CODE_PAD + (ATR_MAX_READ_LEN + 31) / 32 * 32 + 2 * WAVE_CODE_PAD];

    const int lane = threadIdx.x;
    const long long r = blockIdx.x;
    const int n = __builtin_amdgcn_readfirstlane(lens ? min(max(lens[r], 0), max_len) : max_len);    // (never beyond the layout)
    // thresholds and the read's codes (one dword per column) into LDS
    {   // (read from the kernel-argument segment itself -- `p` is the first argument: indexing the by-value copy with
        //  a run-time index would move all of it to scratch memory)
        const int16_t *kthr = (const int16_t *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WaveParams, p) +
                                                offsetof(LocateParams, thr));
        for (int i = lane; i <= p.m + 1; i += 64) s_thr[i] = kthr[i];
    }
    if (!packed) {
        const uint8_t *ktab = (const uint8_t *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WaveParams, table);
        const uint32_t *row = (const uint32_t *)(ascii + (size_t)r * ascii_stride);
        for (int i = lane; 4 * i < n; i += 64) {                     // four bases per lane and trip
            const uint32_t w = row[i];
            uint4 c;
            c.x = ktab[w & 255u] & 15u; c.y = ktab[(w >> 8) & 255u] & 15u; c.z = ktab[(w >> 16) & 255u] & 15u; c.w = ktab[w >> 24] & 15u;
            *(uint4 *)(s_code + WAVE_CODE_PAD + 4 * i) = c;          // (bytes beyond n: never looked at by an active lane)
        }
    } else if (lane < (n + 31) / 32) {
        const uint4 v = packed[((size_t)(r >> 6) * nchunks + lane) * 64 + (size_t)(r & 63)];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint4 *dst = (uint4 *)(s_code + WAVE_CODE_PAD + 32 * lane);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            dst[2 * d] = make_uint4(w[d] & 15u, (w[d] >> 4) & 15u, (w[d] >> 8) & 15u, (w[d] >> 12) & 15u);
            dst[2 * d + 1] = make_uint4((w[d] >> 16) & 15u, (w[d] >> 20) & 15u, (w[d] >> 24) & 15u, w[d] >> 28);
        }
    }
    uint32_t rec[4];
    wave_locate<XREP, SQ, R>(p, s_thr, s_code + WAVE_CODE_PAD, n, lane, rec);
    if (lane == 0) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

template <int R>
static void launch_wave_r(const atr_aligner *a, const uint4 *packed, const uint8_t *ascii, long long stride, const int32_t *lens,
                          long long nreads, int nchunks, int max_len, uint4 *out, hipStream_t st) {
    const dim3 grid((unsigned)nreads), block(64);
    const bool xrep = (a->flags & ATR_STOP_WITHIN_SEQ2) != 0, sq = (a->flags & ATR_START_WITHIN_SEQ2) != 0;
    WaveParams wp;
    wp.p = a->p;
    memcpy(wp.table, a->qtable, 256);
    if (xrep && sq)  hipLaunchKernelGGL((locate_wave_kernel<true, true, R>), grid, block, 0, st, wp, packed, ascii, stride, lens, nreads, nchunks, max_len, out);
    if (xrep && !sq) hipLaunchKernelGGL((locate_wave_kernel<true, false, R>), grid, block, 0, st, wp, packed, ascii, stride, lens, nreads, nchunks, max_len, out);
    if (!xrep && sq) hipLaunchKernelGGL((locate_wave_kernel<false, true, R>), grid, block, 0, st, wp, packed, ascii, stride, lens, nreads, nchunks, max_len, out);
    if (!xrep && !sq) hipLaunchKernelGGL((locate_wave_kernel<false, false, R>), grid, block, 0, st, wp, packed, ascii, stride, lens, nreads, nchunks, max_len, out);
}

// packed == nullptr: the reads as ASCII rows (see the kernel)
int launch_locate_wave(const atr_aligner *a, const uint4 *packed, const uint8_t *ascii, long long stride, const int32_t *lens,
                       long long nreads, int nchunks, int max_len, uint4 *out, hipStream_t st) {
    switch (wave_pair_rows(a->p.m)) {                               // rows per lane: 1 up to 63 bases, 2 up to 127, 3 for 128
        case 1: launch_wave_r<1>(a, packed, ascii, stride, lens, nreads, nchunks, max_len, out, st); break;
        case 2: launch_wave_r<2>(a, packed, ascii, stride, lens, nreads, nchunks, max_len, out, st); break;
        default: launch_wave_r<3>(a, packed, ascii, stride, lens, nreads, nchunks, max_len, out, st); break;
    }
    return (int)hipGetLastError();
}

}  // namespace atr
