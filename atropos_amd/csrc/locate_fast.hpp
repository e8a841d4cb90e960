// locate_fast.hpp -- the filtered locate pipeline (see filter_core.hpp for why it is exact):
//
//   K1 filter_kernel   one read per lane: Myers bit-vector sweep over the whole read
//                      (~30 VALU ops per column instead of ~7 per CELL); writes the result
//                      record of every read it can resolve (None / first perfect hit) and a
//                      window word for the rest; per-block histogram of window starts.
//   K2 scan_kernel     exclusive scan of the (bin, block) histogram -> scatter offsets.
//   K3 scatter_kernel  read indices of the unresolved reads, ordered by window start, so
//                      that the 64 lanes of a K4 wave sweep nearly the same columns.
//   K4 window_kernel   the packed-word DP (locate_core.hpp, window mode) over each read's
//                      window only; reads are gathered by index (16-byte chunk loads).
//
// Blocks of K1/K3 own contiguous tile ranges, so K3 needs no global atomics: its offsets
// come from K2's scan and an LDS cursor per bin.
#ifndef ATR_LOCATE_FAST_HPP
#define ATR_LOCATE_FAST_HPP

#include "locate_kernel.hpp"
#include "filter_core.hpp"

namespace atr {

constexpr int FAST_BLOCKS = 2048;                   // persistent grid of K1 / K3

struct FastWork {                                    // carve-up of the caller's workspace
    uint32_t *win;                                   // [nreads]
    uint32_t *order;                                 // [nreads]
    uint32_t *counts;                                // [FAST_BLOCKS][FILTER_BINS] -> offsets after K2
    uint32_t *total;                                 // [1] number of unresolved reads
};

inline size_t fast_work_bytes(long long nreads) {
    return (size_t)nreads * 8 + (size_t)FAST_BLOCKS * FILTER_BINS * 4 + 256;
}

inline FastWork fast_carve(void *work, long long nreads) {
    FastWork w;
    w.win = (uint32_t *)work;
    w.order = w.win + nreads;
    w.counts = w.order + nreads;
    w.total = w.counts + (size_t)FAST_BLOCKS * FILTER_BINS;
    return w;
}

// tiles [t0, t1) owned by a block of the persistent grid
__device__ __forceinline__ void block_tiles(long long ntiles, long long &t0, long long &t1) {
    const long long per = (ntiles + FAST_BLOCKS - 1) / FAST_BLOCKS;
    t0 = min(ntiles, per * (long long)blockIdx.x);
    t1 = min(ntiles, t0 + per);
}

#ifdef ATR_DEFINE_FILTER_KERNELS   // K1..K3 are defined once, in filter_kernels.hip
__global__ __launch_bounds__(256) void filter_kernel(const LocateParams p, const FilterParams fp,
                                                     const uint4 *__restrict__ packed,
                                                     const int32_t *__restrict__ lens, long long nreads, int nchunks,
                                                     int max_len, uint4 *__restrict__ out, FastWork wk) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint64_t s_peq[16];
    __shared__ uint32_t s_hist[FILTER_BINS];
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    for (int i = threadIdx.x; i <= u.m + 1; i += 256) s_thr[i] = p.thr[i];
    if (threadIdx.x < 16) s_peq[threadIdx.x] = fp.peq[threadIdx.x];
    if (threadIdx.x < FILTER_BINS) s_hist[threadIdx.x] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1);
    for (long long tile = t0 + wave; tile < t1; tile += 4) {
        const long long r = tile * 64 + lane;
        const bool live = r < nreads;
        const int n = live ? (lens ? lens[r] : max_len) : 0;
        FilterState F;
        filter_init(F, u);
        const int jhi = wave_max_i32(n);
        const uint4 *tp = packed + (size_t)tile * nchunks * 64 + lane;
        if (jhi > 0) {
            const int c1 = (jhi + 31) >> 5;
            uint4 nxt = tp[0];
            for (int c = 0; c < c1; ++c) {
                uint4 cur = nxt;
                if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
                int j = c * 32;
#pragma unroll 1
                for (int d = 0; d < 4; ++d) {
                    uint32_t w = cur.x;
                    cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
#pragma unroll 1
                    for (int b = 0; b < 8; ++b) {
                        ++j;
                        const uint32_t q = w & 15u;
                        w >>= 4;
                        if (j <= n) filter_step(F, u, s_peq[q], j);
                    }
                }
            }
        }
        if (live) {
            uint32_t rec[4];
            const uint32_t ww = filter_decide(F, u, n, s_thr, rec);
            wk.win[r] = ww;
            if (!window_valid(ww)) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
            else atomicAdd(&s_hist[window_lo(ww) >> 3], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < FILTER_BINS) wk.counts[(size_t)blockIdx.x * FILTER_BINS + threadIdx.x] = s_hist[threadIdx.x];
}

// offsets[block][bin] = number of unresolved reads in earlier bins, plus those of the same
// bin in earlier blocks.  One block of FILTER_BINS*... threads is plenty (196 k counters).
__global__ __launch_bounds__(1024) void scan_kernel(FastWork wk) {
    __shared__ uint32_t s_bin[FILTER_BINS];
    // per-bin totals: thread b sums its column
    for (int b = threadIdx.x; b < FILTER_BINS; b += 1024) {
        uint32_t tot = 0;
        for (int k = 0; k < FAST_BLOCKS; ++k) tot += wk.counts[(size_t)k * FILTER_BINS + b];
        s_bin[b] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < FILTER_BINS; ++b) { const uint32_t t = s_bin[b]; s_bin[b] = run; run += t; }
        wk.total[0] = run;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < FILTER_BINS; b += 1024) {
        uint32_t run = s_bin[b];
        for (int k = 0; k < FAST_BLOCKS; ++k) {
            const uint32_t c = wk.counts[(size_t)k * FILTER_BINS + b];
            wk.counts[(size_t)k * FILTER_BINS + b] = run;
            run += c;
        }
    }
}

__global__ __launch_bounds__(256) void scatter_kernel(long long nreads, FastWork wk) {
    __shared__ uint32_t s_cur[FILTER_BINS];
    if (threadIdx.x < FILTER_BINS) s_cur[threadIdx.x] = wk.counts[(size_t)blockIdx.x * FILTER_BINS + threadIdx.x];
    __syncthreads();
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1);
    for (long long r = t0 * 64 + threadIdx.x; r < min(nreads, t1 * 64); r += 256) {
        const uint32_t ww = wk.win[r];
        if (window_valid(ww)) wk.order[atomicAdd(&s_cur[window_lo(ww) >> 3], 1u)] = (uint32_t)r;
    }
}

#endif  // ATR_DEFINE_FILTER_KERNELS

template <int MT, bool EQ, bool NOINDEL>
__global__ __launch_bounds__(256) void window_kernel(const LocateParams p, const uint4 *__restrict__ packed,
                                                     const int32_t *__restrict__ lens, long long nreads,
                                                     int nchunks, int max_len, uint4 *__restrict__ out, FastWork wk) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_init[ATR_MAX_REF_LEN + 1];
    const Uniform u = make_uniform(p, MT);
    for (int i = threadIdx.x; i <= MT + 1; i += 256) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    __syncthreads();
    const long long total = (long long)wk.total[0];
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    if (((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 >= total) return;     // whole wave
    const bool live = slot < total;
    const long long r = live ? (long long)wk.order[slot] : 0;
    const uint32_t ww = live ? wk.win[r] : 0u;
    const int n = live ? (lens ? lens[r] : max_len) : 0;
    const int j_lo = window_lo(ww), j_hi = live ? window_hi(ww) : 0;

    LaneState<MT> L;
    lane_init_window<MT>(L, u, n, j_lo, j_hi, live && window_scan(ww), s_init, s_thr);

    const bool has_window = live && j_hi > j_lo;
    const int jlo = wave_min_i32(has_window ? j_lo : 0x7fffffff);
    const int jhi = wave_max_i32(has_window ? j_hi : 0);
    // this lane's read inside the tile64 layout (gathered: 16 bytes per lane per chunk)
    const uint4 *tp = packed + ((size_t)(r >> 6) * nchunks) * 64 + (r & 63);
    if (jhi > jlo) {
        const int c0 = jlo >> 5, c1 = (jhi + 31) >> 5;
        uint4 nxt = tp[(size_t)c0 * 64];
        for (int c = c0; c < c1; ++c) {
            uint4 cur = nxt;
            if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
            int j = c * 32;
#pragma unroll 1
            for (int d = 0; d < 4; ++d) {
                uint32_t w = cur.x;
                cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
#pragma unroll 1
                for (int b = 0; b < 8; ++b) {
                    ++j;
                    const uint32_t q = w & 15u;
                    w >>= 4;
                    if (j <= jlo || j > jhi) continue;
                    lane_step<MT, EQ, NOINDEL, true>(L, p, u, j, q, s_thr);
                }
            }
        }
    }
    if (live) {
        uint32_t rec[4];
        lane_result<MT>(L, u, rec);
        out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

typedef int (*window_launcher)(const atr_aligner *, const uint4 *, const int32_t *, long long, int, int, uint4 *,
                               FastWork, hipStream_t);

template <int MT>
int launch_window_mt(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                     int max_len, uint4 *out, FastWork wk, hipStream_t st) {
    const bool eqmode = !(a->wildcard_ref || a->wildcard_query);
    const bool noindel = a->indel_cost > a->p.k;
    const dim3 grid((unsigned)((nreads + 255) / 256)), block(256);
    if (eqmode) {
        if (noindel) hipLaunchKernelGGL((window_kernel<MT, true, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk);
        else         hipLaunchKernelGGL((window_kernel<MT, true, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk);
    } else {
        if (noindel) hipLaunchKernelGGL((window_kernel<MT, false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk);
        else         hipLaunchKernelGGL((window_kernel<MT, false, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk);
    }
    return (int)hipGetLastError();
}

constexpr int WINDOW_SIZES = FILTER_MAX_M / ROW_GRAN;        // MT = 4 .. 64

}  // namespace atr
#endif
