// fast_work.hpp -- what the kernels of the filtered pipelines share and the run-time compiled pre-pass
// (piece_filter.hpp, jit.hpp) needs as well: the carve-up of the caller's workspace, the tile ranges of the
// persistent grids, wave reductions.  Device code only (plus two host helpers kept out of hiprtc's sight).
#ifndef ATR_FAST_WORK_HPP
#define ATR_FAST_WORK_HPP

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "filter_core.hpp"
#include "piece_core.hpp"

namespace atr {

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

constexpr int FAST_BLOCKS = 8192;                   // grid of K1 / K3: four rounds of resident blocks -- finer than one
                                                    // block per CU slot balances the chip (K1 0.74 -> 0.65 ms)

struct FastWork {                                    // carve-up of the caller's workspace
    uint32_t *win;                                   // [nreads] window words (single aligner: per tile, the unresolved reads' words first)
    uint64_t *mask;                                  // [ntiles] single aligner: the tile's unresolved lanes
    uint2 *order;                                    // [nreads] (read, its window word): the DP kernels read both with ONE
                                                     // coalesced load instead of gathering win[read] (a 64-byte line per read)
    uint32_t *counts;                                // [FAST_BLOCKS][nbins] -> in-bin offsets after K2a
    uint32_t *chunks;                                // [SCAN_CHUNKS][nbins] chunk totals -> chunk offsets inside the bin (K2b)
    uint32_t *binbase;                               // [nbins + 1] first slot of each bin after K2b; [nbins] = total
    uint32_t *total;                                 // [1] number of unresolved reads
    int nbins;                                       // FILTER_BINS, or FILTER_BINS per adapter of a linked set
    int nused;                                       // blocks K1 / K3 are launched with (FAST_BLOCKS; fewer for a short batch:
                                                     // the scans then only walk the histogram rows that exist)
    int lpw;                                         // tasks a wave of the DP kernels takes: 64, or 0 = as few as the grid
                                                     // allows (short batches, dp_lanes_per_wave)
    uint2 *tmp;                                      // [nreads] two-pass pre-pass (piece_kernels.hip): per block, the list of its
                                                     // unresolved (read, window word) pairs, at the block's first read
    uint32_t *lcount;                                // [FAST_BLOCKS] entries of each block's list
    uint32_t *wide;                                  // [nreads] (the `win` region) per block: its reads that take the full sweep
    uint32_t *nwide;                                 // (unused)
    uint4 *wdata;                                    // [wcap][nchunks] per block: a copy of their planes, written by the wave that had
    long long wcap;                                  //   them in registers (a block's reads beyond its share are gathered from the batch)
    uint4 *tdata;                                    // [nreads][2] next to tmp: the 64 codes from the entry's first diagonal on
                                                     //   (what band_stage would gather from the batch, 16 bytes per 128-byte line)
    uint32_t *dref;                                  // [nreads] next to order: the slot's index into tdata
    int fused;                                       // 1: no scan launches -- the pre-pass blocks take their offsets inside the bins
                                                     // with atomics on `chunks`[bin] (fused_hist_flush / fused_bin_bases below)
};

#ifndef __HIPCC_RTC__
// reads whose planes the two-pass pre-pass copies for its full-sweep kernel (the others are gathered): a quarter
inline long long fast_wide_cap(long long nreads) { return nreads / 4 + 16ll * FAST_BLOCKS + 64; }   // (+ 16 per block)

// (the dense buffers of the two-pass pre-pass only in the single-aligner workspace, nbins == FILTER_BINS)
inline size_t fast_work_bytes(long long nreads, int nbins = FILTER_BINS) {
    const bool pieces = nbins == FILTER_BINS;
    return (size_t)nreads * 20 + (size_t)((nreads + 63) / 64) * 8 + (size_t)(FAST_BLOCKS + FAST_BLOCKS / 64) * nbins * 4 + (size_t)(nbins + 1) * 4 +
           (size_t)FAST_BLOCKS * 4 + 256 + 64 +
           (pieces ? (size_t)nreads * (32 + 4) + (size_t)fast_wide_cap(nreads) * PIECE_MAX_WORDS * 16 : 0);   // tdata, dref, wdata
}

inline FastWork fast_carve(void *work, long long nreads, int nbins = FILTER_BINS) {
    FastWork w;
    w.win = (uint32_t *)work;
    w.order = (uint2 *)(w.win + ((nreads + 1) & ~1ll));              // 8-byte aligned
    w.mask = (uint64_t *)(w.order + nreads);
    w.counts = (uint32_t *)(((uintptr_t)(w.mask + (nreads + 63) / 64) + 15) & ~(uintptr_t)15);    // K2a reads 16-byte segments
    w.chunks = w.counts + (size_t)FAST_BLOCKS * nbins;
    w.binbase = w.chunks + (size_t)(FAST_BLOCKS / 64) * nbins;          // SCAN_CHUNKS rows (SCAN_CHUNK = 64, below)
    w.total = w.binbase + nbins + 1;
    w.lcount = w.total + 1;
    w.nwide = w.lcount + FAST_BLOCKS;
    w.wide = w.win;
    w.tmp = (uint2 *)(((uintptr_t)(w.nwide + 1) + 7) & ~(uintptr_t)7);
    w.tdata = (uint4 *)(((uintptr_t)(w.tmp + nreads) + 15) & ~(uintptr_t)15);
    w.wdata = w.tdata + (size_t)nreads * 2;
    w.wcap = fast_wide_cap(nreads);
    w.dref = (uint32_t *)(w.wdata + (size_t)w.wcap * PIECE_MAX_WORDS);
    if (nbins != FILTER_BINS) { w.tdata = nullptr; w.wdata = nullptr; w.dref = nullptr; w.wcap = 0; }
    w.nbins = nbins;
    w.nused = FAST_BLOCKS;
    w.lpw = 64;
    w.fused = 0;
    return w;
}
// what the first half of a plane64 locate call (piece_kernels.hip: launch_planes_prepass) hands to its second half
struct PlanesCall { FastWork wk; int nw; };

// ATR_FUSED_SCAN=0: the two scan launches instead of the atomics (A/B switch)
inline bool fast_fused_scan() {
    static const int v = [] { const char *e = getenv("ATR_FUSED_SCAN"); return (e && e[0] == '0') ? 0 : 1; }();
    return v != 0;
}
#endif  // __HIPCC_RTC__

// tiles [t0, t1) owned by a block of the persistent grid of `nblocks` blocks
__device__ __forceinline__ void block_tiles(long long ntiles, long long &t0, long long &t1, int nblocks = FAST_BLOCKS) {
    const long long per = (ntiles + nblocks - 1) / nblocks;
    t0 = min(ntiles, per * (long long)blockIdx.x);
    t1 = min(ntiles, t0 + per);
}


// The scan-free form of K2 (long batches of the two-pass pre-pass and of linked sets): the pre-pass block adds its
// count of every bin it filled to the bin's total -- `chunks`[bin], zeroed by the host before the launch -- and keeps the
// value the atomic returned as its offset inside the bin (`counts`[block][bin]): the order of the blocks inside a bin is
// whatever the atomics made it, which no result depends on (a task writes the record of its own read).  The two scan
// launches (13 us of C2's step, 39 of C4's) are gone; what is left of K2 is the exclusive scan of the nbins totals, done
// by every block of the scatter pass for itself (256 threads, nbins <= 1024), block 0 leaving `binbase` / `total` for the
// DP kernels.
__device__ __forceinline__ void fused_hist_flush(const FastWork &wk, const uint32_t *s_hist) {
    for (int b = threadIdx.x; b < wk.nbins; b += blockDim.x) {
        const uint32_t c = s_hist[b];
        wk.counts[(size_t)blockIdx.x * wk.nbins + b] = !wk.fused ? c : c ? atomicAdd(&wk.chunks[b], c) : 0u;
    }
}
// s_base: [nbins + 1] in LDS, s_tmp: [256]; a block of 256 threads
__device__ __forceinline__ void fused_bin_bases(const FastWork &wk, uint32_t *s_base, uint32_t *s_tmp) {
    const int t = threadIdx.x, per = wk.nbins >> 8;                // 1 .. 4 consecutive bins per thread
    uint32_t v[4], sum = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = i < per ? wk.chunks[t * per + i] : 0u; sum += v[i]; }
    // inclusive scan of the 256 partial sums: inside the wave by shuffles, the four waves' totals through LDS (two
    // barriers; the scatter pass of a linked set runs 8 192 blocks through this prologue)
    const int lane = t & 63, wv = t >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wv; ++w) run += s_tmp[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < per) { s_base[t * per + i] = run; run += v[i]; }
    if (t == 255) s_base[wk.nbins] = run;
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int b = t; b <= wk.nbins; b += 256) wk.binbase[b] = s_base[b];
        if (t == 0) wk.total[0] = s_base[wk.nbins];
    }
}

}  // namespace atr
#endif
