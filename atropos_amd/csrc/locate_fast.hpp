// locate_fast.hpp -- the filtered locate pipeline (see filter_core.hpp for why it is exact):
//
//   K1  filter_kernel   one read per lane: Myers bit-vector sweep over the whole read
//                       (~28 VALU ops per column instead of ~7 per CELL); writes the result
//                       record of every read it can resolve (None / first perfect hit / perfect
//                       overlap at the read end) and a window word for the rest; per-block
//                       histogram of the scatter bins.
//   K2  scan kernels    exclusive scan of the (bin, block) histogram -> scatter offsets.
//   K3  scatter_kernel  (read index, window word) of the unresolved reads, ordered by bin
//                       (filter_core.hpp, window_bin), so that the 64 lanes of a K4 wave sweep nearly
//                       the same cells and fetch their task with one coalesced load.
//   K4a band_kernel     reads whose candidates all sit on <= 16 neighbouring diagonals -- row-m cells of an
//                       adapter inside the read, or last-column cells of a partial adapter at the read end:
//                       row-major banded DP (filter_core.hpp, band_locate / band_locate_last).
//   K4  window_kernel   everything else: the packed-word DP (locate_core.hpp, window mode) over
//                       each read's window only; reads are gathered by index (16-byte chunk
//                       loads).
//
// Blocks of K1/K3 own contiguous tile ranges, so K3 needs no global atomics: its offsets
// come from K2's scan and an LDS cursor per bin.
#ifndef ATR_LOCATE_FAST_HPP
#define ATR_LOCATE_FAST_HPP

#include <algorithm>
#include "locate_kernel.hpp"
#include "filter_core.hpp"
#include "linked_core.hpp"
#include "piece_core.hpp"
#include "fast_work.hpp"

namespace atr {
// filter_params() of a handle, computed once per state of the handle and host thread (a four-entry cache keyed by
// atr_aligner::uid): the certificates in it are small dynamic programmes over the adapter (filter_core.hpp) -- 0.15 ms
// of host time that every call of a 1 000-read batch paid again (round 5: 39 -> 176 us per call until this cache).
inline const FilterParams &aligner_filter_params(const atr_aligner *a) {
    struct Slot { unsigned long long uid = 0; FilterParams fp; };
    static thread_local Slot slots[4];
    static thread_local unsigned turn = 0;
    if (a->uid != 0)
        for (Slot &s : slots) if (s.uid == a->uid) return s.fp;
    Slot &s = slots[turn++ & 3u];
    s.fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap);
    s.uid = a->uid;
    return s.fp;
}
// ... and the parameters the two-pass pre-pass works with (filter_params(.., planes_path): extended NARROW mode for adapters
// of 41 .. 64 bases), cached the same way
inline const FilterParams &aligner_piece_filter_params(const atr_aligner *a) {
    struct Slot { unsigned long long uid = 0; FilterParams fp; };
    static thread_local Slot slots[4];
    static thread_local unsigned turn = 0;
    if (a->uid != 0)
        for (Slot &s : slots) if (s.uid == a->uid) return s.fp;
    Slot &s = slots[turn++ & 3u];
    s.fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap, true);
    s.uid = a->uid;
    return s.fp;
}
}  // namespace atr

namespace atr {

// Extra arguments of the band / window kernels when they finish the 3' part of ONE adapter of a
// linked set (linked_core.hpp): its bins start at bin0, the read's alignment starts at front[r]'s
// querystop, the record goes through Adapter.match_to's acceptance test and is re-based.
struct LinkedArgs {
    int bin0;
    const uint4 *front;
    LinkedPost post;
    const void *multi;                               // host side only: the set's device blob for linked_band_kernel (or null:
    bool multi_and;                                  //   one band_kernel launch per adapter), its compare mode
    // window_kernel<.., LINKED> reads its adapter's parameters from the set's device blob (scalar loads): block row y of
    // the launch serves adapter a0 + y, so that the window reads of every adapter of a set whose 3' parts share the
    // kernel's template parameters go out in ONE launch (win_count rows; host side: 0 = this adapter rides with adapter 0)
    const LocateParams *multi_p;
    const LinkedPost *multi_post;
    int a0, win_count;
};

// blocks a batch of ntiles tiles keeps busy with four waves each (K1 / K3 of a short batch are launched with these)
inline int fast_blocks_for(long long ntiles) {
    return (int)std::max<long long>(1, std::min<long long>(FAST_BLOCKS, (ntiles + 3) / 4));
}

// Tasks per wave of a DP kernel (K4a / K4) for a SHORT batch: the smallest power of two that still gives every
// wave of the grid work.  A wave sweeps the union of its tasks' windows, one task per lane: a short batch leaves
// most SIMDs idle, so fewer lanes per wave mean shorter sweeps on more SIMDs (1000 reads: 101 -> 89 us per call).
// Not for long batches -- there the chip is VALU-bound and more, emptier waves only add instructions (tried on
// C2's 48 k window reads: 0.92 -> 1.06 ms).
__device__ __forceinline__ int dp_lanes_per_wave(long long tasks, long long grid_waves, int fixed) {
    if (fixed) return fixed;
    int lpw = 1;
    while (lpw < 64 && (long long)lpw * grid_waves < tasks) lpw <<= 1;
    return lpw;
}

constexpr int SCAN_CHUNK = 64, SCAN_CHUNKS = FAST_BLOCKS / SCAN_CHUNK;       // two-level scan of the histogram (K2a / K2b)
static_assert(FAST_BLOCKS % SCAN_CHUNK == 0 && SCAN_CHUNKS % 16 == 0 && SCAN_CHUNK % 16 == 0, "scan kernels walk 16 rows at a time");
// first slot of (bin, block) in `order`, for the scatter kernels
__device__ __forceinline__ uint32_t fast_slot0(const FastWork &wk, int bin) {
    return wk.binbase[bin] + wk.chunks[(size_t)(blockIdx.x / SCAN_CHUNK) * wk.nbins + bin] +
           wk.counts[(size_t)blockIdx.x * wk.nbins + bin];
}

#ifdef ATR_DEFINE_FILTER_KERNELS   // K1..K3 are defined once, in filter_kernels.hip
// WIDE: adapter longer than 32 bases.  RAGGED: reads of different lengths in the batch
// (a lane stops updating its state after its own last column).
template <bool WIDE, bool RAGGED>
__global__ __launch_bounds__(256) void filter_kernel(const LocateParams p, const FilterParams fp,
                                                     const uint4 *__restrict__ packed,
                                                     const int32_t *__restrict__ lens, long long nreads, int nchunks,
                                                     int max_len, uint4 *__restrict__ out, FastWork wk) {
    __shared__ uint2 s_peq[16];
    __shared__ uint32_t s_hist[FILTER_BINS];
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    if (threadIdx.x < 16) s_peq[threadIdx.x] = make_uint2((uint32_t)fp.peq[threadIdx.x], (uint32_t)(fp.peq[threadIdx.x] >> 32));
    if (threadIdx.x < FILTER_BINS) s_hist[threadIdx.x] = 0;
    __syncthreads();

    // readfirstlane: the wave index is uniform, and this is how the compiler gets to know (scalar loop control)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    for (long long tile = t0 + wave; tile < t1; tile += 4) {
        const long long r = tile * 64 + lane;
        const bool live = r < nreads;
        const int n = live ? (RAGGED ? lens[r] : max_len) : 0;
        FilterState F;
        filter_init(F, u, fp.rows);
        const int jhi = RAGGED ? wave_max_i32(n) : max_len;
        const int jfull = RAGGED ? wave_min_i32(live ? n : 0x7fffffff) : max_len;     // columns every live lane has
        const uint4 *tp = packed + (size_t)tile * nchunks * 64 + lane;
        if (jhi > 0) {
            const int c1 = (jhi + 31) >> 5;
            uint4 nxt = tp[0];
            // The eight match masks of a dword are fetched from LDS ONE DWORD AHEAD of the (serially
            // dependent) column updates that use them: two sets of eight, swapped every dword.
            uint2 ea[8], eb[8];
            auto fetch_masks = [&](uint32_t w, uint2 (&e)[8]) { fetch_peq8(s_peq, w, e); };
            const uint32_t kreg = (uint32_t)u.k;
            fetch_masks(nxt.x, ea);
            for (int c = 0; c < c1; ++c) {
                uint4 cur = nxt;
                if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
                int j = c * 32;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint2 (&e)[8] = (d & 1) ? eb : ea;       // this dword's masks; the other set takes the next dword's
                    fetch_masks(d == 0 ? cur.y : d == 1 ? cur.z : d == 2 ? cur.w : nxt.x, (d & 1) ? ea : eb);
                    if (j + 8 <= (RAGGED ? jfull : jhi)) {   // wave-uniform: every lane of the wave owns all eight columns
#pragma unroll
                        for (int b = 0; b < 8; ++b) filter_step<WIDE>(F, e[b].x, e[b].y, kreg);
                        j += 8;
                    } else {
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            ++j;
                            if (j <= jhi && (!RAGGED || j <= n)) filter_step<WIDE>(F, e[b].x, e[b].y, kreg);
                        }
                    }
                    if (j >= jhi) break;                     // wave-uniform
                }
                filter_fold(F, RAGGED ? min(n, min(j, jhi)) : min(j, jhi), fp.rows, kreg);   // at most 32 columns since the last fold
            }
        }
        uint32_t ww_lane = 0u;
        if (live) {
            uint32_t rec[4];
            const uint32_t ww = filter_decide<WIDE>(F, u, fp, (const uint32_t *)tp, nchunks, n, rec);
            ww_lane = ww;
            if (!window_valid(ww)) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
            else atomicAdd(&s_hist[window_bin(ww, u.m, !RAGGED || ragged_rows_bins(u.sr))], 1u);
        }
        // Window words of the unresolved reads only, compacted to the front of the tile's 64 slots of `win`, plus the
        // tile's 64-bit mask of unresolved lanes: one line written (and read back by K3) per tile instead of four.
        const uint64_t um = __ballot(window_valid(ww_lane));
        if (lane == 0) wk.mask[tile] = um;
        if (window_valid(ww_lane)) wk.win[tile * 64 + __popcll(um & ((1ull << lane) - 1ull))] = ww_lane;
    }
    __syncthreads();
    // counts are stored block-major ([block][bin]): one coalesced run per block here and in K3
    if (threadIdx.x < FILTER_BINS) wk.counts[(size_t)blockIdx.x * FILTER_BINS + threadIdx.x] = s_hist[threadIdx.x];
}

// K2a / K2b: exclusive scan over the FAST_BLOCKS blocks of every bin's per-block count, in two levels.
// The counts are stored block-major ([block][bin]: K1 writes and K3 reads its row as one coalesced run -- the
// bin-major layout of round 1 cost a 4-byte store into a line of its own per (bin, block), 60 MB of counted
// writes per call).  K2a: one thread per bin and chunk of SCAN_CHUNK blocks walks its column (coalesced rows,
// loads issued sixteen at a time), leaves the offsets inside the chunk in place and the chunk's total in
// `chunks`.  K2b: one thread per bin scans the SCAN_CHUNKS chunk totals (in place: chunk offsets inside the bin),
// then the bin totals are scanned across the block -> bin bases.  K3 adds the three levels up.
__global__ __launch_bounds__(256) void scan_bins_kernel(FastWork wk) {
    const int bin = blockIdx.y * 256 + threadIdx.x;
    uint32_t *col = wk.counts + (size_t)blockIdx.x * SCAN_CHUNK * wk.nbins + bin;
    const int rows = min(SCAN_CHUNK, wk.nused - (int)blockIdx.x * SCAN_CHUNK);      // histogram rows of this chunk that exist
    uint32_t run = 0;
    for (int b0 = 0; b0 < rows; b0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = b0 + i < rows ? col[(size_t)(b0 + i) * wk.nbins] : 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (b0 + i < rows) col[(size_t)(b0 + i) * wk.nbins] = run; run += v[i]; }
    }
    wk.chunks[(size_t)blockIdx.x * wk.nbins + bin] = run;
}

__global__ __launch_bounds__(1024) void scan_total_kernel(FastWork wk) {
    __shared__ uint32_t s_part[1024];
    const int b = threadIdx.x;
    uint32_t mine = 0u;
    if (b < wk.nbins) {
        uint32_t *col = wk.chunks + b;
        const int chunks = (wk.nused + SCAN_CHUNK - 1) / SCAN_CHUNK;          // chunk totals K2a wrote
        for (int c0 = 0; c0 < chunks; c0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = c0 + i < chunks ? col[(size_t)(c0 + i) * wk.nbins] : 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) { if (c0 + i < chunks) col[(size_t)(c0 + i) * wk.nbins] = mine; mine += v[i]; }
        }
    }
    s_part[b] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // Hillis-Steele inclusive scan of the bin totals
        const uint32_t add = b >= off ? s_part[b - off] : 0u;
        __syncthreads();
        s_part[b] += add;
        __syncthreads();
    }
    if (b < wk.nbins) wk.binbase[b] = s_part[b] - mine;
    if (b == 1023) { wk.binbase[wk.nbins] = s_part[1023]; wk.total[0] = s_part[1023]; }
}

__global__ __launch_bounds__(256) void scatter_kernel(long long nreads, int m, int by_rows, FastWork wk) {
    __shared__ uint32_t s_cur[FILTER_BINS];
    if (threadIdx.x < FILTER_BINS)
        s_cur[threadIdx.x] = fast_slot0(wk, threadIdx.x);
    __syncthreads();
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    // a wave takes one tile per round; the loads of SCATTER_ROUNDS rounds are issued together (the block is a chain of
    // dependent round trips otherwise: mask -> word -> LDS cursor -> store, 34 us for 30 MB of traffic)
    constexpr int SCATTER_ROUNDS = 4;
    const long long rend = min(nreads, t1 * 64);
    for (long long r0 = t0 * 64 + threadIdx.x; r0 < rend; r0 += 256 * SCATTER_ROUNDS) {
        uint64_t um[SCATTER_ROUNDS];
        uint32_t ww[SCATTER_ROUNDS];
#pragma unroll
        for (int t = 0; t < SCATTER_ROUNDS; ++t) {
            const long long r = r0 + 256 * t;
            um[t] = r < rend ? wk.mask[r >> 6] : 0ull;
        }
#pragma unroll
        for (int t = 0; t < SCATTER_ROUNDS; ++t) {
            const long long r = r0 + 256 * t;
            const int lane = (int)(r & 63);
            ww[t] = ((um[t] >> lane) & 1ull) ? wk.win[(r & ~63ll) + __popcll(um[t] & ((1ull << lane) - 1ull))] : 0u;
        }
#pragma unroll
        for (int t = 0; t < SCATTER_ROUNDS; ++t) {
            const long long r = r0 + 256 * t;
            if (window_valid(ww[t]))
                wk.order[atomicAdd(&s_cur[window_bin(ww[t], m, by_rows != 0)], 1u)] = make_uint2((uint32_t)r, ww[t]);
        }
    }
}
// K4a: the banded DP over the band reads = the slots [0, binbase[BAND_BINS]) of `order`
// (persistent grid like K4; reads gathered by index).
template <bool AND_MODE, bool LINKED, bool PLANES = false>
__global__ __launch_bounds__(256) void band_kernel(const LocateParams p, const BandParams bp,
                                                   const uint4 *__restrict__ packed, const int32_t *__restrict__ lens,
                                                   long long nreads, int nchunks, int max_len, uint4 *__restrict__ out,
                                                   FastWork wk, const LinkedArgs la) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_stream[4][BAND_STREAM][64];              // per wave: the staged reads, [dword][lane]
    __shared__ uint32_t s_spread_flat[PLANES ? 1024 : 1];          // plane64 reads: bytes of a plane -> nibbles (piece_core.hpp)
    uint32_t (*s_spread)[256] = (uint32_t (*)[256])s_spread_flat;
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    for (int i = threadIdx.x; i <= u.m + 1; i += 256) s_thr[i] = p.thr[i];
    if constexpr (PLANES) piece_spread_fill(s_spread);
    __syncthreads();
    // band reads come first in `order` (linked: first among the bins of this adapter)
    const long long base = LINKED ? (long long)wk.binbase[la.bin0] : 0;
    const long long total = (long long)wk.binbase[(LINKED ? la.bin0 : 0) + BAND_BINS];
    const int lane = threadIdx.x & 63;
    const int lpw = dp_lanes_per_wave(total - base, (long long)gridDim.x * 4, wk.lpw);
    const long long nwaves = (total - base + lpw - 1) / lpw;
    for (long long wv = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); wv < nwaves;
         wv += (long long)gridDim.x * 4) {
        const long long slot = base + wv * lpw + lane;
        const bool live = lane < lpw && slot < total;
        const uint2 task = live ? wk.order[slot] : make_uint2(0u, 0u);
        const long long r = (long long)task.x;
        const uint32_t ww = task.y;
        const int n = live ? (lens ? lens[r] : max_len) : 0;
        // two kinds of band reads (filter_core.hpp): row-m candidates on <= 16 diagonals, and last-column candidates
        // (scan bit; possibly with row-m candidates besides); their bins are disjoint, so only a wave that straddles
        // the boundary runs both sweeps
        const bool last = live && window_scan(ww);
        const bool any_last = wave_max_i32(last ? 1 : 0) != 0, any_rowm = wave_max_i32(live && !last ? 1 : 0) != 0;
        const int s_lane = (live && !last) ? window_hi(ww) - u.m + u.k - window_lo(ww) : 0;
        const int smax = min(BAND_W - 1, wave_max_i32(s_lane));
        const uint32_t *q = (const uint32_t *)(packed + ((size_t)(r >> 6) * nchunks) * 64 + (r & 63));
        uint32_t *ns = &s_stream[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))][0][lane];
        if constexpr (PLANES) {
            // the 64 codes from the first diagonal on: the record the pre-pass wrote (two coalesced 16-byte loads per
            // lane), or -- entries of the full-sweep kernel -- gathered from the plane64 batch
            const bool dense = live && (ww & PIECE_NODENSE) == 0u;
            if (dense) {
                const uint32_t di = wk.dref[slot];
                const uint4 a = wk.tdata[2 * (size_t)di], b = wk.tdata[2 * (size_t)di + 1];
                ns[0] = a.x; ns[64] = a.y; ns[128] = a.z; ns[192] = a.w;
                ns[256] = b.x; ns[320] = b.y; ns[384] = b.z; ns[448] = b.w;
            }
            if (wave_max_i32(live && !dense ? 1 : 0) != 0) {
                if (!dense) band_stage_planes(q, nchunks, window_lo(ww), ns, 64, s_spread, band_stream_dwords(u.m));
            }
        } else band_stage(q, nchunks, window_lo(ww), ns, 64, band_stream_dwords(u.m));
        uint32_t rec[4] = {0xFFFF0000u, 0u, 0u, 0u};
        if (any_rowm) band_locate<AND_MODE>(u, bp.rrep, bp.noindel != 0, ns, 64, n, ww, smax, s_thr, rec);
        if (any_last) {
            const int smax_l = min(BAND_W - 1, wave_max_i32(last ? last_band_width(ww) : 0));
            const int rows_max = wave_max_i32(last ? (last_band_rowm(ww) ? u.m : window_rows(ww)) : 0);
            const int cap_lo = wave_min_i32(last ? window_rows(ww) - last_band_span(ww) : 0x7fffffff);
            uint32_t rec_l[4];
            band_locate_last<AND_MODE>(u, bp.rrep, bp.noindel != 0, ns, 64, n, ww, last, smax_l, rows_max, cap_lo, s_thr, rec_l);
            if (last) { rec[0] = rec_l[0]; rec[1] = rec_l[1]; rec[2] = rec_l[2]; rec[3] = rec_l[3]; }
        }
        if (LINKED && live)
            linked_finish(rec, (int)(la.front[r].y >> 16), la.post.m, la.post.min_overlap, la.post.pf_thr,
                          la.post.accept_full != 0, la.post.rmp, la.post.rmp_ld, la.post.max_rmp);
        if (live) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}
// Anchored 5' adapters: the banded DP over all reads of the batch, no pre-pass
// (filter_core.hpp, band_locate_prefix).  One read per lane, tile64 order.
template <bool AND_MODE>
__global__ __launch_bounds__(256) void prefix_band_kernel(const LocateParams p, const BandParams bp,
                                                          const uint4 *__restrict__ packed,
                                                          const int32_t *__restrict__ lens, long long nreads,
                                                          int nchunks, int max_len, uint4 *__restrict__ out) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_stream[4][BAND_STREAM][64];
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    for (int i = threadIdx.x; i <= u.m + 1; i += 256) s_thr[i] = p.thr[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile * 64 >= nreads) return;
    const long long r = tile * 64 + lane;
    const bool live = r < nreads;
    const int n = live ? (lens ? lens[r] : max_len) : 0;
    const uint32_t *q = (const uint32_t *)(packed + (size_t)tile * nchunks * 64 + lane);
    uint32_t *ns = &s_stream[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))][0][lane];
    band_stage(q, nchunks, -u.k, ns, 64, band_stream_dwords(u.m));
    uint32_t rec[4];
    band_locate_prefix<AND_MODE>(u, bp.rrep, bp.noindel != 0, ns, 64, n, s_thr, rec);
    if (live) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}
#endif  // ATR_DEFINE_FILTER_KERNELS

// the aligner parameters of a window launch: the kernel argument, or -- linked sets -- the adapter's block of the set's
// device blob (LinkedArgs)
template <bool LINKED>
__device__ __forceinline__ const LocateParams &window_params(const LocateParams &p_arg, const LinkedArgs &la, int ad) {
    if constexpr (LINKED) return la.multi_p[ad];
    else return p_arg;
}

template <int MT, bool NOINDEL, bool LINKED, bool PLANES = false>
__global__ __launch_bounds__(256) void window_kernel(const LocateParams p_arg, const uint4 *__restrict__ packed,
                                                     const int32_t *__restrict__ lens, long long nreads,
                                                     int nchunks, int max_len, uint4 *__restrict__ out, FastWork wk,
                                                     const LinkedArgs la) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_init[ATR_MAX_REF_LEN + 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_nm[16][4];
    __shared__ uint32_t s_spread_flat[PLANES ? 1024 : 1];          // plane64 reads
    uint32_t (*s_spread)[256] = (uint32_t (*)[256])s_spread_flat;
    const int ad = LINKED ? la.a0 + (int)blockIdx.y : 0;          // linked set: the adapter this block row serves
    const LocateParams &p = window_params<LINKED>(p_arg, la, ad);
    const Uniform u = make_uniform(p, MT);
    if (threadIdx.x < 64) s_nm[threadIdx.x >> 2][threadIdx.x & 3] = p.nmask[threadIdx.x >> 2][threadIdx.x & 3];
    for (int i = threadIdx.x; i <= MT + 1; i += 256) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    if constexpr (PLANES) piece_spread_fill(s_spread);
    __syncthreads();
    // the slots [first, total) of `order`: everything but the band reads (K4a)
    const int bin0 = LINKED ? ad * FILTER_BINS : 0;
    const long long first = (long long)wk.binbase[bin0 + BAND_BINS], total = (long long)wk.binbase[bin0 + FILTER_BINS];
    const int lane = threadIdx.x & 63;
    const int lpw = dp_lanes_per_wave(total - first, (long long)gridDim.x * 4, wk.lpw);
    const long long nwaves = (total - first + lpw - 1) / lpw;
    // persistent grid: each wave takes every (gridDim*4)-th group of lpw slots of `order`
    for (long long wv = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); wv < nwaves;
         wv += (long long)gridDim.x * 4) {
        const long long slot = first + wv * lpw + lane;
        const bool live = lane < lpw && slot < total;
        const uint2 task = live ? wk.order[slot] : make_uint2(0u, 0u);
        const long long r = (long long)task.x;
        const uint32_t ww = task.y;
        const int n = live ? (lens ? lens[r] : max_len) : 0;
        const int j_lo = window_lo(ww), j_hi = live ? window_hi(ww) : 0;
        const bool has_window = live && j_hi > j_lo;
        const int jlo = wave_min_i32(live ? j_lo : 0x7fffffff);          // common start column
        const int jhi = wave_max_i32(has_window ? j_hi : 0);
        const int rows_w = wave_max_i32(live ? window_rows(ww) : 0);
        const int plimit = u.p0 + rows_w;                                // highest position any lane needs
        // a wave of by-rows bins only; not from a real column 0 of zeros (START_WITHIN_SEQ1), which
        // is no upper bound of the cells to its right
        const bool head = !(u.sr && jlo == 0);
        // (equal-length batches only: in a ragged one such a wave runs in tail mode below, or without the triangle)
        const bool tri = __builtin_amdgcn_readfirstlane((int)(first + wv * lpw >= (long long)wk.binbase[bin0 + ROWS_BIN0])) != 0 &&
                         head && lens == nullptr;       // (wave-uniform by construction; tell the compiler)
        // linked: the alignment of a lane starts at its own column s (= front.rstop).  The wave's
        // common start column may lie before that -- such a lane is re-initialised when the sweep
        // reaches its s (its own window starts at or after s, so nothing of value was computed).
        const int s_lane = (LINKED && live) ? (int)(la.front[r].y >> 16) : 0;
        const int s_top = LINKED ? wave_max_i32(s_lane) : 0;

        LaneState<MT> L;
        // this lane's read inside the tile64 layout (gathered: 16 bytes per lane per chunk)
        const uint4 *tp = packed + ((size_t)(r >> 6) * nchunks) * 64 + (r & 63);

        // TAIL MODE (ragged batches, waves of the row-count bins): every lane's window ends at its own read
        // end, so the sweep runs in columns counted from the END -- lane column j stands for the lane's own
        // column j - shift, shift = max_len - n -- which makes window, row limit and triangle wave-uniform
        // exactly as in an equal-length batch.  The lane's last TAIL_COLUMNS bases are fetched once, re-aligned
        // to the read end.  A lane whose read starts inside the swept range (column shift) restarts there from
        // the initial column, which without START_WITHIN_SEQ1 is the fresh-window column itself.
        const bool rows_wave = __builtin_amdgcn_readfirstlane((int)(first + wv * lpw >= (long long)wk.binbase[bin0 + ROWS_BIN0])) != 0;
        if (!LINKED && lens != nullptr && rows_wave && !u.sr) {
            const int shift = live ? max_len - n : 0;
            const int v0 = wave_min_i32(live ? j_lo + shift : 0x7fffffff);
            if (max_len - v0 <= TAIL_COLUMNS && v0 < max_len) {
                const int s_top_t = wave_max_i32(shift);
                lane_init_window<MT, NOINDEL>(L, u, max_len, v0, live ? max_len : 0, live && window_scan(ww), s_init, s_thr);
                uint32_t tb[TAIL_COLUMNS / 8];                         // bases n - 64 .. n - 1 (0-based), eight per dword
                {
                    const int dlo = n - TAIL_COLUMNS;                  // may be negative: those bases read as code 0
                    const int z0 = dlo >> 3;
                    const uint32_t sh = 4u * (uint32_t)(dlo & 7);
                    uint32_t raw[TAIL_COLUMNS / 8 + 1];
#pragma unroll
                    for (int t = 0; t <= TAIL_COLUMNS / 8; ++t) {
                        if constexpr (PLANES) raw[t] = live ? read_dword_planes((const uint32_t *)tp, nchunks, z0 + t, s_spread) : 0u;
                        else raw[t] = live ? read_dword((const uint32_t *)tp, nchunks, z0 + t) : 0u;
                    }
#pragma unroll
                    for (int t = 0; t < TAIL_COLUMNS / 8; ++t) tb[t] = sh ? ((raw[t] >> sh) | (raw[t + 1] << (32u - sh))) : raw[t];
                }
                int j = max_len - TAIL_COLUMNS;                        // column of the base before tb's first
#pragma unroll 1
                for (int d = 0; d < TAIL_COLUMNS / 8; ++d) {
                    uint32_t w = tb[0];
#pragma unroll
                    for (int t = 0; t + 1 < TAIL_COLUMNS / 8; ++t) tb[t] = tb[t + 1];
#pragma unroll 1
                    for (int b = 0; b < 8; ++b) {
                        ++j;
                        const uint32_t q = w & 15u;
                        w >>= 4;
                        if (j <= v0) continue;
                        uint32_t nm[(MT + 31) / 32];
                        load_mask(nm, s_nm, q);
                        int pl = min(plimit, u.p0 + (j - v0) + u.k);
                        pl = min(pl, u.p0 + triangle_rows(rows_w, max_len, j, u.k));
                        lane_step<MT, NOINDEL, true, true>(L, u, j, nm, s_thr, pl);
                        if (j <= s_top_t && shift == j) lane_restart_window<MT>(L, u, j);
                    }
                }
                if (live) {
                    uint32_t rec[4];
                    lane_result<MT>(L, u, rec);
                    if ((rec[0] >> 16) != 0xFFFFu) rec[1] -= (uint32_t)shift * 0x00010001u;      // back to the read's own columns
                    out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
                }
                continue;
            }
        }
        lane_init_window<MT, NOINDEL>(L, u, n, jlo, j_hi, live && window_scan(ww), s_init, s_thr);
        if (jhi > jlo) {
            const int c0 = jlo >> 5, c1 = (jhi + 31) >> 5;
            uint4 nxt = tp[(size_t)c0 * 64];
            for (int c = c0; c < c1; ++c) {
                uint4 cur = nxt;
                if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
                if constexpr (PLANES) cur = make_uint4(piece_nibbles(s_spread, cur.x, cur.y, cur.z, cur.w, 0), piece_nibbles(s_spread, cur.x, cur.y, cur.z, cur.w, 1),
                                             piece_nibbles(s_spread, cur.x, cur.y, cur.z, cur.w, 2), piece_nibbles(s_spread, cur.x, cur.y, cur.z, cur.w, 3));
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
                        uint32_t nm[(MT + 31) / 32];
                        load_mask(nm, s_nm, q);
                        // rows above (j - jlo) + k cannot lie on a path that left row 0 at a column >= jlo with at
                        // most k insertions (same stale-cell argument as the triangle at the read end)
                        int pl = head ? min(plimit, u.p0 + (j - jlo) + u.k) : plimit;
                        if (tri) pl = min(pl, u.p0 + triangle_rows(rows_w, max_len, j, u.k));
                        lane_step<MT, NOINDEL, true, true>(L, u, j, nm, s_thr, pl);
                        if (LINKED && j <= s_top && s_lane == j) lane_restart_window<MT>(L, u, j);
                    }
                }
            }
        }
        if (live) {
            uint32_t rec[4];
            lane_result<MT>(L, u, rec);
            if constexpr (LINKED) {
                const LinkedPost &post = la.multi_post[ad];
                linked_finish(rec, s_lane, post.m, post.min_overlap, post.pf_thr, post.accept_full != 0,
                              post.rmp, post.rmp_ld, post.max_rmp);
            }
            out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        }
    }
}

// la == nullptr: the single-aligner pipeline
typedef int (*window_launcher)(const atr_aligner *, const uint4 *, const int32_t *, long long, int, int, uint4 *,
                               FastWork, const LinkedArgs *, hipStream_t, bool);

template <int MT>
int launch_window_mt(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                     int max_len, uint4 *out, FastWork wk, const LinkedArgs *la, hipStream_t st, bool planes) {
    const bool noindel = a->indel_cost > a->p.k;
    // (one block of four waves per 4 reads at the low end: a short batch gets a wave per task)
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((nreads + 3) / 4, 4096))), block(256);
    if (la) {
        if (la->win_count < 1) return 0;                         // (rides with adapter 0's launch)
        const dim3 lgrid(grid.x, (unsigned)la->win_count);
        if (noindel) hipLaunchKernelGGL((window_kernel<MT, true, true>), lgrid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, *la);
        else         hipLaunchKernelGGL((window_kernel<MT, false, true>), lgrid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, *la);
    } else if (planes) {                                         // plane64 reads (two-pass pre-pass, piece_kernels.hip)
        LinkedArgs none;
        memset(&none, 0, sizeof(none));
        if (noindel) hipLaunchKernelGGL((window_kernel<MT, true, false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, none);
        else         hipLaunchKernelGGL((window_kernel<MT, false, false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, none);
    } else {
        LinkedArgs none;
        memset(&none, 0, sizeof(none));
        if (noindel) hipLaunchKernelGGL((window_kernel<MT, true, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, none);
        else         hipLaunchKernelGGL((window_kernel<MT, false, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out, wk, none);
    }
    return (int)hipGetLastError();
}

constexpr int WINDOW_SIZES = FILTER_MAX_M / ROW_GRAN;        // MT = 4 .. 64

}  // namespace atr
#endif
