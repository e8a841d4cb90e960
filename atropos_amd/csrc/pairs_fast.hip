// pairs_fast.hip -- the fast pipeline of atr_locate_pairs_batch (pairs_fast_core.hpp):
//
//   P1  pairs_myers_kernel   one pair per lane: per-lane match masks in LDS, Myers' bit-vector sweep of the whole
//                            matrix (the exact cost of every row-m and last-column cell), threat analysis; writes
//                            the record of pairs without any acceptable alignment, a 16-byte task for the rest
//                            and its scatter bin (band class x rows swept; one bin for the full sweep).
//   P2  scan kernels         the (bin, block) histogram -> offsets (scan_bins_kernel / scan_total_kernel of the
//                            single-aligner pipeline, locate_fast.hpp).
//   P3  pairs_scatter_kernel tasks ordered by bin: the 64 lanes of a P4 wave sweep bands of one width and nearly
//                            the same number of rows.
//   P4  pairs_band_kernel<WB> the banded packed-word DP (payload of the threats), one launch per band class.
//   P5  the full sweep (pairs_core.hpp) over the pairs of the last bin, gathered by index.
//
// VALU-bound like every DP here (no MFMA: a min-plus recurrence).  LDS: P1 (5 + 2) x NW words per lane; P4 none
// (the lane's reference / query streams are read from the batch as the sweep goes).
#include <hip/hip_runtime.h>
#include <algorithm>

#include "atropos_hip.h"
#include "locate_fast.hpp"
#include "side_stream.hpp"
#include "pairs_fast_core.hpp"

namespace atr {

int hip_fail(hipError_t e, const char *what);             // api.hip
__global__ void scan_bins_kernel(FastWork wk);            // filter_kernels.hip
__global__ void scan_total_kernel(FastWork wk);

struct PairsFastWork {
    FastWork fw;                                    // counts / chunks / binbase (nbins = PF_BINS); win, mask, order unused
    uint4 *tasks;                                   // [npairs] PairTask per pair (P1)
    uint4 *order;                                   // [npairs] tasks by bin (P3)
    uint8_t *bins;                                  // [npairs] scatter bin, 0xFF: resolved in P1
};

struct PairsFastArgs {
    const uint32_t *ref_packed, *qry_packed;
    const int32_t *ref_lens, *qry_lens, *need;
    int ref_chunks, ref_max_len, revcomp, qry_chunks, qry_max_len;
    long long npairs;
    uint4 *out;
};

// tiles [t0, t1) of a block of P1 / P3: contiguous ranges of a whole number of wave rounds (a short batch then keeps
// all waves of its blocks busy instead of one wave in each of FAST_BLOCKS blocks); the blocks behind the last tile
// write empty histogram rows
__device__ __forceinline__ void pairs_block_tiles(long long ntiles, int nwaves, long long &t0, long long &t1) {
    long long per = (ntiles + FAST_BLOCKS - 1) / FAST_BLOCKS;
    per = (per + nwaves - 1) / nwaves * nwaves;
    t0 = min(ntiles, per * (long long)blockIdx.x);
    t1 = min(ntiles, t0 + per);
}

// ---- P1 ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void pairs_myers_kernel(const PairFastParams fp, const PairsFastArgs a, PairsFastWork wk) {
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3], s_gap[PAIRS_MAX_LEN + 2], s_gas[PAIRS_MAX_LEN + 2];
    __shared__ uint32_t s_hist[PF_BINS];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6;     // 4 waves, fewer when the LDS of long reads demands it
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 3; i += nthreads) s_thr[i] = fp.pp.thr[i];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 2; i += nthreads) { s_gap[i] = fp.g_ap[i]; s_gas[i] = fp.g_as[i]; }
    for (int i = threadIdx.x; i < PF_BINS; i += nthreads) s_hist[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool sr = (fp.pp.flags & ATR_START_WITHIN_SEQ1) != 0, sq = (fp.pp.flags & ATR_START_WITHIN_SEQ2) != 0;
    // per wave: PF_TAB_ROWS x NW mask words per lane, and row m's horizontal deltas (two words per 32 columns: the
    // +1 bits and the -1 bits) from which pass 2 replays the row-m costs -- no candidate list (pairs_fast_core.hpp)
    constexpr int WAVE_WORDS = (PF_TAB_ROWS + 2) * NW * 64;
    uint32_t *tab = s_dyn + (size_t)wave * WAVE_WORDS + lane;
    uint32_t *hd = tab + (size_t)PF_TAB_ROWS * NW * 64;
    const long long ntiles = (a.npairs + 63) >> 6;
    long long t0, t1;
    pairs_block_tiles(ntiles, nwaves, t0, t1);
    const int rndw = a.ref_chunks * 4;
    const bool er = (fp.pp.flags & ATR_STOP_WITHIN_SEQ1) != 0;
    for (long long tile = t0 + wave; tile < t1; tile += nwaves) {
        const long long r = tile * 64 + lane;
        const bool live = r < a.npairs;
        const int m = live ? min(a.ref_lens ? a.ref_lens[r] : a.ref_max_len, a.ref_max_len) : 0;
        const int n = live ? min(a.qry_lens ? a.qry_lens[r] : a.qry_max_len, a.qry_max_len) : 0;
        const uint32_t *rp = a.ref_packed + ((size_t)tile * a.ref_chunks * 64 + lane) * 4;
        const uint4 *qt = (const uint4 *)a.qry_packed + (size_t)tile * a.qry_chunks * 64 + lane;
        bool known = pf_build_masks<NW>(tab, 64, rp, rndw, m, a.revcomp != 0, sq);
        const int k = m >= 1 ? (int)s_thr[m] : 0;
        const int n_sweep = sq ? n : min(n, m + k);                   // _align.pyx:314-321
        const int jhi = wave_max_i32(n_sweep);
        PfMyers<NW> S;
        pf_myers_init<NW>(S, m, sr);
        const uint32_t hin = sq ? 0u : 1u;
        const int need = a.need && live ? a.need[r] : 1;
        PfScan T;
        pf_stream_init(T, need);
        uint32_t chunks_hit = 0u;                                      // 32-column chunks with a row-m candidate
        const int c1 = jhi > 0 ? (jhi + 31) >> 5 : 0;
        if (jhi > 0) {
            uint4 nxt = qt[0];
            // the match masks of a column are fetched from LDS one column ahead of the (serially dependent) update
            uint32_t eqn[NW];
            {
                const uint32_t row0 = min(pf_code_rows8(nxt.x) & 7u, (uint32_t)(PF_TAB_ROWS - 1));
#pragma unroll
                for (int x = 0; x < NW; ++x) eqn[x] = tab[(size_t)(row0 * NW + x) * 64];
            }
            for (int c = 0; c < c1; ++c) {
                const uint4 cur = nxt;
                if (c + 1 < c1) nxt = qt[(size_t)(c + 1) * 64];
                int j = c * 32;
                uint32_t pbits = 0u, mbits = 0u;
                bool hit = false;
#pragma unroll 1
                for (int d = 0; d < 4; ++d) {
                    const uint32_t w = d == 0 ? cur.x : d == 1 ? cur.y : d == 2 ? cur.z : cur.w;
                    const uint32_t wn = d == 0 ? cur.y : d == 1 ? cur.z : d == 2 ? cur.w : nxt.x;     // the dword after
                    known = known && pf_codes_known(w);
                    // table rows of the bases of columns j + 2 .. j + 9 (the prefetch runs one column ahead)
                    uint32_t rows = (pf_code_rows8(w) >> 4) | (pf_code_rows8(wn) << 28);
#pragma unroll                                                    // (eight columns: 1.49 -> 1.41 ms per 2 M pairs 2 x 150)
                    for (int b = 0; b < 8; ++b) {
                        ++j;
                        if (j > jhi) break;                           // wave-uniform
                        uint32_t eq[NW];
#pragma unroll
                        for (int x = 0; x < NW; ++x) eq[x] = eqn[x];
                        const uint32_t row = min(rows & 7u, (uint32_t)(PF_TAB_ROWS - 1));
                        rows >>= 4;
#pragma unroll
                        for (int x = 0; x < NW; ++x) eqn[x] = tab[(size_t)(row * NW + x) * 64];
                        if (j <= n_sweep) {
                            const int before = S.score;
                            pf_myers_step<NW>(S, eq, hin);
                            const uint32_t bit = 1u << ((j - 1) & 31);
                            pbits |= S.score > before ? bit : 0u;
                            mbits |= S.score < before ? bit : 0u;
                            hit = pf_rowm_pass1(T, m, j, S.score, k, (int)s_gap[j], (int)s_gas[j], fp.pp.min_overlap) || hit;
                        }
                    }
                    if (j >= jhi) break;
                }
                hd[(size_t)(2 * c) * 64] = pbits;
                hd[(size_t)(2 * c + 1) * 64] = mbits;
                if (__any(hit)) chunks_hit |= 1u << c;                 // (wave-uniform; c < NW <= 10)
            }
        }
        // the bases behind n_sweep (only without START_WITHIN_SEQ2) are never compared; they need no code check
        const bool scan_last = n_sweep == n;
        const bool lastcol = live && scan_last && m >= 1 && n >= 1;
        if (lastcol)
            pf_stream_lastcol<NW>(T, S.pv, S.mv, m, n, sq, er, s_thr, (int)s_gap[n], (int)s_gas[n], fp.pp.min_overlap, need, false);
        {   // pass 2 over row m: the costs replayed from the horizontal deltas, chunks without a candidate skipped
            const int mlb = pf_mlb_eff(T, need);
            int score = sr ? 0 : m;                                    // D[m][0] (pf_myers_init)
            for (int c = 0; c < c1; ++c) {
                uint32_t pb = hd[(size_t)(2 * c) * 64], mb = hd[(size_t)(2 * c + 1) * 64];
                if (!((chunks_hit >> c) & 1u)) {
                    score += __builtin_popcount(pb) - __builtin_popcount(mb);
                    continue;
                }
#pragma unroll ATR_PF_WALK_UNROLL
                for (int b = 0; b < 32; ++b) {
                    const int j = 32 * c + b + 1;
                    if (j > jhi) break;                               // wave-uniform
                    score += (int)(pb & 1u) - (int)(mb & 1u);
                    pb >>= 1; mb >>= 1;
                    if (j <= n_sweep) pf_rowm_pass2(T, m, j, score, k, (int)s_gap[j], fp.pp.min_overlap, mlb);
                }
            }
        }
        if (lastcol)
            pf_stream_lastcol<NW>(T, S.pv, S.mv, m, n, sq, er, s_thr, (int)s_gap[n], (int)s_gas[n], fp.pp.min_overlap, need, true);
        PfDecision D;
        D.kind = 2; D.cls = 0;
        const bool ok = live && known && m >= 1 && n >= 1 && k <= PF_MAX_K;
        if (ok) pf_decide(T, m, n, need, (uint32_t)r, D);
        if (live) {
            if (D.kind == 0) {
                a.out[r] = make_uint4(0xFFFF0000u, 0u, 0u, 0u);
                wk.bins[r] = 0xFFu;
            } else {
                const int bin = pf_task_bin(D);
                if (D.kind != 1) {
                    D.task.pair = (uint32_t)r; D.task.d_lo = 0; D.task.row_first = 0; D.task.row_last = 0; D.task.mlb = 0;
                    D.task.cand_first = 0; D.task.reserved = 0;
                }
                wk.tasks[r] = make_uint4(D.task.pair, (uint32_t)(uint16_t)D.task.d_lo | ((uint32_t)(uint16_t)D.task.row_first << 16),
                                         (uint32_t)(uint16_t)D.task.row_last | ((uint32_t)(uint16_t)D.task.mlb << 16),
                                         (uint32_t)(uint16_t)D.task.cand_first);
                wk.bins[r] = (uint8_t)bin;
                atomicAdd(&s_hist[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PF_BINS; i += nthreads) wk.fw.counts[(size_t)blockIdx.x * PF_BINS + i] = s_hist[i];
}

// ---- P3 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairs_scatter_kernel(long long npairs, int p1_waves, PairsFastWork wk) {
    __shared__ uint32_t s_cur[PF_BINS];
    if (threadIdx.x < PF_BINS) s_cur[threadIdx.x] = fast_slot0(wk.fw, threadIdx.x);
    __syncthreads();
    const long long ntiles = (npairs + 63) >> 6;
    long long t0, t1;
    pairs_block_tiles(ntiles, p1_waves, t0, t1);                 // the tiles whose bins P1's block of the same index counted
    const long long rend = min(npairs, t1 * 64);
    for (long long r = t0 * 64 + threadIdx.x; r < rend; r += 256) {
        const uint32_t bin = wk.bins[r];
        if (bin != 0xFFu) wk.order[atomicAdd(&s_cur[bin], 1u)] = wk.tasks[r];
    }
}

// ---- P4 ------------------------------------------------------------------------------------------------
// Two waves per block; the lane's reference / query streams come straight from the batch (PfRefStream / PfQueryStream).
template <int WB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WB <= 32 ? 5 : WB <= 64 ? 4 : WB <= 96 ? 3 : 2)))
void pairs_band_kernel(const PairFastParams fp, const PairsFastArgs a, PairsFastWork wk, int cls) {
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3], s_gap[PAIRS_MAX_LEN + 2];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 3; i += 128) s_thr[i] = fp.pp.thr[i];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 2; i += 128) s_gap[i] = fp.g_ap[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long first = (long long)wk.fw.binbase[cls * PF_ROW_BINS], total = (long long)wk.fw.binbase[(cls + 1) * PF_ROW_BINS];
    const long long nwaves = (total - first + 63) >> 6;
    const bool sq = (fp.pp.flags & ATR_START_WITHIN_SEQ2) != 0;
    const int rndw = a.ref_chunks * 4, qndw = a.qry_chunks * 4;
    for (long long wv = (long long)blockIdx.x * 2 + wave; wv < nwaves; wv += (long long)gridDim.x * 2) {
        const long long slot = first + wv * 64 + lane;
        const bool live = slot < total;
        const uint4 task = live ? wk.order[slot] : make_uint4(0u, 0u, 0u, 0u);
        const long long r = (long long)task.x;
        PfBandLane L;
        L.live = live;
        L.d_lo = (int)(int16_t)(task.y & 0xFFFFu); L.row_first = live ? (int)(int16_t)(task.y >> 16) : 1;
        L.row_last = (int)(int16_t)(task.z & 0xFFFFu); L.mlb = (int)(int16_t)(task.z >> 16);
        L.cand_first = (int)(int16_t)(task.w & 0xFFFFu);
        L.m = live ? min(a.ref_lens ? a.ref_lens[r] : a.ref_max_len, a.ref_max_len) : 0;
        L.n = live ? min(a.qry_lens ? a.qry_lens[r] : a.qry_max_len, a.qry_max_len) : 0;
        const int k = L.m >= 1 ? (int)s_thr[L.m] : 0;
        L.n_sweep = sq ? L.n : min(L.n, L.m + k);
        L.scan_last = L.n_sweep == L.n;
        const int nrows = wave_max_i32(live ? L.row_last - L.row_first + 1 : 0);
        const uint32_t *rp = a.ref_packed + ((size_t)(r >> 6) * a.ref_chunks * 64 + (r & 63)) * 4;
        const uint32_t *qp = a.qry_packed + ((size_t)(r >> 6) * a.qry_chunks * 64 + (r & 63)) * 4;
        PfRefStream rs;
        PfQueryStream qs;
        rs.init(rp, live ? rndw : 0, L.m, a.revcomp != 0, L.row_first);
        qs.init(qp, live ? qndw : 0, L.row_first + L.d_lo - 1);
        uint32_t rec[4];
        pf_band_sweep<WB>(L, nrows, rs, qs, fp.pp, s_thr, s_gap, rec);
        if (live) a.out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

size_t pairs_fast_work_bytes(long long npairs) {
    return (size_t)npairs * 33 + (size_t)(FAST_BLOCKS + FAST_BLOCKS / 64) * PF_BINS * 4 + (size_t)(PF_BINS + 1) * 4 + 1024;
}

static PairsFastWork pairs_fast_carve(void *work, long long npairs) {
    PairsFastWork w;
    w.tasks = (uint4 *)work;
    w.order = w.tasks + npairs;
    uint32_t *p = (uint32_t *)(w.order + npairs);
    w.fw.win = nullptr; w.fw.mask = nullptr; w.fw.order = nullptr;
    w.fw.counts = p;
    w.fw.chunks = w.fw.counts + (size_t)FAST_BLOCKS * PF_BINS;
    w.fw.binbase = w.fw.chunks + (size_t)(FAST_BLOCKS / 64) * PF_BINS;
    w.fw.total = w.fw.binbase + PF_BINS + 1;
    w.fw.nbins = PF_BINS;
    w.fw.nused = FAST_BLOCKS;
    w.bins = (uint8_t *)(w.fw.total + 4);
    return w;
}

// the full sweep over the pairs of the last bin (pairs_kernel.hip)
hipError_t launch_pairs_full_indexed(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                                     const uint32_t *qp, const int32_t *ql, int qmax, long long npairs, uint4 *out,
                                     const uint4 *order, const uint32_t *range, hipStream_t st);

template <int WB>
static hipError_t launch_band(const PairFastParams &fp, const PairsFastArgs &a, const PairsFastWork &wk, int cls,
                              hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<long long>((a.npairs + 127) / 128, 2048);
    hipLaunchKernelGGL((pairs_band_kernel<WB>), dim3(blocks), dim3(128), 0, st, fp, a, wk, cls);
    return hipGetLastError();
}

// waves per block of P1
static int myers_waves(int nw) { (void)nw; return 4; }

template <int NW>
static hipError_t launch_myers(const PairFastParams &fp, const PairsFastArgs &a, const PairsFastWork &wk, hipStream_t st) {
    const size_t per_wave = (size_t)(PF_TAB_ROWS + 2) * NW * 64 * 4;
    const int waves = myers_waves(NW);
    const size_t lds = (size_t)waves * per_wave;
    hipError_t e = hipFuncSetAttribute((const void *)pairs_myers_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pairs_myers_kernel<NW>), dim3(FAST_BLOCKS), dim3(64 * waves), lds, st, fp, a, wk);
    return hipGetLastError();
}

// Returns hipSuccess when the whole batch went through the fast pipeline.
hipError_t launch_pairs_fast(const PairParams &p, double e_rate, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                             const uint32_t *qp, const int32_t *ql, int qmax, const int32_t *need, long long npairs,
                             uint4 *out, hipStream_t st) {
    static thread_local PairFastParams fp;
    static thread_local double fp_e = -1.0;
    if (fp_e != e_rate) { pairs_fast_tables(e_rate, fp); fp_e = e_rate; }
    fp.pp = p;
    void *work = nullptr;
    hipError_t e = hipMallocAsync(&work, pairs_fast_work_bytes(npairs), st);
    if (e != hipSuccess) return e;
    PairsFastWork wk = pairs_fast_carve(work, npairs);
    PairsFastArgs a;
    a.ref_packed = rp; a.qry_packed = qp; a.ref_lens = rl; a.qry_lens = ql; a.need = need;
    a.ref_chunks = (rmax + 31) / 32; a.ref_max_len = rmax; a.revcomp = revcomp;
    a.qry_chunks = (qmax + 31) / 32; a.qry_max_len = qmax; a.npairs = npairs; a.out = out;
    if (rmax <= 160) e = launch_myers<5>(fp, a, wk, st);
    else if (rmax <= 256) e = launch_myers<8>(fp, a, wk, st);
    else e = launch_myers<10>(fp, a, wk, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(scan_bins_kernel, dim3(SCAN_CHUNKS, PF_BINS / 256), dim3(256), 0, st, wk.fw);
        hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(1024), 0, st, wk.fw);
        hipLaunchKernelGGL(pairs_scatter_kernel, dim3(FAST_BLOCKS), dim3(256), 0, st, npairs,
                           myers_waves(rmax <= 160 ? 5 : rmax <= 256 ? 8 : 10), wk);
        e = hipGetLastError();
    }
    // One launch per band class (which classes hold tasks is only known on the device; an empty class returns at once),
    // each on a stream of its own forked off `st` behind the scatter pass and joined again before the workspace goes:
    // the classes work on disjoint tasks, and a class with few tasks is a handful of waves whose single lanes run
    // for 0.1 - 0.3 ms -- back to back that was 1.2 ms of latency per call whatever the batch held.
    static thread_local SideStream side[PF_CLASSES + 1];             // (+ 1: the full sweep of the pairs no band class takes)
    constexpr int NSIDE = PF_CLASSES + 1;
    bool forked = false;
    if (e == hipSuccess) {
        forked = true;
        // (More hardware queues for these launches -- priority streams, GPU_MAX_HW_QUEUES -- were measured: every launch
        //  then starts at once and the call gets slower, profiles/round5_pairs_hw_queues_experiment.txt.)
        for (int c = 0; c < NSIDE && forked; ++c) forked = side[c].ready();
        if (forked) {
            e = hipEventRecord(side[0].fork, st);
            for (int c = 0; c < NSIDE && e == hipSuccess; ++c) e = hipStreamWaitEvent(side[c].stream, side[0].fork, 0);
        }
    }
#define ATR_PF_BAND(WB, C) if (e == hipSuccess) e = launch_band<WB>(fp, a, wk, C, forked ? side[C].stream : st)
    ATR_PF_BAND(16, 0); ATR_PF_BAND(32, 1); ATR_PF_BAND(48, 2); ATR_PF_BAND(64, 3);
    ATR_PF_BAND(80, 4); ATR_PF_BAND(96, 5); ATR_PF_BAND(112, 6); ATR_PF_BAND(128, 7);
#undef ATR_PF_BAND
    // the full sweep next to the band classes, not behind them (2 x 250-base pairs: a fifth of the call when it ran last)
    if (e == hipSuccess)
        e = launch_pairs_full_indexed(p, rp, rl, rmax, revcomp, qp, ql, qmax, npairs, out, wk.order, wk.fw.binbase + PF_FALLBACK_BIN,
                                      forked ? side[PF_CLASSES].stream : st);
    if (forked) {                                                    // join whatever was launched (also after an error)
        for (int c = 0; c < NSIDE; ++c) {
            hipError_t j = hipEventRecord(side[c].join, side[c].stream);
            if (j == hipSuccess) j = hipStreamWaitEvent(st, side[c].join, 0);
            if (e == hipSuccess) e = j;
        }
    }
    const hipError_t freed = hipFreeAsync(work, st);
    return e != hipSuccess ? e : freed;
}

}  // namespace atr
