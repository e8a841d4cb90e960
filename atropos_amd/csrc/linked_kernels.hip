// linked_kernels.hip -- the fused linked-adapter pipeline (linked_core.hpp) and its C ABI.
//
//   L1  linked_filter_kernel   one read per lane, the read is loaded once: every anchored 5' adapter
//                              (literal compare, anchored Myers sweep, banded DP for the few that need
//                              it), choice of the matching linked adapter, then the bit-parallel
//                              pre-pass of that adapter's 3' part from column front.rstop on.  Writes
//                              the `which` bytes, the 5' record, the 3' record of every read it can
//                              resolve and a window word for the rest; per-block histogram of the
//                              scatter bins (FILTER_BINS per adapter).
//   L2  scan kernels           (locate_fast.hpp) over n * FILTER_BINS bins
//   L3  linked_scatter_kernel  read indices of the unresolved reads, ordered by (adapter, bin)
//   L4  band_kernel / window_kernel<.., LINKED> once per adapter over its own bins
//
// No torch kernel and no re-pack between the first load of a read and the last store of its records.
#include <hip/hip_runtime.h>
#include <string>

#include "atropos_hip.h"
#include "linked_host.hpp"
#include "locate_fast.hpp"
#include "wave_sweep.hpp"
#include "linked_blob.hpp"

namespace atr {

void launch_fast_scan(FastWork wk, hipStream_t st);
int launch_fast_dp(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                   int max_len, uint4 *out, FastWork wk, const LinkedArgs *la, int idx, int count, hipStream_t st, bool planes, bool one_stream);
int hip_fail(hipError_t e, const char *what);



static_assert(LINKED_MAX * FILTER_BINS <= 1024, "scan_total_kernel scans at most 1024 bins");

// The wave's queued DP tasks, 64 at a time, every lane against its own (read, adapter): the read's
// first chunk is gathered again (L2), the banded DP runs once per DP group that has a task in the
// pass, results are merged into the per-read words (smallest adapter index wins) and counters.
template <bool RAGGED, bool AND_MODE>
__device__ __forceinline__ void linked_drain(const LinkedBlob &S, int ngroups, const uint16_t *queue, int ntasks,
                                             const uint4 *__restrict__ packed, const int32_t *__restrict__ lens,
                                             long long tile_first, int nchunks, int max_len, uint32_t *s_word,
                                             uint32_t *s_count, uint32_t *ns, int lane) {
    linked_drain_with<RAGGED, AND_MODE>(S, ngroups, queue, ntasks, lens, tile_first, max_len, s_word, s_count, ns, lane,
                                        [&](long long tile, int l, int, uint32_t (&w)[4]) {
                                            const uint4 c0 = packed[(size_t)tile * nchunks * 64 + l];
                                            w[0] = c0.x; w[1] = c0.y; w[2] = c0.z; w[3] = c0.w;
                                        });
}

template <bool WIDE, bool RAGGED, bool AND_MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void linked_filter_kernel(const LinkedBlob *__restrict__ blob,
                                                            const uint4 *__restrict__ packed,
                                                            const int32_t *__restrict__ lens, long long nreads,
                                                            int nchunks, int max_len, uint16_t *__restrict__ which_out,
                                                            uint4 *__restrict__ front_out, uint4 *__restrict__ back_out,
                                                            FastWork wk, int lists) {
    __shared__ __attribute__((aligned(16))) LinkedBlob S;
    __shared__ uint32_t s_hist[LINKED_MAX * FILTER_BINS];
    __shared__ uint32_t s_stream[4][FRONT_STREAM][64];             // per wave: the staged reads of the 5' banded DP
    __shared__ uint32_t s_words[4][LINKED_ROUND * 64];             // per wave and read of the round: front word (linked_core.hpp)
    __shared__ uint32_t s_counts[4][LINKED_ROUND * 16];            //                                  number of matching 5' parts (a byte each)
    __shared__ uint16_t s_queue[4][LINKED_TASKS];
    __shared__ uint16_t s_list[4][LINKED_ROUND * 64];               // per wave: the round's reads that have a 5' match
    __shared__ uint32_t s_lcur;                                     // lists: entries of the block's list of unresolved reads
    for (int i = threadIdx.x; i < (int)(sizeof(LinkedBlob) / 4); i += 256) ((uint32_t *)&S)[i] = ((const uint32_t *)blob)[i];
    for (int i = threadIdx.x; i < LINKED_MAX * FILTER_BINS; i += 256) s_hist[i] = 0;
    if (threadIdx.x == 0) s_lcur = 0;
    __syncthreads();

    const int nad = rfl(S.p.n), ngroups = rfl(S.p.ngroups);
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));     // wave-uniform, and the compiler knows
    uint32_t *s_word = s_words[wave], *s_count = s_counts[wave];
    uint16_t *queue = s_queue[wave];
    uint32_t *ns = &s_stream[wave][0][lane];
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    for (long long tile_first = t0 + wave; tile_first < t1; tile_first += 4 * LINKED_ROUND) {
        const int slots = (int)min((long long)LINKED_ROUND, (t1 - tile_first + 3) / 4);
        // ---- 5' parts of up to LINKED_ROUND tiles: literal compare, exact-piece test, DP tasks ------------
        // (1) the literal compare of every adapter on every read.  A literal occurrence of adapter a rules out the
        // adapters of excl[a]; a read with none left is through (four in five of C4's reads with a 5' part carry it
        // verbatim), the others are listed with the adapters still open: s_list entry = cell | open << 9.
        int nopen = 0;                                                        // wave-uniform
        const uint32_t all_ad = (1u << nad) - 1u;
        for (int slot = 0; slot < slots; ++slot) {
            const long long tile = tile_first + 4 * slot;
            const bool live = tile * 64 + lane < nreads;
            const uint4 c0v = packed[(size_t)tile * nchunks * 64 + lane];
            const uint32_t w0[4] = {c0v.x, c0v.y, c0v.z, c0v.w};
            uint32_t word = FRONT_NONE, count = 0u, open = live ? all_ad : 0u;
            for (int a = 0; a < nad; ++a) {                                  // wave-uniform
                const FrontParams &fp = S.p.f[a];
                const int m = rfl(fp.m);
                const bool exact = rfl(fp.accept_full) != 0 && front_exact(fp.code, fp.code_mask, w0);
                if (exact) {
                    ++count;
                    word = min(word, front_word(a, m, m, 0));                // Match(0, m, 0, m, m, 0)
                    open &= ~((1u << a) | (uint32_t)rfl((int)S.p.excl[a]));
                }
            }
            s_word[slot * 64 + lane] = live ? word : FRONT_NONE;
            // four counters to a dword: gathered with two shuffles, stored by every fourth lane
            uint32_t packed_counts = live ? count : 0u;
            packed_counts |= (uint32_t)__shfl_down((int)packed_counts, 1, 64) << 8;
            packed_counts |= (uint32_t)__shfl_down((int)packed_counts, 2, 64) << 16;
            if ((lane & 3) == 0) s_count[(slot * 64 + lane) >> 2] = packed_counts;
            const unsigned long long om = __ballot(open != 0u);
            if (open != 0u) s_list[wave][nopen + __popcll(om & ((1ull << lane) - 1ull))] = (uint16_t)((slot * 64 + lane) | (open << 9));
            nopen += (int)__popcll(om);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // (2) the exact-piece test of the open (read, adapter) pairs, the listed reads 64 at a time
        int ntasks = 0;
        for (int base = 0; base < nopen; base += 64) {
            if (ntasks + 64 * nad > LINKED_TASKS) {                          // wave-uniform: room for this group's tasks?
                linked_drain<RAGGED, AND_MODE>(S, ngroups, queue, ntasks, packed, lens, tile_first, nchunks, max_len,
                                               s_word, s_count, ns, lane);
                ntasks = 0;
            }
            const bool valid = base + lane < nopen;
            const uint32_t entry = (uint32_t)s_list[wave][valid ? base + lane : base];
            const int cell = (int)(entry & 511u);
            const uint32_t open = valid ? entry >> 9 : 0u;
            const long long tile = tile_first + 4 * (cell >> 6);
            const uint4 c0v = packed[(size_t)tile * nchunks * 64 + (cell & 63)];
            const uint32_t w0[4] = {c0v.x, c0v.y, c0v.z, c0v.w};
            bool pex_ok[LINKED_MAX];
            const bool shared = rfl(S.p.pex_shared) != 0;
            if (shared) front_pex_candidates_shared<AND_MODE>(S.p.f, nad, w0, pex_ok);
            for (int a = 0; a < nad; ++a) {                                  // wave-uniform
                const FrontParams &fp = S.p.f[a];
                bool hit;
                if (shared) {
                    hit = pex_ok[0];
#pragma unroll
                    for (int t = 1; t < LINKED_MAX; ++t) if (a == t) hit = pex_ok[t];
                } else {
                    hit = front_pex_candidate<AND_MODE>(fp.pex_code, fp.pex_mask, fp.pex_off, rfl(fp.npieces), rfl(fp.k), w0);
                }
                const bool cand = ((open >> a) & 1u) != 0u && hit;
                const unsigned long long votes = __ballot(cand);
                if (cand) queue[ntasks + __popcll(votes & ((1ull << lane) - 1ull))] = (uint16_t)((cell << 6) | a);
                ntasks += (int)__popcll(votes);
            }
        }
        linked_drain<RAGGED, AND_MODE>(S, ngroups, queue, ntasks, packed, lens, tile_first, nchunks, max_len, s_word, s_count,
                                       ns, lane);

        // ---- 3' part of the adapter whose 5' part matched, on read[front.rstop:] -----------------------------
        // First the records every read gets (5' record, `which`), and the reads WITH a 5' match listed in LDS: one
        // read in five of C4 has none, and a lane without one would idle through the whole 3' sweep of its tile.
        // The sweep below takes the listed reads 64 at a time, whichever tiles of the round they come from (a
        // lane's chunks are gathered through its own pointer either way; neighbours in the list are neighbours in
        // memory most of the time).
        int nact = 0;                                                          // wave-uniform
        for (int slot = 0; slot < slots; ++slot) {
            const long long tile = tile_first + 4 * slot;
            const long long r = tile * 64 + lane;
            const bool live = r < nreads;
            const uint32_t word = s_word[slot * 64 + lane];
            const int which = word == FRONT_NONE ? -1 : (int)(word >> 24);
            const bool has = live && which >= 0;
            if (live) {
                uint32_t frec[4];
                front_word_record(word, S.p.f[which < 0 ? 0 : which].m, frec);
                front_out[r] = make_uint4(frec[0], frec[1], frec[2], frec[3]);
                const uint32_t count = (s_count[(slot * 64 + lane) >> 2] >> (8 * (lane & 3))) & 0xFFu;
                which_out[r] = (uint16_t)((uint32_t)(which & 0xFF) | (count << 8));
                if (!has) {
                    uint32_t none[4];
                    rec_none(none);
                    if (!lists) wk.win[r] = 0u;
                    back_out[r] = make_uint4(none[0], none[1], none[2], none[3]);
                }
            }
            const unsigned long long hm = __ballot(has);
            if (has) s_list[wave][nact + __popcll(hm & ((1ull << lane) - 1ull))] = (uint16_t)(slot * 64 + lane);
            nact += (int)__popcll(hm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int base = 0; base < nact; base += 64) {
            const bool has = base + lane < nact;
            const int cell = has ? (int)s_list[wave][base + lane] : (int)s_list[wave][base];
            const long long tile = tile_first + 4 * (cell >> 6);
            const long long r = tile * 64 + (cell & 63);
            const bool live = has;
            const int n = has ? (RAGGED ? lens[r] : max_len) : 0;
            const uint4 *tp = packed + (size_t)tile * nchunks * 64 + (cell & 63);
            const uint32_t word = s_word[cell];
            const int which = has ? (int)(word >> 24) : 0;
            const BackParams &bp = S.p.b[which];
            uint32_t ww = 0u;
            uint32_t brec[4];
            rec_none(brec);
            int mb = 0;
            if (wave_any(has)) {
                const int s = has ? (int)((word >> 16) & 0xFFu) : 0x3fff;      // lanes without a 5' match see no base at all
                Uniform ub = front_uniform(bp.m, bp.k, bp.indel, bp.min_overlap);
                ub.sq = true; ub.er = true;                                    // (only m, k, indel, min_overlap, sr are read)
                mb = ub.m;
                const int mf = bp.rows;
                FilterState F;
                filter_init(F, ub, mf, WIDE);
                const uint32_t kreg = (uint32_t)ub.k;                          // the lane's own adapter's k
                const int jhi = RAGGED ? wave_max_i32(has ? n : 0) : max_len;
                const int jfull = RAGGED ? wave_min_i32(has ? n : 0x7fffffff) : max_len;   // columns every matched lane has
                const int s_lo = wave_min_i32(has ? s : 0x7fffffff), s_hi = wave_max_i32(has ? s : 0);
                const int z_first = s_lo >> 3;                                 // first dword holding a live base
                const char *peq_base = (const char *)bp.peq;
                if (jhi > 8 * z_first) {
                    const int cfirst = z_first >> 2, c1 = (jhi + 31) >> 5;
                    uint4 nxt = tp[(size_t)cfirst * 64];
                    int jlast = 8 * z_first;
                    for (int c = cfirst; c < c1; ++c) {
                        const uint4 cur = nxt;
                        if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int z = 4 * c + d;
                            if (z < z_first || 8 * z >= jhi) continue;         // wave-uniform
                            uint32_t w = d == 0 ? cur.x : d == 1 ? cur.y : d == 2 ? cur.z : cur.w;
                            if (8 * z < s_hi) w &= start_mask(z, s);           // wave-uniform: only the first dwords
                            uint2 e[8];
                            fetch_peq8((const uint2 *)peq_base, w, e);
                            const int j0 = 8 * z;
                            if (j0 + 8 <= jhi && j0 + 8 <= jfull) {             // wave-uniform
#pragma unroll
                                for (int b = 0; b < 8; ++b) filter_step<WIDE>(F, e[b].x, e[b].y, kreg);
                            } else {
#pragma unroll
                                for (int b = 0; b < 8; ++b) {
                                    const int j = j0 + b + 1;
                                    if (j <= jhi && (!RAGGED || j <= n)) filter_step<WIDE>(F, e[b].x, e[b].y, kreg);
                                }
                            }
                            jlast = min(jhi, j0 + 8);
                        }
                        filter_fold(F, RAGGED ? min(n, jlast) : jlast, mf, kreg);        // at most 32 columns since the last fold
                    }
                }
                LaneFilterParams lf;
                lf.rows = mf; lf.and_mode = AND_MODE ? 1 : 0; lf.tail = bp.tail; lf.thr_row = bp.thr_row; lf.cert = bp.cert;
                ww = filter_decide<WIDE>(F, ub, lf, (const uint32_t *)tp, nchunks, n, brec, s);
                if (!has) { ww = 0u; rec_none(brec); }
                else if (!window_valid(ww))
                    linked_finish(brec, s, ub.m, ub.min_overlap, bp.pf_thr, bp.accept_full != 0, S.rmp.back[which],
                                  S.rmp.back_ld[which], S.rmp.back_max[which]);
            }
            if (live) {
                if (!lists) wk.win[r] = ww;
                if (!window_valid(ww)) back_out[r] = make_uint4(brec[0], brec[1], brec[2], brec[3]);
                else atomicAdd(&s_hist[linked_bin(ww, which, mb, !RAGGED)], 1u);
            }
            if (lists) {
                // long batches: the unresolved reads go into the block's list (at its first read in `tmp`) instead of a
                // window word per read -- the scatter pass reads 8 bytes per unresolved read, not 4 per read
                const bool open = live && window_valid(ww);
                const unsigned long long om = __ballot(open);
                if (om != 0ull) {                                              // wave-uniform
                    uint32_t at = 0u;
                    if (lane == 0) at = atomicAdd(&s_lcur, (uint32_t)__popcll(om));
                    at = (uint32_t)rfl((int)at);
                    if (open) wk.tmp[t0 * 64 + at + __popcll(om & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)r, ww);
                }
            }
        }
    }
    __syncthreads();
    const int nbins = nad * FILTER_BINS;
    fused_hist_flush(wk, s_hist);                                  // (long batches: offsets inside the bins by atomics, fast_work.hpp)
    if (threadIdx.x == 0) wk.lcount[blockIdx.x] = s_lcur;
}

// Short batches: the 3' part of every read the pre-pass left open on a wavefront of its own (wave_sweep.hpp) -- the
// aligner of the read's own adapter on read[front.rstop:], exactly the reference's second match_to -- instead of
// counting sort + four band and four window launches whose single lanes run for 17 - 34 us each.
constexpr long long LINKED_WAVE_MAX_READS = 262144;

// K4a of a linked set: the band reads of EVERY adapter in one launch (band_kernel<.., LINKED> once per adapter ran
// four latency-bound launches one after the other on the band stream: 4 x 70 us on the critical path of C4's step for
// the work of one C2-sized launch).  A wave belongs to one adapter -- the waves are dealt over the adapters' band
// slots in adapter order -- and reads that adapter's parameters with scalar loads; thresholds of all adapters in LDS.
template <bool AND_MODE>
__global__ __launch_bounds__(256) void linked_band_kernel(const LinkedWaveBlob *__restrict__ blob, int nad,
                                                          const uint4 *__restrict__ packed, const int32_t *__restrict__ lens,
                                                          long long nreads, int nchunks, int max_len,
                                                          const uint4 *__restrict__ front, uint4 *__restrict__ out, FastWork wk) {
    __shared__ int16_t s_thr[LINKED_MAX][ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_stream[4][BAND_STREAM][64];
    for (int i = threadIdx.x; i < LINKED_MAX * (ATR_MAX_REF_LEN + 2); i += 256)
        s_thr[i / (ATR_MAX_REF_LEN + 2)][i % (ATR_MAX_REF_LEN + 2)] = blob->p[i / (ATR_MAX_REF_LEN + 2)].thr[i % (ATR_MAX_REF_LEN + 2)];
    __syncthreads();
    long long base[LINKED_MAX], total[LINKED_MAX], wfirst[LINKED_MAX + 1], tasks = 0;
#pragma unroll
    for (int a = 0; a < LINKED_MAX; ++a) {
        base[a] = a < nad ? (long long)wk.binbase[a * FILTER_BINS] : 0;
        total[a] = a < nad ? (long long)wk.binbase[a * FILTER_BINS + BAND_BINS] : 0;
        tasks += total[a] - base[a];
    }
    const int lpw = dp_lanes_per_wave(tasks, (long long)gridDim.x * 4, wk.lpw);
    wfirst[0] = 0;
#pragma unroll
    for (int a = 0; a < LINKED_MAX; ++a) wfirst[a + 1] = wfirst[a] + (total[a] - base[a] + lpw - 1) / lpw;
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    for (long long wv = (long long)blockIdx.x * 4 + wave; wv < wfirst[LINKED_MAX]; wv += (long long)gridDim.x * 4) {
        int a = 0;
#pragma unroll
        for (int t = 1; t < LINKED_MAX; ++t) a += wv >= wfirst[t] ? 1 : 0;              // wave-uniform
        a = rfl(a);
        const LocateParams &p = blob->p[a];
        const BandParams &bp = blob->bp[a];
        const LinkedPost &post = blob->post[a];
        const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
        long long b0 = base[0], t0 = total[0], w0 = wfirst[0];
#pragma unroll
        for (int t = 1; t < LINKED_MAX; ++t) if (a == t) { b0 = base[t]; t0 = total[t]; w0 = wfirst[t]; }
        const long long slot = b0 + (wv - w0) * lpw + lane;
        const bool live = lane < lpw && slot < t0;
        const uint2 task = live ? wk.order[slot] : make_uint2(0u, 0u);
        const long long r = (long long)task.x;
        const uint32_t ww = task.y;
        const int n = live ? (lens ? lens[r] : max_len) : 0;
        const bool last = live && window_scan(ww);
        const bool any_last = wave_max_i32(last ? 1 : 0) != 0, any_rowm = wave_max_i32(live && !last ? 1 : 0) != 0;
        const int s_lane = (live && !last) ? window_hi(ww) - u.m + u.k - window_lo(ww) : 0;
        const int smax = min(BAND_W - 1, wave_max_i32(s_lane));
        const uint32_t *q = (const uint32_t *)(packed + ((size_t)(r >> 6) * nchunks) * 64 + (r & 63));
        uint32_t *ns = &s_stream[wave][0][lane];
        band_stage(q, nchunks, window_lo(ww), ns, 64, band_stream_dwords(u.m));
        uint32_t rec[4] = {0xFFFF0000u, 0u, 0u, 0u};
        const int16_t *thr = s_thr[a];
        if (any_rowm) band_locate<AND_MODE>(u, bp.rrep, bp.noindel != 0, ns, 64, n, ww, smax, thr, rec);
        if (any_last) {
            const int smax_l = min(BAND_W - 1, wave_max_i32(last ? last_band_width(ww) : 0));
            const int rows_max = wave_max_i32(last ? (last_band_rowm(ww) ? u.m : window_rows(ww)) : 0);
            const int cap_lo = wave_min_i32(last ? window_rows(ww) - last_band_span(ww) : 0x7fffffff);
            uint32_t rec_l[4];
            band_locate_last<AND_MODE>(u, bp.rrep, bp.noindel != 0, ns, 64, n, ww, last, smax_l, rows_max, cap_lo, thr, rec_l);
            if (last) { rec[0] = rec_l[0]; rec[1] = rec_l[1]; rec[2] = rec_l[2]; rec[3] = rec_l[3]; }
        }
        if (live) {
            linked_finish(rec, (int)(front[r].y >> 16), post.m, post.min_overlap, post.pf_thr, post.accept_full != 0, post.rmp,
                          post.rmp_ld, post.max_rmp);
            out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        }
    }
}

// (called by launch_fast_dp, filter_kernels.hip, on the band stream after the fork)
int launch_linked_band(const void *d_wave, int nad, bool and_mode, const uint4 *packed, const int32_t *lens, long long nreads,
                       int nchunks, int max_len, const uint4 *front, uint4 *out, FastWork wk, hipStream_t st) {
    const dim3 grid((unsigned)std::min<long long>((nreads + 255) / 256, 4096)), block(256);
    const LinkedWaveBlob *blob = (const LinkedWaveBlob *)d_wave;
    if (and_mode) hipLaunchKernelGGL(linked_band_kernel<true>, grid, block, 0, st, blob, nad, packed, lens, nreads, nchunks, max_len, front, out, wk);
    else          hipLaunchKernelGGL(linked_band_kernel<false>, grid, block, 0, st, blob, nad, packed, lens, nreads, nchunks, max_len, front, out, wk);
    return (int)hipGetLastError();
}

template <int R>
__global__ __launch_bounds__(64) void linked_wave_kernel(const LinkedWaveBlob *__restrict__ blob, const uint4 *__restrict__ packed,
                                                         const int32_t *__restrict__ lens, long long nreads, int nchunks,
                                                         int max_len, const uint16_t *__restrict__ which_out,
                                                         const uint4 *__restrict__ front, uint4 *__restrict__ back_out,
                                                         const uint32_t *__restrict__ win) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ __attribute__((aligned(16))) uint32_t s_code[WAVE_CODE_PAD + (ATR_MAX_READ_LEN + 31) / 32 * 32 + 2 * WAVE_CODE_PAD];
    const int lane = threadIdx.x;
    const long long r = blockIdx.x;
    if (!window_valid(win[r])) return;                               // resolved by the pre-pass (whole wave)
    const int which = __builtin_amdgcn_readfirstlane((int)(which_out[r] & 0xFFu));
    const int s = __builtin_amdgcn_readfirstlane((int)(front[r].y >> 16));
    const int n = __builtin_amdgcn_readfirstlane(lens ? min(max(lens[r], 0), max_len) : max_len);
    const LocateParams &p = blob->p[which];
    const LinkedPost &post = blob->post[which];
    for (int i = lane; i <= p.m + 1; i += 64) s_thr[i] = p.thr[i];
    if (lane < (n + 31) / 32) {
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
    wave_locate<true, true, R>(p, s_thr, s_code + WAVE_CODE_PAD + s, max(0, n - s), lane, rec);     // regular 3' adapters: flags 14
    if (lane == 0) {
        if (rec_found(rec)) {                                        // Adapter.match_to's own test (adapters/__init__.py:386-398)
            const int refstart = (int)(rec[0] & 0xFFFFu), refstop = (int)(rec[0] >> 16);
            if (!linked_accept(refstop - refstart, (int)(rec[2] & 0xFFFFu), (int)(rec[2] >> 16), post.m, post.min_overlap,
                               post.pf_thr, post.accept_full != 0, post.rmp, post.rmp_ld, post.max_rmp))
                rec_none(rec);
        } else {
            rec_none(rec);
        }
        back_out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

struct LinkedLens { int m[LINKED_MAX]; };

__global__ __launch_bounds__(256) void linked_scatter_kernel(long long nreads, const LinkedLens ms, int by_rows,
                                                             const uint16_t *__restrict__ which_out, FastWork wk, int lists) {
    __shared__ uint32_t s_cur[LINKED_MAX * FILTER_BINS + 1], s_tmp[256];
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    // (the block is one chain of dependent round trips; what does not depend on the scan is requested before it: the
    //  block's offsets, its list length, every thread's first entry and that read's adapter -- 8 192 blocks, four rounds)
    const uint2 *list = wk.tmp + t0 * 64;
    const uint32_t count = lists ? wk.lcount[blockIdx.x] : 0u;
    const uint2 first = threadIdx.x < count ? list[threadIdx.x] : make_uint2(0u, 0u);
    const int first_which = threadIdx.x < count ? (int)(which_out[first.x] & 0xFFu) : 0;
    if (wk.fused) {                                                // (no scan launches before this one: fast_work.hpp)
        uint32_t mine[LINKED_MAX];
#pragma unroll
        for (int q = 0; q < LINKED_MAX; ++q)
            mine[q] = (int)threadIdx.x + 256 * q < wk.nbins ? wk.counts[(size_t)blockIdx.x * wk.nbins + threadIdx.x + 256 * q] : 0u;
        fused_bin_bases(wk, s_cur, s_tmp);
#pragma unroll
        for (int q = 0; q < LINKED_MAX; ++q)
            if ((int)threadIdx.x + 256 * q < wk.nbins) s_cur[threadIdx.x + 256 * q] += mine[q];
    } else {
        for (int b = threadIdx.x; b < wk.nbins; b += 256) s_cur[b] = fast_slot0(wk, b);
    }
    __syncthreads();
    if (lists) {                                                   // the block's list of (read, window word)
        for (uint32_t i = threadIdx.x; i < count; i += 256) {
            const uint2 e = i == threadIdx.x ? first : list[i];
            const int which = i == threadIdx.x ? first_which : (int)(which_out[e.x] & 0xFFu);
            wk.order[atomicAdd(&s_cur[linked_bin(e.y, which, ms.m[which], by_rows != 0)], 1u)] = e;
        }
        return;
    }
    for (long long r = t0 * 64 + threadIdx.x; r < min(nreads, t1 * 64); r += 256) {
        const uint32_t ww = wk.win[r];
        if (window_valid(ww)) {
            const int which = (int)(which_out[r] & 0xFFu);
            wk.order[atomicAdd(&s_cur[linked_bin(ww, which, ms.m[which], by_rows != 0)], 1u)] = make_uint2((uint32_t)r, ww);
        }
    }
}

template <bool WIDE, bool RAGGED>
static void launch_l1(bool and_mode, const LinkedBlob *blob, const uint4 *packed, const int32_t *lens, long long nreads,
                      int nchunks, int max_len, uint16_t *which, uint4 *front, uint4 *back, FastWork wk, int lists, hipStream_t st) {
    const dim3 grid(wk.nused), block(256);
    if (and_mode) hipLaunchKernelGGL((linked_filter_kernel<WIDE, RAGGED, true>), grid, block, 0, st, blob, packed, lens, nreads, nchunks, max_len, which, front, back, wk, lists);
    else          hipLaunchKernelGGL((linked_filter_kernel<WIDE, RAGGED, false>), grid, block, 0, st, blob, packed, lens, nreads, nchunks, max_len, which, front, back, wk, lists);
}

}  // namespace atr

using namespace atr;

extern "C" {

int atr_linked_create(const atr_linked_adapter *adapters, int n_adapters, atr_linked_set **out) {
    if (!out) return ATR_ERR_INVALID;
    *out = nullptr;
    atr_linked_set *s = new (std::nothrow) atr_linked_set();
    if (!s) return ATR_ERR_NOMEM;
    const int rc = linked_fill(s, adapters, n_adapters);
    if (rc != ATR_OK) { delete s; return rc; }
    LinkedBlob blob;
    blob.p = s->p;
    blob.rmp = s->rmp;
    hipError_t e = hipMalloc(&s->d_params, sizeof(LinkedBlob));
    if (e != hipSuccess) { delete s; return e == hipErrorOutOfMemory ? ATR_ERR_NOMEM : hip_fail(e, "hipMalloc(linked set)"); }
    e = hipMemcpy(s->d_params, &blob, sizeof(LinkedBlob), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s->d_params); delete s; return hip_fail(e, "hipMemcpy(linked set)"); }
    {   // the 3' aligners as the wavefront-per-read kernel wants them
        LinkedWaveBlob *hb = new (std::nothrow) LinkedWaveBlob();
        s->d_wave = nullptr;
        if (hb) {
            for (int a = 0; a < LINKED_MAX; ++a) {
                const atr_aligner &b = s->back[a < s->p.n ? a : 0];
                hb->p[a] = b.p;
                hb->post[a] = s->post[a < s->p.n ? a : 0];
                memset(&hb->bp[a], 0, sizeof(BandParams));
                for (int i = 0; i < b.p.m && i < FILTER_MAX_M; ++i) hb->bp[a].rrep[i] = (uint32_t)(b.codes[i] & 15u) * 0x11111111u;
                hb->bp[a].and_mode = (b.wildcard_ref || b.wildcard_query) ? 1 : 0;
                hb->bp[a].noindel = b.indel_cost > b.p.k ? 1 : 0;
            }
            if (hipMalloc(&s->d_wave, sizeof(LinkedWaveBlob)) == hipSuccess) {
                if (hipMemcpy(s->d_wave, hb, sizeof(LinkedWaveBlob), hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipFree(s->d_wave);
                    s->d_wave = nullptr;
                }
            } else {
                s->d_wave = nullptr;
                (void)hipGetLastError();
            }
            delete hb;
        }
        if (!s->d_wave) {                                          // the band and window launches read it: no set without it
            (void)hipFree(s->d_params);
            delete s;
            return ATR_ERR_NOMEM;
        }
    }
    *out = s;
    return ATR_OK;
}

void atr_linked_destroy(atr_linked_set *s) {
    if (!s) return;
    if (s->d_wave) (void)hipFree(s->d_wave);
    if (s->d_params) (void)hipFree(s->d_params);
    delete s;
}

int atr_linked_query_table(const atr_linked_set *s) { return s ? s->table_kind : ATR_ERR_INVALID; }

size_t atr_linked_work_bytes(const atr_linked_set *s, int64_t nreads) {
    return (!s || nreads < 0) ? 0 : fast_work_bytes(nreads, s->p.n * FILTER_BINS);
}

int atr_linked_match_batch(const atr_linked_set *s, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads,
                           int max_len, int8_t *d_which, atr_result *d_front, atr_result *d_back, void *d_work,
                           void *stream) {
    if (!s || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_which || !d_front || !d_back || !d_work || (max_len > 0 && !d_packed)) return ATR_ERR_INVALID;
    if (max_len == 0) return ATR_ERR_UNSUPPORTED;                  // an all-empty batch has no packed chunk to read
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = (max_len + 31) / 32;
    FastWork wk = fast_carve(d_work, nreads, s->p.n * FILTER_BINS);
    wk.nused = fast_blocks_for((nreads + 63) / 64);                // a short batch: only the blocks it fills (and their histogram rows)
    wk.lpw = nreads <= 8192 ? 0 : 64;
    const LinkedBlob *blob = (const LinkedBlob *)s->d_params;
    const bool ragged = d_lens != nullptr, and_mode = s->p.and_mode != 0;
    // (the wavefront-per-read finish of short batches reads a window word per read; long batches keep per-block lists)
    const int lists = nreads > LINKED_WAVE_MAX_READS ? 1 : 0;
    wk.fused = (lists && fast_fused_scan()) ? 1 : 0;
    if (wk.fused) {                                                // the bins' totals the pre-pass blocks add to (fused_hist_flush)
        const hipError_t rc = hipMemsetAsync(wk.chunks, 0, (size_t)wk.nbins * 4, st);
        if (rc != hipSuccess) return hip_fail(rc, "linked bin totals memset");
    }
    uint16_t *which = (uint16_t *)d_which;
    uint4 *front = (uint4 *)d_front, *back = (uint4 *)d_back;
    const uint4 *packed = (const uint4 *)d_packed;
    if (s->p.wide) {
        if (ragged) launch_l1<true, true>(and_mode, blob, packed, d_lens, nreads, nchunks, max_len, which, front, back, wk, lists, st);
        else        launch_l1<true, false>(and_mode, blob, packed, d_lens, nreads, nchunks, max_len, which, front, back, wk, lists, st);
    } else {
        if (ragged) launch_l1<false, true>(and_mode, blob, packed, d_lens, nreads, nchunks, max_len, which, front, back, wk, lists, st);
        else        launch_l1<false, false>(and_mode, blob, packed, d_lens, nreads, nchunks, max_len, which, front, back, wk, lists, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "linked_filter_kernel launch");
    {   // short batch: the open 3' parts on a wavefront each, no sort, no band / window launches
        bool regular = s->d_wave != nullptr && nreads <= LINKED_WAVE_MAX_READS;
        int mmax = 0;
        for (int a = 0; a < s->p.n; ++a) {
            regular = regular && s->back[a].flags == (ATR_START_WITHIN_SEQ2 | ATR_STOP_WITHIN_SEQ2 | ATR_STOP_WITHIN_SEQ1);
            mmax = std::max(mmax, s->back[a].p.m);
        }
        if (regular) {
            const LinkedWaveBlob *wb = (const LinkedWaveBlob *)s->d_wave;
            const dim3 grid((unsigned)nreads), block(64);
            switch (wave_pair_rows(mmax)) {
                case 1: hipLaunchKernelGGL(linked_wave_kernel<1>, grid, block, 0, st, wb, packed, d_lens, (long long)nreads, nchunks, max_len, (const uint16_t *)which, (const uint4 *)front, back, (const uint32_t *)wk.win); break;
                case 2: hipLaunchKernelGGL(linked_wave_kernel<2>, grid, block, 0, st, wb, packed, d_lens, (long long)nreads, nchunks, max_len, (const uint16_t *)which, (const uint4 *)front, back, (const uint32_t *)wk.win); break;
                default: hipLaunchKernelGGL(linked_wave_kernel<3>, grid, block, 0, st, wb, packed, d_lens, (long long)nreads, nchunks, max_len, (const uint16_t *)which, (const uint4 *)front, back, (const uint32_t *)wk.win); break;
            }
            e = hipGetLastError();
            return e == hipSuccess ? ATR_OK : hip_fail(e, "linked_wave_kernel launch");
        }
    }
    if (!wk.fused) launch_fast_scan(wk, st);
    LinkedLens ms;
    for (int a = 0; a < LINKED_MAX; ++a) ms.m[a] = a < s->p.n ? s->p.b[a].m : 0;
    hipLaunchKernelGGL(linked_scatter_kernel, dim3(wk.nused), dim3(256), 0, st, (long long)nreads, ms, ragged ? 0 : 1,
                       (const uint16_t *)which, wk, lists);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "linked scatter launch");
    // one window launch for the whole set when the 3' aligners share window_kernel's template parameters (row class,
    // indel mode): block row y serves adapter y (locate_fast.hpp) -- four launches of a few hundred long tasks each
    // otherwise, on three side streams (C4: the fourth started when the first had drained, 40 us past the band launch)
    static const bool one_window_on = [] { const char *x = getenv("ATR_ONE_WINDOW"); return !(x && x[0] == '0'); }();   // (A/B switch)
    bool one_window = s->p.n > 1 && one_window_on;
    for (int a = 1; a < s->p.n; ++a)
        one_window = one_window && round_up_rows(s->back[a].p.m) == round_up_rows(s->back[0].p.m) &&
                     (s->back[a].indel_cost > s->back[a].p.k) == (s->back[0].indel_cost > s->back[0].p.k);
    const LinkedWaveBlob *wblob = (const LinkedWaveBlob *)s->d_wave;
    for (int a = 0; a < s->p.n; ++a) {
        LinkedArgs la;
        la.bin0 = a * FILTER_BINS;
        la.front = front;
        la.post = s->post[a];
        la.multi = s->d_wave;                                      // every adapter's band reads in one launch (idx 0)
        la.multi_and = s->p.and_mode != 0;
        la.multi_p = wblob->p;                                     // (addresses inside the device blob; not read here)
        la.multi_post = wblob->post;
        la.a0 = a;
        la.win_count = one_window ? (a == 0 ? s->p.n : 0) : 1;
        const int rc = launch_fast_dp(&s->back[a], packed, d_lens, nreads, nchunks, max_len, back, wk, &la, a, s->p.n, st, false, false);
        if (rc != 0) return hip_fail((hipError_t)rc, "linked band / window launch");
    }
    return ATR_OK;
}

}  // extern "C"
