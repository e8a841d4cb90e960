// piece_filter.hpp -- the kernel body of the two-pass pre-pass on plane64 reads (piece_core.hpp), shared by
//   * piece_kernels.hip: piece_filter_kernel<NW, RAGGED>, the aligner's parameters as kernel arguments, and
//   * piece_spec.hip: the same body compiled at run time for ONE aligner (jit.hpp), its parameters constexpr.
// Device code only: this header and what it includes must stay clean for hiprtc (no host library calls).
#ifndef ATR_PIECE_FILTER_HPP
#define ATR_PIECE_FILTER_HPP

#include "fast_work.hpp"
#include "pack_fast.hpp"

namespace atr {

// waves per SIMD the kernel is built for, by word count (96 VGPRs at five)
#ifndef ATR_PIECE_PREFETCH_EARLY
#define ATR_PIECE_PREFETCH_EARLY 1
#endif
// Pass B's window out of LDS: a lane that queues a task stores the planes pass A holds for it (NW x 16 bytes) next to the
// task, [word][slot] per wave, and pass B picks its three chunks from there -- instead of gathering them from the batch
// (three 16-byte pieces per flagged read out of lines the L2 had dropped by then: 212 MB of C2's counted fetch,
// profiles/round5_c2_fetch_calibration.txt).  Word counts whose stash would cost a resident block (9, 10) keep the gather.
#ifndef ATR_PIECE_STASH
#define ATR_PIECE_STASH 1
#endif
#ifndef ATR_PIECE_WAVES
#if defined(ATR_SPEC) && defined(ATR_SPEC_ASCII)
#define ATR_PIECE_WAVES(NW) 2                                           // (the tile's ASCII rows are staged in LDS: two blocks per CU)
#elif defined(ATR_SPEC)
#define ATR_PIECE_WAVES(NW) ((NW) <= 6 ? 4 : 3)                         // (no spills at these: tools/jit/spec_offline.sh)
#else
#define ATR_PIECE_WAVES(NW) ((NW) <= 6 ? 4 : (NW) <= 8 ? 3 : 2)          // (the generic kernel holds five piece accumulators)
#endif
#endif

constexpr int PIECE_QF = 2;                         // queue fields per task: read, meta
template <bool B> struct PieceBool { static constexpr bool value = B; };

// set bits of `mask` below this lane (v_mbcnt: no lane-mask register to keep alive -- the 64-bit mask of round 4 was
// spilled in the tile loop, and a scratch reload waits for every load in flight, the next tile's planes included)
__device__ __forceinline__ int lane_rank(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The record of a resolved read / the list entry of an unresolved one.  `active`: this lane holds a decided read.
// nibs: the lane's eight window dwords in LDS ([dword][lane]; the window's first position is column j0 + 1), from
// which the 64 codes that follow the entry's first diagonal go into tdata -- what band_stage would gather.
// nd: the dwords of a lane's window in nibs (8, or 12 with the 96-column window).
__device__ __forceinline__ void piece_emit(bool active, long long r, uint32_t ww, const uint32_t (&rec)[4], int m, uint4 *out,
                                           uint2 *list, uint4 *ldata, const uint32_t *nibs, int j0, uint32_t *s_lcur, uint32_t *s_hist,
                                           int nd = 8) {
    const bool open = active && window_valid(ww);
    if (active && !open) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    const uint64_t om = __ballot(open);
    if (om != 0ull) {                                            // wave-uniform
        uint32_t base = 0u;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(s_lcur, (uint32_t)__popcll(om));
        base = __builtin_amdgcn_readfirstlane(base);
        const int lane = threadIdx.x & 63;
        const int off = window_lo(ww) - j0;                      // codes between the window's start and the first diagonal
        // (the band kernels read 3 + m / 8 dwords from the record: eight hold them for adapters of up to 47 bases)
        const bool dense = nibs != nullptr && off >= 0 && off < 64 && band_stream_dwords(m) <= 8;
        if (open) {
            const uint32_t slot = base + (uint32_t)lane_rank(om);
            list[slot] = make_uint2((uint32_t)r, dense ? ww : (ww | PIECE_NODENSE));
            atomicAdd(&s_hist[window_bin(ww, m, true)], 1u);
            if (dense) {
                const int z = off >> 3;
                const uint32_t sh = 4u * (uint32_t)(off & 7);
                uint32_t raw[9], d[8];
#pragma unroll
                for (int k = 0; k < 9; ++k) raw[k] = z + k < nd ? nibs[(z + k) * 64 + lane] : 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = sh ? ((raw[k] >> sh) | (raw[k + 1] << (32u - sh))) : raw[k];
                ldata[2 * (size_t)slot] = make_uint4(d[0], d[1], d[2], d[3]);
                ldata[2 * (size_t)slot + 1] = make_uint4(d[4], d[5], d[6], d[7]);
            }
        }
    }
}

// RAGGED: reads of different lengths (lens[r] <= max_len).  Pass A then works on the read moved to the END of its
// NW words (a per-lane shift of the planes by 32 NW - n positions, zeros coming in below): every read ends at
// position 32 NW like a read of an equal-length batch, the read-end conditions keep their wave-uniform masks, and the
// columns it reports are shifted back by the lane's own amount.  Positions before the read hold code 0 either way.
// WW: plane words a pass-B task carries (2: PIECE_WINDOW columns; 3: PIECE_WINDOW_MAX, adapters whose rows + 2 k exceed 64)
template <int NW, bool RAGGED, int WW = 2>
__device__ __forceinline__ void piece_filter_body(const LocateParams &p, const FilterParams &fp_arg, const PieceParams &pp_arg,
                                                  const uint4 *__restrict__ planes, const int32_t *__restrict__ lens,
                                                  long long nreads, int max_len, uint4 *__restrict__ out, FastWork wk,
                                                  const uint8_t *__restrict__ ascii = nullptr, const uint8_t *tab = nullptr) {
#ifdef ATR_SPEC
    if (!RAGGED) max_len = spec::N;                               // (what the host compiled this kernel for)
#endif
#ifdef ATR_SPEC_ASCII
    // FUSED ASCII ENTRY (round 6): the batch arrives as rows of ASCII (row stride ATR_SPEC_STRIDE, a constant of this
    // build) and `planes` is an OUTPUT: a wave stages its tile's rows in LDS with coalesced 16-byte loads (requested a
    // tile ahead into registers), packs its read into the four bit planes in registers (pack_fast.hpp: the very code of
    // pack_kernel), stores them -- the DP kernels and the full sweep of the listed reads gather from them -- and goes
    // straight into pass A.  What the two-kernel form (atr_pack_planes, then this pre-pass) writes and reads again -- 80
    // bytes per read each way -- is written once and read by the few reads that need it.
    static_assert(NW <= 8, "the fused ASCII entry keeps pass B's window in the LDS stash");
    constexpr int STRIDE = ATR_SPEC_STRIDE;
    constexpr int NPC = (15 + 64 * STRIDE + 1023) / 1024;          // 16-byte pieces per lane of a tile's aligned window
    constexpr int STAGE_BYTES = (64 * STRIDE + PACK_STAGE_SLACK + 15) & ~15;
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][STAGE_BYTES];
    __shared__ uint8_t s_tab[256];
    s_tab[threadIdx.x] = tab[threadIdx.x];
    uint4 *planes_out = const_cast<uint4 *>(planes);
    const uint8_t *buf_end = ascii + nreads * STRIDE;
#endif
    const int n = RAGGED ? 32 * NW : max_len;                     // the length pass A sees (wave-uniform)
    __shared__ uint2 s_peq[16];
    __shared__ uint32_t s_spread[4][256];
    __shared__ uint32_t s_hist[FILTER_BINS];
    __shared__ uint32_t s_lcur, s_wcnt;
    constexpr int PW = 32 * WW, ND = 4 * WW;                      // columns / nibble dwords of a pass-B window
#ifdef ATR_SPEC
    constexpr bool WIDE_OK = true;                                // (mf is a constant: only one of the two sweeps is compiled)
#else
    constexpr bool WIDE_OK = WW == 3;                             // the generic kernels: the two-word sweep in the "big" instantiation only
#endif
    static_assert(WW == 2 || WW == 3, "windows of 64 or 96 columns");
    __shared__ uint32_t s_nibs[4][ND][64];                        // pass B: the task's nibble dwords, [dword][lane] (NARROW tail check)
    __shared__ uint32_t s_queue[4][PIECE_QF][64];
    constexpr bool STASH = ATR_PIECE_STASH != 0 && NW <= 8;
    __shared__ uint4 s_stash[STASH ? 4 : 1][STASH ? NW : 1][64];
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
#ifndef ATR_SPEC
    // The piece parameters go through LDS: as ~800 bytes of kernel argument the compiler kept them in scalar registers
    // across the tile loop and spilled 116 of those into vector lanes (round-4 verdict).  Pass A reads its wave-uniform
    // words with v_readfirstlane (piece_uniform), the masks as vector operands.
    __shared__ PieceParams s_pp;
    __shared__ FilterParams s_fp;
    for (int i = threadIdx.x; i < (int)(sizeof(PieceParams) / 4); i += 256) ((uint32_t *)&s_pp)[i] = ((const uint32_t *)&pp_arg)[i];
    for (int i = threadIdx.x; i < (int)(sizeof(FilterParams) / 4); i += 256) ((uint32_t *)&s_fp)[i] = ((const uint32_t *)&fp_arg)[i];
    const PieceParams &pp = s_pp;
    const FilterParams &fp = s_fp;
#else
    const PieceParams &pp = pp_arg;
    const FilterParams &fp = fp_arg;
#endif
    if (threadIdx.x < 16) s_peq[threadIdx.x] = make_uint2((uint32_t)fp_arg.peq[threadIdx.x], (uint32_t)(fp_arg.peq[threadIdx.x] >> 32));
    if (threadIdx.x < FILTER_BINS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_lcur = 0; s_wcnt = 0; }
    piece_spread_fill(s_spread);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int mf = fp_arg.rows, T = u.m - mf;
    const uint32_t kreg = (uint32_t)u.k;
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    uint2 *list = wk.tmp + t0 * 64;                               // this block's list: at most one entry per read it owns
    uint4 *ldata = wk.tdata + t0 * 128;                           // ... and its 64-code records (two uint4 each)
    // this block's reads that need the full sweep: their numbers (one slot per read it owns)
    uint32_t *wlist = wk.wide + t0 * 64;
    uint32_t (*queue)[64] = s_queue[wave];
    uint32_t *nibs = &s_nibs[wave][0][0];
    uint4 (*stash)[64] = s_stash[STASH ? wave : 0];

    // ---- pass B: one queued task per lane (lane < count) ---------------------------------------------------
    // A task is (read, meta): the lane fetches the 64 positions of every plane that end at its window's last column
    // itself -- bits [j_e - 64, j_e) of the read = its chunks w0, w0 + 1, w0 + 2 (zeros outside the read), funnel-
    // shifted.  The chunks were streamed by pass A a moment ago and sit in the L2; a lane's chunk is the same 16 bytes
    // of the same 1 KiB row its tile mates read.  (Round 4 picked the three words out of pass A's registers with per-
    // lane masks: 150 VALU ops per tile and the planes live through all of pass A; a first version of round 5 fetched
    // them per TILE: a memory round trip in every tile iteration, ATR_X_TIMING.  Here it is one round trip per 64
    // tasks, every lane busy.)
    auto pass_b = [&](int count) __attribute__((always_inline)) {
        const bool act = lane < count;
        const long long r = (long long)queue[0][lane];
        const uint32_t meta = queue[1][lane];
        const int j_e = act ? (int)(meta & 1023u) : 0, need = act ? (int)((meta >> 10) & 127u) : 0;
        const int nr = RAGGED ? (act ? (int)(meta >> 17) : 0) : max_len;       // the read's own length
        uint32_t wp[4][WW];
        {
            // (the stash holds what pass A saw: a ragged read moved to the end of its NW words, 32 NW - nr positions up)
            const int b0 = j_e - PW + (STASH && RAGGED ? 32 * NW - nr : 0), w0 = b0 >> 5;     // floor: -WW .. NW - 2
            uint4 g[WW + 1];
#pragma unroll
            for (int i = 0; i <= WW; ++i) g[i] = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (STASH) {
                if (act) {
#pragma unroll
                    for (int i = 0; i <= WW; ++i) if (w0 + i >= 0 && w0 + i < NW) g[i] = stash[w0 + i][lane];
                }
            } else {
#ifndef ATR_X_NOGATHER                                                     // (traffic calibration only: pass A's stream alone, records wrong)
                const uint4 *tcur = planes + ((size_t)(r >> 6) * NW) * 64 + (r & 63);
                if (act) {
#pragma unroll
                    for (int i = 0; i <= WW; ++i) if (w0 + i >= 0 && w0 + i < NW) g[i] = tcur[(size_t)(w0 + i) * 64];
                }
#endif
            }
            const uint32_t sh = (uint32_t)(b0 & 31);
#pragma unroll
            for (int i = 0; i < WW; ++i) {
                wp[0][i] = __builtin_amdgcn_alignbit(g[i + 1].x, g[i].x, sh);
                wp[1][i] = __builtin_amdgcn_alignbit(g[i + 1].y, g[i].y, sh);
                wp[2][i] = __builtin_amdgcn_alignbit(g[i + 1].z, g[i].z, sh);
                wp[3][i] = __builtin_amdgcn_alignbit(g[i + 1].w, g[i].w, sh);
            }
        }
        const int W = min(PW, (wave_max_i32(need) + 7) & ~7);                     // columns swept, a multiple of eight (<= pp.narrow)
        const int dw0 = ND - (W >> 3);                                         // first of the window's dwords swept
        uint32_t nb[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            nb[d] = 0u;
            // (wave-uniform; one dword before the sweep: the DP's first diagonal may lie k columns before the window)
            if (d >= dw0 - 1) nb[d] = piece_nibbles(s_spread, wp[0][d >> 2], wp[1][d >> 2], wp[2][d >> 2], wp[3][d >> 2], d & 3);
            nibs[d * 64 + lane] = nb[d];                                         // NARROW mode's tail rows, the DP kernels' record
        }
        // WD: two-word bit-vectors -- adapters of 41 .. 64 bases sweep ALL their rows (round 6): every last-column row is
        // exact, so a partial adapter at the read end goes to the last-column band instead of the column window
        const auto sweep_decide = [&](auto wc) __attribute__((always_inline)) {
        constexpr bool WD = decltype(wc)::value;
        FilterState F;
        filter_init(F, u, mf, WD);
        // START_WITHIN_SEQ1: a lane whose sweep reaches back to column 0 starts from the all-zero column (filter_init) and
        // takes the window positions before its first base as columns in which every row matches -- they leave the zero
        // column as it is; a lane whose sweep begins later starts like any other aligner's, "row i reached by i insertions"
        // (an upper bound that is exact for every cell a <= k-error path from row 0 reaches: piece_task).
        int vstart = 0;                                                        // window positions before column 1 (sr lanes from column 0)
        if (u.sr) {
            const bool zero0 = j_e - W <= 0;
            Uniform uf = u;
            uf.sr = false;
            FilterState G;
            filter_init(G, uf, mf, WD);                                        // "row i reached by i insertions"
            if (!zero0) { F.pvl = G.pvl; F.pvh = G.pvh; F.score = G.score; }
            vstart = zero0 ? PW - j_e : 0;
        }
        uint2 ea[8], eb[8];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (d < dw0) continue;                                             // wave-uniform
            uint2 (&e)[8] = (d & 1) ? eb : ea;
            if (d == dw0) fetch_peq8(s_peq, nb[d], e);
            if (d + 1 < ND) fetch_peq8(s_peq, nb[d + 1], (d & 1) ? ea : eb);    // one dword ahead of the columns that use it
            if (u.sr) {                                                        // (wave-uniform; a constant of the run-time compiled kernel)
#pragma unroll
                for (int b = 0; b < 8; ++b) if (8 * d + b < vstart) { e[b].x = ~0u; e[b].y = ~0u; }
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) filter_step<WD>(F, e[b].x, e[b].y, kreg);
            if ((d & 3) == 3 && d + 1 < ND) filter_fold(F, j_e - PW + 8 * (d + 1), mf, kreg);   // 32 columns at most between two folds
        }
        filter_fold(F, j_e, mf, kreg);
        uint32_t rec[4];
        // the T bases after column jp: window position jp - (j_e - PW), eight bases per dword (T <= 32: up to five dwords)
        const auto tm = [&](int jp) {
            const int rp = max(0, jp - (j_e - PW));
            return filter_tail_cmp(fp, T, rp, [&](int z) { return z < ND ? nibs[z * 64 + lane] : 0u; });
        };
        // the adapter's rows against the read on diagonal d (filter_decide_tm's substitution certificate): the window's
        // planes moved down to base d, one "base == code" mask per plane against the rows that hold that code
        const auto dg = [&](int d) -> uint64_t {
            if constexpr (WW != 2 || WD) return ~0ull;                         // (the 96-column window / two words: no diagonal view, the DP decides)
            const int o = d - (j_e - PW);                                      // window bit of the diagonal's first base
            if (o < 0 || o + u.m > PW) return ~0ull;
            uint64_t eq = 0ull;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t lo = wp[c][0], hi = wp[c][1];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q != c && !fp.and_mode) { lo &= ~wp[q][0]; hi &= ~wp[q][1]; }
                eq |= ((((uint64_t)hi << 32) | lo) >> o) & fp.rowsel[c];
            }
            return ~eq & (u.m >= 64 ? ~0ull : (1ull << u.m) - 1ull);
        };
        // (j_e < nr: no read-end condition and the window not cut at the read end -- then j_e - need = d_min - k, or 0, and
        //  j_e - m = d_max + k: the pieces' diagonals, filter_decide_tm)
        const uint32_t ww = filter_decide_tm<WD>(F, u, fp, tm, nr, rec, 0, j_e == nr, dg, j_e - need, j_e < nr ? j_e - u.m : -0x10000);
        piece_emit(act, r, ww, rec, u.m, out, list, ldata, nibs, j_e - PW, &s_lcur, s_hist, ND);
        };
        if (mf > 32) {                                                         // (wave-uniform; a constant of the run-time compiled kernel)
            if constexpr (WIDE_OK) sweep_decide(PieceBool<true>{});
        } else {
            sweep_decide(PieceBool<false>{});
        }
    };

    int qn = 0;                                                   // tasks queued (wave-uniform)

    // The planes of the wave's NEXT tile are requested right after the current tile's have been taken over, BEFORE its
    // pass A: the round trip to HBM runs under ~2.5 k cycles of pass A.  (Round 4 requested them after pass A, "when
    // its registers are free": two iterations in three then consumed them a queue insertion later -- the wave sat
    // through the whole latency, 40 % of its time by s_memtime.)  20 more registers through pass A: 72 of the 96.
#ifdef ATR_SPEC_ASCII
    // the tile's 16-byte aligned window [al, al + need): inside the matrix for every tile but the batch's first (when the
    // matrix does not start on a 16-byte boundary) and last -- those are staged by pack_stage_tile, piece by piece
    uint4 ax[NPC];
    const auto ascii_edge = [&](long long tile) {
        const uint8_t *src = ascii + tile * 64 * STRIDE;
        const uint8_t *al = src - ((uintptr_t)src & 15);
        const long long need = (long long)((uintptr_t)src & 15) + (nreads - tile * 64 < 64 ? nreads - tile * 64 : 64) * STRIDE;
        return al < ascii || al + ((need + 15) & ~15ll) > buf_end;
    };
    const auto ascii_request = [&](long long tile) {
        if (ascii_edge(tile)) return;                              // wave-uniform
        const uint8_t *src = ascii + tile * 64 * STRIDE;
        const uint8_t *al = src - ((uintptr_t)src & 15);
        const long long need = (long long)((uintptr_t)src & 15) + (nreads - tile * 64 < 64 ? nreads - tile * 64 : 64) * STRIDE;
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const long long o = (long long)u * 1024 + (long long)lane * 16;
            ax[u] = o < need ? *(const uint4 *)(al + o) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (t0 + wave < t1) ascii_request(t0 + wave);
#else
    uint4 nx[NW];
    if (t0 + wave < t1) {
        const uint4 *tp = planes + (size_t)(t0 + wave) * NW * 64;             // (wave-uniform base + lane: scalar base, 32-bit offset)
#pragma unroll
        for (int w = 0; w < NW; ++w) nx[w] = tp[w * 64 + lane];
    }
#endif
    for (long long tile = t0 + wave; tile < t1; tile += 4) {
        const long long r = tile * 64 + lane;
        const bool live = r < nreads;
        // ---- pass A ----
        uint32_t pl[NW][4];
#ifdef ATR_SPEC_ASCII
        {
            uint8_t *stage = s_stage[wave];
            uint32_t mis;
            if (ascii_edge(tile)) {
                mis = pack_stage_tile(stage, ascii, STRIDE, nreads, tile, lane);
            } else {
                const uint8_t *src = ascii + tile * 64 * STRIDE;
                mis = (uint32_t)((uintptr_t)src & 15);
                const long long need = (long long)mis + (nreads - tile * 64 < 64 ? nreads - tile * 64 : 64) * STRIDE;
#pragma unroll
                for (int u = 0; u < NPC; ++u) {
                    const long long o = (long long)u * 1024 + (long long)lane * 16;
                    if (o < need) *(uint4 *)(stage + o) = ax[u];
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0);
            }
            if (tile + 4 < t1) ascii_request(tile + 4);           // the next tile's rows: in flight under the pack and pass A
            __builtin_amdgcn_sched_barrier(0);
            const int np = live ? (RAGGED ? min(max(lens[r], 0), max_len) : max_len) : 0;
            const uint32_t rowoff = mis + (uint32_t)lane * (uint32_t)STRIDE;
            uint4 pk[NW];
            bool zero_seen = false;
            pack_planes_row_regs<NW>((const uint32_t *)stage, rowoff >> 2, rowoff & 3u, np, s_tab, pk, zero_seen);
            uint4 *tp = planes_out + (size_t)tile * NW * 64;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                tp[w * 64 + lane] = pk[w];
                pl[w][0] = pk[w].x; pl[w][1] = pk[w].y; pl[w][2] = pk[w].z; pl[w][3] = pk[w].w;
            }
        }
#else
#pragma unroll
        for (int w = 0; w < NW; ++w) { pl[w][0] = nx[w].x; pl[w][1] = nx[w].y; pl[w][2] = nx[w].z; pl[w][3] = nx[w].w; }
#if ATR_PIECE_PREFETCH_EARLY
        if (tile + 4 < t1) {
            const uint4 *tp = planes + (size_t)(tile + 4) * NW * 64;
#pragma unroll
            for (int w = 0; w < NW; ++w) nx[w] = tp[w * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);                        // (the loads stay up here)
#endif
#endif
        int nr = max_len, back = 0;                               // the read's own length; positions it is moved up by
        // (START_WITHIN_SEQ1: the read-start conditions want the words in which the read starts at position 0)
        constexpr int HWN = NW < PIECE_HEAD_WORDS ? NW : PIECE_HEAD_WORDS;
        uint32_t hpl[HWN][4];
#pragma unroll
        for (int w = 0; w < HWN; ++w)
#pragma unroll
            for (int c = 0; c < 4; ++c) hpl[w][c] = pl[w][c];
        if (RAGGED) {
            nr = live ? min(max(lens[r], 0), max_len) : 0;
            back = 32 * NW - nr;
            // out[w] = (in[w - q] : in[w - q - 1]) >> ((-back) & 31), q = (back - 1) >> 5 in -1 .. NW - 1: the two words
            // picked with per-lane masks (q == v) and (a & b) | c ops, one funnel shift per word
            const int q = (back - 1) >> 5;
            const uint32_t sh = (uint32_t)(-back) & 31u;
            uint32_t x0[NW][4], x1[NW][4];
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int c = 0; c < 4; ++c) x0[w][c] = x1[w][c] = 0u;
#pragma unroll
            for (int v = -1; v <= NW - 1; ++v) {
                const uint32_t sel = q == v ? ~0u : 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (w - v - 1 >= 0 && w - v - 1 < NW) x0[w][c] |= pl[w - v - 1 >= 0 && w - v - 1 < NW ? w - v - 1 : 0][c] & sel;
                        if (w - v >= 0 && w - v < NW) x1[w][c] |= pl[w - v >= 0 && w - v < NW ? w - v : 0][c] & sel;
                    }
            }
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int c = 0; c < 4; ++c) pl[w][c] = __builtin_amdgcn_alignbit(x1[w][c], x0[w][c], sh);
        }
        uint32_t twp[4];                                          // the last 32 positions of every plane
        {
            const int sh = n & 31;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                twp[q] = sh == 0 ? pl[NW - 1][q] : __builtin_amdgcn_alignbit(pl[NW - 1][q], NW >= 2 ? pl[NW >= 2 ? NW - 2 : 0][q] : 0u, (uint32_t)sh);
        }
#ifdef ATR_SPEC
        const PieceScan S = piece_scan_spec<NW>(pl, twp, mf, T, u.k, hpl);
#else
        const PieceScan S = piece_scan<NW, (WW == 3 ? PIECE_NB : 5)>(pp, pl, twp, n, mf, T, u.k, hpl);
#endif
        const PieceTask pt = piece_task(S, back, nr, u.sr, u.m, u.k, piece_uniform(pp.head_cols));
        // the adapter verbatim: resolved here (_align.pyx:456-458), no pass B (22 % of C2's reads)
        const bool exact = live && S.j_exact != 0 && u.m >= u.min_overlap;
        if (exact) {
            const int j = S.j_exact - back;                       // in the read's own columns
            out[r] = make_uint4((uint32_t)u.m << 16, (uint32_t)(j - u.m) | ((uint32_t)j << 16), (uint32_t)u.m, 0u);
        }
        const bool flagged = live && pt.flagged && !exact;
        if (piece_uniform(pp.aonly) != 0) {                       // (wave-uniform; a constant of the run-time compiled kernel)
            // pass A only: None / the adapter verbatim decided above, every other read to the window DP with pass A's columns
            if (live && !flagged && !exact) out[r] = make_uint4(0xFFFF0000u, 0u, 0u, 0u);
            const int wlo = pt.full ? 0 : pt.j_e - pt.need, whi = pt.full ? nr : pt.j_e;
            const uint32_t ww = flagged ? window_word(wlo, whi, whi == nr, u.m, false) : 0u;
            const uint32_t none[4] = {0xFFFF0000u, 0u, 0u, 0u};
            piece_emit(flagged, r, ww, none, u.m, out, list, ldata, nullptr, 0, &s_lcur, s_hist);
            continue;
        }
        const int need = pt.need;                                 // (columns before the read: nothing to sweep)
        const bool narrow = flagged && !pt.full && need <= piece_uniform(pp.narrow), wide = flagged && !narrow;
        if (live && !flagged && !exact) out[r] = make_uint4(0xFFFF0000u, 0u, 0u, 0u);   // None
        // (the task carries the read's own columns: a ragged batch was scanned moved to the end of its words)
        const uint32_t meta = (uint32_t)pt.j_e | ((uint32_t)need << 10) | (RAGGED ? (uint32_t)nr << 17 : 0u);
        {
            // reads that need the full sweep (1.3 % on C2, but SOME lane of more than half the tiles): into the block's
            // list; the sweep at the end of the kernel gathers them from the batch (40 MB of sectors on C2).  Round 4
            // stored a copy of their planes from pass A's registers, which kept 20 registers alive to this point.
            const uint64_t wm = __ballot(wide);
            if (wm != 0ull) {                                                 // wave-uniform
                uint32_t base = 0u;
                if (lane == 0) base = atomicAdd(&s_wcnt, (uint32_t)__popcll(wm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (wide) wlist[(long long)base + lane_rank(wm)] = (uint32_t)r;
            }
        }
#if !ATR_PIECE_PREFETCH_EARLY && !defined(ATR_SPEC_ASCII)
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 4 < t1) {                                                  // the next tile's planes
            const uint4 *tp = planes + (size_t)(tile + 4) * NW * 64;
#pragma unroll
            for (int w = 0; w < NW; ++w) nx[w] = tp[w * 64 + lane];
        }
#endif
        // ---- queue the narrow lanes; 64 tasks -> pass B ----
        {
            const uint64_t nm = __ballot(narrow);
            const int cnt = (int)__popcll(nm), rank = lane_rank(nm);
            const int room = 64 - qn;
            const auto put = [&](int slot) {
                queue[0][slot] = (uint32_t)r; queue[1][slot] = meta;
                if constexpr (STASH) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) stash[w][slot] = make_uint4(pl[w][0], pl[w][1], pl[w][2], pl[w][3]);
                }
            };
            if (narrow && rank < room) put(qn + rank);
            if (cnt >= room) {                                                // wave-uniform
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                pass_b(64);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (narrow && rank >= room) put(rank - room);
                qn = cnt - room;
            } else {
                qn += cnt;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (qn > 0) pass_b(qn);
    // ---- the full sweep of the block's listed reads, one read per lane, the batches dealt round the four waves ----
    __threadfence_block();
    __syncthreads();
    {
        const long long total = (long long)s_wcnt;
        for (long long base = (long long)wave * 64; base < total; base += 256) {
            const bool act = base + lane < total;
            const long long r = act ? (long long)wlist[base + lane] : t0 * 64;
            const uint4 *bp = planes + ((size_t)(r >> 6) * NW) * 64 + (r & 63);       // the read in the batch
            const uint4 *tp = bp;
            const size_t tstride = 64;
            // (a ragged batch: a lane stops at its own last column, the wave at the longest read's)
            const int nl = RAGGED ? (act ? min(max(lens[r], 0), max_len) : 0) : max_len;
            const int nhi = RAGGED ? wave_max_i32(nl) : max_len;
            const auto full_sweep = [&](auto wc) __attribute__((always_inline)) {
            constexpr bool WD = decltype(wc)::value;
            FilterState F;
            filter_init(F, u, mf, WD);
            // (cold code, a few reads per block: real loops -- unrolled over the run-time read length the compiler kept a
            // compare result per column in scalar registers and spilled them)
            int j = 0;
#pragma clang loop unroll(disable)
            for (int c = 0; c < NW; ++c) {
                if (j >= nhi) break;                                           // wave-uniform
                const uint4 v = tp[(size_t)c * tstride];
#pragma clang loop unroll(disable)
                for (int d = 0; d < 4; ++d) {
                    if (j >= nhi) break;                                       // wave-uniform
                    uint2 e[8];
                    fetch_peq8(s_peq, piece_nibbles(s_spread, v.x, v.y, v.z, v.w, d), e);
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (j + b < nl) filter_step<WD>(F, e[b].x, e[b].y, kreg);
                    j += 8;
                }
                filter_fold(F, min(j, nl), mf, kreg);
            }
            uint32_t rec[4];
            const auto tm = [&](int jp) {
                return filter_tail_cmp(fp, T, jp, [&](int z) { return read_dword_planes((const uint32_t *)bp, NW, z, s_spread); });
            };
            const uint32_t ww = filter_decide_tm<WD>(F, u, fp, tm, nl, rec, 0, true);
            piece_emit(act, r, ww, rec, u.m, out, list, ldata, nullptr, 0, &s_lcur, s_hist);
            };
            if (mf > 32) {
                if constexpr (WIDE_OK) full_sweep(PieceBool<true>{});
            } else {
                full_sweep(PieceBool<false>{});
            }
        }
    }
    __syncthreads();
    fused_hist_flush(wk, s_hist);                                  // (locate_fast.hpp: offsets inside the bins by atomics, no scan launches)
    if (threadIdx.x == 0) wk.lcount[blockIdx.x] = s_lcur;
}

}  // namespace atr
#endif
