// wave_kernel.hip -- locate_wave_kernel: one read per wavefront, anti-diagonal sweep (wave_core.hpp).
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstring>
#include <type_traits>
#include "aligner_host.hpp"
#include "wave_core.hpp"

namespace atr {

// value of lane - 1; lane 0 keeps what `keep` holds there (v_mov_b32_dpp wave_shr:1, bound_ctrl off)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t keep) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int wave_max_key(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

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
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    const int n = __builtin_amdgcn_readfirstlane(lens ? min(max(lens[r], 0), max_len) : max_len);    // (never beyond the layout)
    const WaveWindow win = wave_window<XREP>(u, n);

    // thresholds and the read's codes (one dword per column) into LDS
    {   // (read from the kernel-argument segment itself -- `p` is the first argument: indexing the by-value copy with
        //  a run-time index would move all of it to scratch memory)
        const int16_t *kthr = (const int16_t *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WaveParams, p) +
                                                offsetof(LocateParams, thr));
        for (int i = lane; i <= u.m + 1; i += 64) s_thr[i] = kthr[i];
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
    // R rows per lane, bottom-aligned: row m is the bottom slot of lane g.lanes - 1, the slots above row 0 are padding
    const WaveGeom g = wave_geom(u.m, R);
    WaveRows<R> W;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int row = wave_slot_row(g, R, lane, rr);
        W.rowmask[rr] = (row >= 1 && row <= u.m) ? wave_rowmask(p, u.p0, row) : 0u;
        W.lstep[rr] = row < 0 ? 0u : row == 0 ? (SQ ? 1u : (uint32_t)u.indel << CSH) : u.delw;
        W.col[rr] = row < 0 ? WAVE_HUGE : init_word(row, win.min_n, u.sr, SQ, u.indel);   // cells of the initial column (min_n)
    }
    __syncthreads();

    // `up` lives in two registers used in turn, so that the one a DPP move writes is the one whose lane 0 still
    // holds WAVE_HUGE.
    uint32_t upa = wave_shr1(W.col[R - 1], WAVE_HUGE), upb = WAVE_HUGE;   // upa: the diagonal input of step 1
    int a = win.min_n - lane - 1;                                    // 0-based query position of this lane's column, before step 1
    const uint32_t *code = s_code + WAVE_CODE_PAD;
    Best best;
    wave_best_init(best, u, n);
    const bool rowm = lane == g.lanes - 1;
    const int steps = win.span > 0 ? win.span + g.lanes - 1 : 0;
    // One trip = TRIP steps (eight of one row, four of more).  The codes of a trip's second half are fetched from LDS at
    // its top (the first half hides the latency), those of the next trip's first half after this one's.
    // GUARDED: with the per-lane activity test (the ramps: some rows have not started yet or are done).
    constexpr int TRIP = R == 1 ? 8 : 4, HALF = TRIP / 2;
    uint32_t q[TRIP];
#pragma unroll
    for (int s = 0; s < HALF; ++s) q[s] = code[a + 1 + s];
    auto trip = [&](auto guarded_tag) {
        constexpr bool GUARDED = decltype(guarded_tag)::value;
        const int a0 = a;
#pragma unroll
        for (int s = HALF; s < TRIP; ++s) q[s] = code[a0 + 1 + s];
        __builtin_amdgcn_sched_barrier(0);                           // (the loads stay where they are written)
        uint32_t bottom[TRIP];
        bool hit[TRIP];
#pragma unroll
        for (int s = 0; s < TRIP; ++s) {
            uint32_t &up = (s & 1) ? upa : upb, &diag = (s & 1) ? upb : upa;
            up = wave_shr1(W.col[R - 1], up);
            uint32_t nw[R];
            wave_rows_step<XREP, SQ, R, WAVE_ROW0_CAP>(W, diag, up, q[s], u.insw, nw);
            bottom[s] = nw[R - 1];
            if (GUARDED) {
                ++a;
                const bool active = (unsigned)(a - win.min_n) < (unsigned)win.span;
                hit[s] = XREP && rowm && active && nw[R - 1] < u.klimit;    // row-m candidate (:433-455)
#pragma unroll
                for (int rr = 0; rr < R; ++rr) W.col[rr] = active ? nw[rr] : W.col[rr];
            } else {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) W.col[rr] = nw[rr];
            }
            if (s == HALF - 1) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < HALF; ++k) q[k] = code[a0 + TRIP + 1 + k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (GUARDED) {
            bool any = false;
#pragma unroll
            for (int s = 0; s < TRIP; ++s) any = any | hit[s];
            if (XREP && any) {
#pragma unroll
                for (int s = 0; s < TRIP; ++s)
                    if (hit[s]) consider<XREP>(best, bottom[s], u.m, a0 + 2 + s, u.min_overlap, s_thr, u.indel);
            }
        } else {
            a += TRIP;
            uint32_t least = bottom[0];
#pragma unroll
            for (int s = 1; s < TRIP; ++s) least = min(least, bottom[s]);
            if (XREP && rowm && least < u.klimit) {                  // lane L - 1 only, and rarely
#pragma unroll
                for (int s = 0; s < TRIP; ++s)
                    if (bottom[s] < u.klimit) consider<XREP>(best, bottom[s], u.m, a0 + 2 + s, u.min_overlap, s_thr, u.indel);
            }
        }
    };
    int t = 1;                                                       // first step of the next trip
    for (; t <= steps && t <= g.lanes - 1; t += TRIP) trip(std::true_type{});    // ramp up: lanes start one by one
    for (; t + TRIP - 1 <= win.span; t += TRIP) trip(std::false_type{});         // every lane in use is active
    for (; t <= steps; t += TRIP) trip(std::true_type{});                         // ramp down (beyond `steps`: no lane is active)

    // the candidates in the reference's order: row m by column (lane L - 1), then the last column by row
    Best fin;
    fin.key = __builtin_amdgcn_readlane(best.key, g.lanes - 1);
    fin.word = (uint32_t)__builtin_amdgcn_readlane((int)best.word, g.lanes - 1);
    fin.ref_stop = __builtin_amdgcn_readlane(best.ref_stop, g.lanes - 1);
    fin.query_stop = __builtin_amdgcn_readlane(best.query_stop, g.lanes - 1);
    fin.matches = __builtin_amdgcn_readlane(best.matches, g.lanes - 1);
    if (win.scan) {
        const int first_row = u.er ? 0 : u.m;
        Best mine;
        mine.key = -1; mine.word = 0; mine.ref_stop = 0; mine.query_stop = n; mine.matches = 0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {                              // ascending rows: the first of equal keys stays
            const int row = wave_slot_row(g, R, lane, rr);
            if (row >= first_row && row <= u.m && W.col[rr] < u.klimit)
                consider<XREP>(mine, W.col[rr], row, n, u.min_overlap, s_thr, u.indel);
        }
        const int top = wave_max_key(mine.key < 0 ? -1 : (mine.key << 6) | (63 - lane));
        if (top >= 0 && (top >> 6) > fin.key) {
            const int src = 63 - (top & 63);
            fin.key = top >> 6;
            fin.word = (uint32_t)__builtin_amdgcn_readlane((int)mine.word, src);
            fin.ref_stop = __builtin_amdgcn_readlane(mine.ref_stop, src);
            fin.query_stop = n;
            fin.matches = __builtin_amdgcn_readlane(mine.matches, src);
        }
    }
    if (lane == 0) {
        uint32_t rec[4];
        wave_result(fin, u, n, rec);
        out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
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
