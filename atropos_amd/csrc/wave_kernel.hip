// wave_kernel.hip -- locate_wave_kernel: one read per wavefront, anti-diagonal sweep (wave_core.hpp).
#include <hip/hip_runtime.h>
#include <cstddef>
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

template <bool XREP, bool SQ>
__global__ __launch_bounds__(64) void locate_wave_kernel(const LocateParams p, const uint4 *__restrict__ packed,
                                                         const int32_t *__restrict__ lens, long long nreads, int nchunks,
                                                         int max_len, uint4 *__restrict__ out) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ __attribute__((aligned(16))) uint32_t s_code[WAVE_CODE_PAD + (ATR_MAX_READ_LEN + 31) / 32 * 32 + 2 * WAVE_CODE_PAD];

    const int lane = threadIdx.x;
    const long long r = blockIdx.x;
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    const int n = __builtin_amdgcn_readfirstlane(lens ? lens[r] : max_len);
    const WaveWindow win = wave_window<XREP>(u, n);

    // thresholds and the read's codes (one dword per column) into LDS
    {   // (read from the kernel-argument segment itself -- `p` is the first argument: indexing the by-value copy with
        //  a run-time index would move all of it to scratch memory)
        const int16_t *kthr = (const int16_t *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LocateParams, thr));
        for (int i = lane; i <= u.m + 1; i += 64) s_thr[i] = kthr[i];
    }
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
    const uint32_t rowmask = lane > 0 ? wave_rowmask(p, u.p0, lane) : 0u;
    const uint32_t left_step = wave_left_step(u, lane);
    __syncthreads();

    // cells of the initial column (min_n); lane 0: row 0.  `up` lives in two registers used in turn, so that the
    // one a DPP move writes is the one whose lane 0 still holds WAVE_HUGE.
    uint32_t cur = init_word(lane, win.min_n, u.sr, u.sq, u.indel);
    uint32_t upa = wave_shr1(cur, WAVE_HUGE), upb = WAVE_HUGE;       // upa: the diagonal input of step 1
    int a = win.min_n - lane - 1;                                    // 0-based query position of this lane's column, before step 1
    const uint32_t *code = s_code + WAVE_CODE_PAD;
    Best best;
    wave_best_init(best, u, n);
    const bool rowm = lane == u.m;
    const int steps = win.span > 0 ? win.span + u.m : 0;
    // One trip = eight steps.  The codes of steps 4 - 7 are fetched from LDS at the top of the trip (steps 0 - 3 hide
    // the latency), those of the next trip's steps 0 - 3 after step 3 (hidden by steps 4 - 7).
    // GUARDED: with the per-lane activity test (the ramps: some rows have not started yet or are done).
    uint32_t q[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) q[s] = code[a + 1 + s];
    auto trip = [&](auto guarded_tag) {
        constexpr bool GUARDED = decltype(guarded_tag)::value;
        const int a0 = a;
#pragma unroll
        for (int s = 4; s < 8; ++s) q[s] = code[a0 + 1 + s];
        __builtin_amdgcn_sched_barrier(0);                           // (the loads stay where they are written)
        uint32_t cell[8];
        bool hit[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint32_t &up = (s & 1) ? upa : upb, &diag = (s & 1) ? upb : upa;
            up = wave_shr1(cur, up);
            const uint32_t nw = wave_cell<XREP, SQ>(diag, cur, up, rowmask, q[s], u.insw, left_step);
            cell[s] = nw;
            if (GUARDED) {
                ++a;
                const bool active = (unsigned)(a - win.min_n) < (unsigned)win.span;
                hit[s] = XREP && rowm && active && nw < u.klimit;    // row-m candidate (:433-455)
                cur = active ? nw : cur;
            } else {
                cur = nw;
            }
            if (s == 3) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = code[a0 + 9 + k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (GUARDED) {
            if (XREP && (hit[0] | hit[1] | hit[2] | hit[3] | hit[4] | hit[5] | hit[6] | hit[7])) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (hit[s]) consider<XREP>(best, cell[s], u.m, a0 + 2 + s, u.min_overlap, s_thr, u.indel);
            }
        } else {
            a += 8;
            const uint32_t least = min(min(min(cell[0], cell[1]), min(cell[2], cell[3])), min(min(cell[4], cell[5]), min(cell[6], cell[7])));
            if (XREP && rowm && least < u.klimit) {                  // lane m only, and rarely
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (cell[s] < u.klimit) consider<XREP>(best, cell[s], u.m, a0 + 2 + s, u.min_overlap, s_thr, u.indel);
            }
        }
    };
    int t = 1;                                                       // first step of the next trip
    for (; t <= steps && t <= u.m; t += 8) trip(std::true_type{});  // ramp up: rows start one by one
    for (; t + 7 <= win.span; t += 8) trip(std::false_type{});      // m < t .. t + 7 <= span: every row 0 .. m is active
    for (; t <= steps; t += 8) trip(std::true_type{});              // ramp down (steps beyond `steps`: no lane is active)

    // the candidates in the reference's order: row m by column (lane m), then the last column by row
    Best fin;
    fin.key = __builtin_amdgcn_readlane(best.key, u.m);
    fin.word = (uint32_t)__builtin_amdgcn_readlane((int)best.word, u.m);
    fin.ref_stop = __builtin_amdgcn_readlane(best.ref_stop, u.m);
    fin.query_stop = __builtin_amdgcn_readlane(best.query_stop, u.m);
    fin.matches = __builtin_amdgcn_readlane(best.matches, u.m);
    if (win.scan) {
        Best mine;
        const int key = wave_last_key<XREP>(cur, lane, lane, lane >= (u.er ? 0 : u.m) && lane <= u.m, u, n, s_thr, mine);
        const int top = wave_max_key(key);
        if (top >= 0 && (top >> 6) > fin.key) {
            const int src = 63 - (top & 63);
            fin.key = top >> 6;
            fin.word = (uint32_t)__builtin_amdgcn_readlane((int)mine.word, src);
            fin.ref_stop = src;
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

int launch_locate_wave(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                       int max_len, uint4 *out, hipStream_t st) {
    const dim3 grid((unsigned)nreads), block(64);
    const bool xrep = (a->flags & ATR_STOP_WITHIN_SEQ2) != 0, sq = (a->flags & ATR_START_WITHIN_SEQ2) != 0;
    if (xrep && sq)  hipLaunchKernelGGL((locate_wave_kernel<true, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    if (xrep && !sq) hipLaunchKernelGGL((locate_wave_kernel<true, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    if (!xrep && sq) hipLaunchKernelGGL((locate_wave_kernel<false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    if (!xrep && !sq) hipLaunchKernelGGL((locate_wave_kernel<false, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    return (int)hipGetLastError();
}

}  // namespace atr
