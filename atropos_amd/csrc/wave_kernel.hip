// wave_kernel.hip -- locate_wave_kernel: one read per wavefront, anti-diagonal sweep (wave_core.hpp).
#include <hip/hip_runtime.h>
#include <cstddef>
#include "aligner_host.hpp"
#include "wave_core.hpp"

namespace atr {

// value of lane - 1; lane 0 takes `first` (v_mov_b32_dpp wave_shr:1, bound_ctrl off: lane 0 keeps `old`)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int wave_max_key(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

template <bool XREP>
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
    const uint32_t rowmask = wave_rowmask(p, u.p0, lane);
    __syncthreads();

    // cells of the initial column (min_n), the diagonal input of the first step
    uint32_t cur = init_word(lane + 1, win.min_n, u.sr, u.sq, u.indel);
    // row 0 by column (wave_row0), incrementally: origin + 1 per column with START_WITHIN_SEQ2, else cost + indel, saturated
    const uint32_t r0_inc = u.sq ? 1u : (uint32_t)u.indel << CSH;
    const uint32_t r0_cap = u.sq ? 0xFFFFFFFFu : (ORG_BIAS | ((uint32_t)INIT_COST_CAP << CSH));
    uint32_t r0 = wave_row0(u, win.min_n);
    uint32_t diag = wave_shr1(cur, r0);
    int a = win.min_n - lane - 1;                                    // 0-based query position of this lane's column, before step 1
    const uint32_t *code = s_code + WAVE_CODE_PAD;
    Best best;
    wave_best_init(best, u, n);
    const bool rowm = lane == u.m - 1;
    const int steps = win.span > 0 ? win.span + u.m - 1 : 0;
    // Four steps per trip: the codes of a group are fetched from LDS while the group before it runs, and the row-m
    // cells of a group are looked at together (one branch per group; only lane m - 1 ever takes it).  Steps beyond
    // `steps` change nothing: no lane is active there.
    uint32_t q[4], qn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) q[s] = code[a + 1 + s];
    for (int t = 0; t < steps; t += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qn[s] = code[a + 5 + s];
        __builtin_amdgcn_sched_barrier(0);                           // (the loads stay up here: a group of steps hides them)
        uint32_t cell[4];
        bool hit[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ++a;
            r0 = min(r0 + r0_inc, r0_cap);
            const uint32_t up = wave_shr1(cur, r0);
            const uint32_t nw = wave_cell<XREP>(diag, cur, up, rowmask, q[s], u.insw, u.delw);
            const bool active = (unsigned)(a - win.min_n) < (unsigned)win.span;
            diag = up;
            cell[s] = nw;
            hit[s] = XREP && rowm && active && nw < u.klimit;        // row-m candidate (:433-455)
            cur = active ? nw : cur;
        }
        if (XREP && (hit[0] | hit[1] | hit[2] | hit[3])) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (hit[s]) consider<XREP>(best, cell[s], u.m, a - 2 + s, u.min_overlap, s_thr, u.indel);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) q[s] = qn[s];
    }

    // the candidates in the reference's order: row m by column (lane m - 1), then the last column by row
    Best fin;
    fin.key = __builtin_amdgcn_readlane(best.key, u.m - 1);
    fin.word = (uint32_t)__builtin_amdgcn_readlane((int)best.word, u.m - 1);
    fin.ref_stop = __builtin_amdgcn_readlane(best.ref_stop, u.m - 1);
    fin.query_stop = __builtin_amdgcn_readlane(best.query_stop, u.m - 1);
    fin.matches = __builtin_amdgcn_readlane(best.matches, u.m - 1);
    if (win.scan) {
        const int first_row = u.er ? 0 : u.m;
        if (first_row == 0) {
            const uint32_t w0 = wave_row0(u, win.max_n);
            if (w0 < u.klimit) consider<XREP>(fin, w0, 0, n, u.min_overlap, s_thr, u.indel);
        }
        Best mine;
        const int key = wave_last_key<XREP>(cur, lane + 1, lane, lane + 1 >= first_row && lane < u.m, u, n, s_thr, mine);
        const int top = wave_max_key(key);
        if (top >= 0 && (top >> 6) > fin.key) {
            const int src = 63 - (top & 63);
            fin.key = top >> 6;
            fin.word = (uint32_t)__builtin_amdgcn_readlane((int)mine.word, src);
            fin.ref_stop = src + 1;
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
    if (a->flags & ATR_STOP_WITHIN_SEQ2)
        hipLaunchKernelGGL(locate_wave_kernel<true>, grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    else
        hipLaunchKernelGGL(locate_wave_kernel<false>, grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    return (int)hipGetLastError();
}

}  // namespace atr
