// wave_sweep.hpp -- the anti-diagonal sweep of one read on one wavefront (wave_core.hpp), shared by
// locate_wave_kernel (wave_kernel.hip) and linked_wave_kernel (linked_kernels.hip).  Device code only.
#ifndef ATR_WAVE_SWEEP_HPP
#define ATR_WAVE_SWEEP_HPP

#include <hip/hip_runtime.h>
#include <type_traits>
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

// Aligner.locate of the query whose codes sit in LDS at code[0 .. n) (one dword per base, WAVE_CODE_PAD readable entries in
// front, twice as many behind) against the aligner `p`; s_thr: its thresholds in LDS.  Both are written by the caller
// BEFORE the call (this function holds the barrier).  rec: the result record, valid in every lane.
template <bool XREP, bool SQ, int R>
__device__ __forceinline__ void wave_locate(const LocateParams &p, const int16_t *s_thr, const uint32_t *code, int n, int lane,
                                            uint32_t rec[4]) {
    const Uniform u = make_uniform(p, round_up_rows_dev(p.m));
    const WaveWindow win = wave_window<XREP>(u, n);
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
    __syncthreads();                                                 // s_thr and the codes are in LDS

    // `up` lives in two registers used in turn, so that the one a DPP move writes is the one whose lane 0 still
    // holds WAVE_HUGE.
    uint32_t upa = wave_shr1(W.col[R - 1], WAVE_HUGE), upb = WAVE_HUGE;   // upa: the diagonal input of step 1
    int a = win.min_n - lane - 1;                                    // 0-based query position of this lane's column, before step 1
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
    wave_result(fin, u, n, rec);
}

}  // namespace atr
#endif
