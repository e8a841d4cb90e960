// pairs_wave.hip -- pairs_wave_kernel: Aligner.locate with a per-pair reference for SHORT batches, one pair per
// wavefront (wave_core.hpp, R rows per lane).  The lane-per-pair kernels of pairs_kernel.hip sweep a 150 x 150 matrix
// as ONE chain of 150 k dependent operations: a call over the 1000 pairs the unchanged trim command hands to
// MergeOverlapping costs 0.36 ms whatever the batch size; here it is 150 + 51 steps of ~25 instructions.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <type_traits>
#include "atropos_hip.h"
#include "pairs_core.hpp"
#include "wave_core.hpp"

namespace atr {

__device__ __forceinline__ uint32_t pw_shr1(uint32_t v, uint32_t keep) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int pw_max_key(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

constexpr int PW_PAD = 64;                                        // LDS entries in front of the query codes, 2 x behind
constexpr uint32_t PW_CAPW = (uint32_t)PAIRS_ORG_BIAS | ((uint32_t)INIT_COST_CAP << CSH);
constexpr uint32_t PW_REBIAS = (uint32_t)(PAIRS_ORG_BIAS - (int)ORG_BIAS);

// the 32 codes of a packed chunk as dwords
__device__ __forceinline__ void pw_unpack(const uint4 v, uint32_t (&c)[32]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int b = 0; b < 8; ++b) c[8 * d + b] = (w[d] >> (4 * b)) & 15u;
}

template <bool XREP, bool SQ, int R, bool AND_MODE>
__global__ __launch_bounds__(64) void pairs_wave_kernel(const PairParams p, const uint4 *__restrict__ ref_packed,
                                                        const int32_t *__restrict__ ref_lens, int ref_chunks, int ref_max_len,
                                                        int revcomp, const uint4 *__restrict__ qry_packed,
                                                        const int32_t *__restrict__ qry_lens, int qry_chunks, int qry_max_len,
                                                        const uint8_t *__restrict__ ref_codes, const uint8_t *__restrict__ qry_codes,
                                                        long long npairs, uint4 *__restrict__ out) {
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3];
    __shared__ __attribute__((aligned(16))) uint32_t s_code[PW_PAD + (PAIRS_MAX_LEN + 31) / 32 * 32 + 2 * PW_PAD];
    __shared__ __attribute__((aligned(16))) uint32_t s_ref[(PAIRS_MAX_LEN + 31) / 32 * 32 + 32];   // s_ref[i]: code of row i + 1

    const int lane = threadIdx.x;
    const long long r = blockIdx.x;
    const int m = __builtin_amdgcn_readfirstlane(min(ref_lens ? ref_lens[r] : ref_max_len, ref_max_len));
    const int n = __builtin_amdgcn_readfirstlane(min(qry_lens ? qry_lens[r] : qry_max_len, qry_max_len));
    const bool sr = (p.flags & ATR_START_WITHIN_SEQ1) != 0, er = (p.flags & ATR_STOP_WITHIN_SEQ1) != 0;
    int k = (int)(p.e * m);                                            // _align.pyx:312
    if (k < 0) k = -1;
    int indel = p.indel_cost > k ? k + 1 : p.indel_cost;
    if (indel < 1) indel = 1;
    const uint32_t insw = (uint32_t)indel * COST1 + PRIO_INS, delw = (uint32_t)indel * COST1 + PRIO_DEL;
    const uint32_t klimit = (uint32_t)(k + 1) << CSH;
    const int max_n = SQ ? n : min(n, m + k);                          // :314-321
    const int min_n = XREP ? 0 : max(0, n - m - k);
    const int span = max(0, max_n - min_n);
    const bool scan = max_n == n;                                      // :461 (an empty matrix: the initial column)

    {   // thresholds from the kernel-argument segment (`p` is the first argument; see wave_kernel.hip)
        const int16_t *kthr = (const int16_t *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PairParams, thr));
        for (int i = lane; i < PAIRS_MAX_LEN + 3; i += 64) s_thr[i] = kthr[i];
    }
    if (ref_codes) {
        // one 4-bit code per byte, row r of each matrix (row stride = the layout width): atr_locate_pair_one hands the
        // translated strings over in page-locked host memory
        const uint8_t *rr = ref_codes + (size_t)r * ref_max_len, *qq = qry_codes + (size_t)r * qry_max_len;
        for (int i = lane; i < n; i += 64) s_code[PW_PAD + i] = qq[i] & 15u;
        for (int i = lane; i < m; i += 64) {
            const uint32_t c = rr[i] & 15u;
            if (revcomp) s_ref[m - 1 - i] = bitrev4(c); else s_ref[i] = c;
        }
    } else {
    if (lane < (n + 31) / 32) {
        uint32_t c[32];
        pw_unpack(qry_packed[((size_t)(r >> 6) * qry_chunks + lane) * 64 + (size_t)(r & 63)], c);
        uint4 *dst = (uint4 *)(s_code + PW_PAD + 32 * lane);
#pragma unroll
        for (int g = 0; g < 8; ++g) dst[g] = make_uint4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
    }
    if (lane < (m + 31) / 32) {
        uint32_t c[32];
        pw_unpack(ref_packed[((size_t)(r >> 6) * ref_chunks + lane) * 64 + (size_t)(r & 63)], c);
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const int i = 32 * lane + b;                                // 0-based base of the packed sequence
            if (i < m) {
                if (revcomp) s_ref[m - 1 - i] = bitrev4(c[b]);          // complement of a DNA / IUPAC bit code = its bit reversal
                else s_ref[i] = c[b];
            }
        }
    }
    }
    __syncthreads();

    const WaveGeom g = wave_geom(m, R);
    WaveRows<R> W;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int row = wave_slot_row(g, R, lane, rr);
        uint32_t mask = 0u;
        if (row >= 1 && row <= m) {
            const uint32_t c = s_ref[row - 1];
            if (AND_MODE) {
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) mask |= ((c & q) == 0u ? 1u : 0u) << q;     // _align.pyx:392-393
            } else {
                mask = ~(1u << c) & 0xFFFFu;                                               // :390-391 on 4-bit codes
            }
        }
        W.rowmask[rr] = mask;
        W.lstep[rr] = row < 0 ? 0u : row == 0 ? (SQ ? 1u : (uint32_t)indel << CSH) : delw;
        W.col[rr] = row < 0 ? WAVE_HUGE : init_word(row, min_n, sr, SQ, indel) + PW_REBIAS;
    }
    uint32_t upa = pw_shr1(W.col[R - 1], WAVE_HUGE), upb = WAVE_HUGE;     // upa: the diagonal input of step 1
    int a = min_n - lane - 1;                                            // 0-based query position of this lane's column, before step 1
    const uint32_t *code = s_code + PW_PAD;
    Best best;
    best.key = COST_FIELD_MAX - (m + n);                                 // (matches 0, cost m + n): :358-363
    best.word = (uint32_t)(m + n) << CSH;
    best.ref_stop = m; best.query_stop = n; best.matches = 0;
    const bool rowm = lane == g.lanes - 1;                               // row m: this lane's bottom row
    const int steps = span > 0 ? span + g.lanes - 1 : 0;

    // One trip = four steps (wave_kernel.hip: eight; a step is R times as long here).  GUARDED: the ramps.
    uint32_t q[4], qn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) q[s] = code[a + 1 + s];
    auto trip = [&](auto guarded_tag) {
        constexpr bool GUARDED = decltype(guarded_tag)::value;
        const int a0 = a;
#pragma unroll
        for (int s = 0; s < 4; ++s) qn[s] = code[a0 + 5 + s];
        __builtin_amdgcn_sched_barrier(0);
        uint32_t bottom[4];
        bool hit[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t &up = (s & 1) ? upa : upb, &diag = (s & 1) ? upb : upa;
            up = pw_shr1(W.col[R - 1], up);
            uint32_t nw[R];
            wave_rows_step<XREP, SQ, R, PW_CAPW>(W, diag, up, q[s], insw, nw);
            bottom[s] = nw[R - 1];
            if (GUARDED) {
                ++a;
                const bool active = (unsigned)(a - min_n) < (unsigned)span;
                hit[s] = XREP && rowm && active && nw[R - 1] < klimit;   // row-m candidate (:433-455)
#pragma unroll
                for (int rr = 0; rr < R; ++rr) W.col[rr] = active ? nw[rr] : W.col[rr];
            } else {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) W.col[rr] = nw[rr];
            }
        }
        if (GUARDED) {
            if (XREP && (hit[0] | hit[1] | hit[2] | hit[3])) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (hit[s]) consider<XREP, PAIRS_ORG_BIAS>(best, bottom[s], m, a0 + 2 + s, p.min_overlap, s_thr, indel);
            }
        } else {
            a += 4;
            const uint32_t least = min(min(bottom[0], bottom[1]), min(bottom[2], bottom[3]));
            if (XREP && rowm && least < klimit) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (bottom[s] < klimit) consider<XREP, PAIRS_ORG_BIAS>(best, bottom[s], m, a0 + 2 + s, p.min_overlap, s_thr, indel);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) q[s] = qn[s];
    };
    int t = 1;
    for (; t <= steps && t <= g.lanes - 1; t += 4) trip(std::true_type{});   // ramp up
    for (; t + 3 <= span; t += 4) trip(std::false_type{});                    // lanes - 1 < t .. t + 3 <= span: every lane in use is active
    for (; t <= steps; t += 4) trip(std::true_type{});                        // ramp down

    // candidates in the reference's order: row m by column (lane L - 1), then the last column by row
    Best fin;
    fin.key = __builtin_amdgcn_readlane(best.key, g.lanes - 1);
    fin.word = (uint32_t)__builtin_amdgcn_readlane((int)best.word, g.lanes - 1);
    fin.ref_stop = __builtin_amdgcn_readlane(best.ref_stop, g.lanes - 1);
    fin.query_stop = __builtin_amdgcn_readlane(best.query_stop, g.lanes - 1);
    fin.matches = __builtin_amdgcn_readlane(best.matches, g.lanes - 1);
    if (scan) {
        const int first_row = er ? 0 : m;
        Best mine;
        mine.key = -1; mine.word = 0; mine.ref_stop = 0; mine.query_stop = n; mine.matches = 0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {                                  // ascending rows: the first of equal keys stays
            const int row = wave_slot_row(g, R, lane, rr);
            if (row >= first_row && row <= m)
                consider<XREP, PAIRS_ORG_BIAS>(mine, W.col[rr], row, n, p.min_overlap, s_thr, indel);
        }
        const int top = pw_max_key(mine.key < 0 ? -1 : (mine.key << 6) | (63 - lane));
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
        const int cost = (int)(fin.word >> CSH);
        int refstart = 0, querystart = 0, refstop = -1, querystop = 0, matches = 0, errors = 0;
        if (cost != m + n) {                                                // :476-480
            const int origin = (int)(fin.word & ORG_MASK) - PAIRS_ORG_BIAS;
            if (origin >= 0) querystart = origin; else refstart = -origin;
            refstop = fin.ref_stop; querystop = fin.query_stop;
            matches = fin.matches; errors = cost;
        }
        out[r] = make_uint4((uint32_t)(refstart & 0xFFFF) | ((uint32_t)(refstop & 0xFFFF) << 16),
                            (uint32_t)(querystart & 0xFFFF) | ((uint32_t)(querystop & 0xFFFF) << 16),
                            (uint32_t)(matches & 0xFFFF) | ((uint32_t)(errors & 0xFFFF) << 16), 0u);
    }
}

template <int R>
static void launch_pw_r(const PairParams &p, const uint4 *rp, const int32_t *rl, int rch, int rmax, int revcomp, const uint4 *qp,
                        const int32_t *ql, int qch, int qmax, const uint8_t *rc, const uint8_t *qc, long long npairs, uint4 *out,
                        hipStream_t st) {
    const dim3 grid((unsigned)npairs), block(64);
    const bool xrep = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0, sq = (p.flags & ATR_START_WITHIN_SEQ2) != 0, am = p.and_mode != 0;
#define ATR_PW_LAUNCH(X, S, A) hipLaunchKernelGGL((pairs_wave_kernel<X, S, R, A>), grid, block, 0, st, p, rp, rl, rch, rmax, revcomp, qp, ql, qch, qmax, rc, qc, npairs, out)
    if (xrep && sq) { if (am) ATR_PW_LAUNCH(true, true, true); else ATR_PW_LAUNCH(true, true, false); }
    else if (xrep) { if (am) ATR_PW_LAUNCH(true, false, true); else ATR_PW_LAUNCH(true, false, false); }
    else if (sq) { if (am) ATR_PW_LAUNCH(false, true, true); else ATR_PW_LAUNCH(false, true, false); }
    else { if (am) ATR_PW_LAUNCH(false, false, true); else ATR_PW_LAUNCH(false, false, false); }
#undef ATR_PW_LAUNCH
}

// every pair of the batch on a wavefront of its own; R from the longest reference.  rp == nullptr: the sides as code
// bytes (ref_codes / qry_codes, rows of rmax / qmax bytes).
hipError_t launch_pairs_wave(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                             const uint32_t *qp, const int32_t *ql, int qmax, const uint8_t *ref_codes, const uint8_t *qry_codes,
                             long long npairs, uint4 *out, hipStream_t st) {
    const int rch = (rmax + 31) / 32, qch = (qmax + 31) / 32;
#define ATR_PW_R(R) launch_pw_r<R>(p, (const uint4 *)rp, rl, rch, rmax, revcomp, (const uint4 *)qp, ql, qch, qmax, ref_codes, qry_codes, npairs, out, st)
    switch (wave_pair_rows(rmax)) {
        case 1: ATR_PW_R(1); break;
        case 2: ATR_PW_R(2); break;
        case 3: ATR_PW_R(3); break;
        case 4: ATR_PW_R(4); break;
        case 5: ATR_PW_R(5); break;
        default: return hipErrorNotSupported;
    }
#undef ATR_PW_R
    return hipGetLastError();
}

}  // namespace atr
