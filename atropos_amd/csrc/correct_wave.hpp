// correct_wave.hpp -- the wave-level part of the plane-guided error correction (device only): the disagreeing
// positions of a wavefront's 64 pairs as (pair, position) TASKS in a per-wave LDS queue, 64 tasks at a time.
// Shared by correct_planes_kernel (misc_kernels.hip: atr_insert_correct_batch, the records come from memory)
// and by the fused insert kernels (insert_kernel.hip: atr_insert_match_correct_batch, the match was found by
// this very lane a moment ago and its reads' planes are still on the CU).
// Reference: commands/trim/modifiers.py:219-350 (correct_errors), :397-404 (which pairs are corrected).
#ifndef ATR_CORRECT_WAVE_HPP
#define ATR_CORRECT_WAVE_HPP

#include <hip/hip_runtime.h>
#include "misc_core.hpp"

namespace atr {

struct CompTable { uint8_t c[256]; };

// queue entries a wave may hold: at most 63 left over + 64 x 32 new ones
constexpr int CORRECT_QUEUE_ENTRIES = 64 * 32 + 64;
// loads in flight per lane and memory round trip: tasks of the first pass / 16-byte pieces of the tied pairs' rows
#ifndef CORRECT_TASKS_PER_LANE
#define CORRECT_TASKS_PER_LANE 4
#endif
#ifndef CORRECT_PIECES_PER_LANE
#define CORRECT_PIECES_PER_LANE 8
#endif

// The wave's LDS (views; the caller decides what they overlay).  Pointers IN the LDS address space and plain accesses
// ordered by wavefront-scope fences (wave_sync_lds): as generic volatile pointers (rounds 4 / 5) every queue entry was
// a flat_store_short followed by s_waitcnt vmcnt(0) -- some sixty store round trips per wave, a third of the tail.
#define ATR_LDS __attribute__((address_space(3)))
struct CorrectWaveLds {
    ATR_LDS uint16_t *queue;           // [CORRECT_QUEUE_ENTRIES]
    ATR_LDS uint32_t *cnt;             // [64] per pair: c1 | c2 << 10 | npend << 20
    ATR_LDS int32_t *err;              // [64]
    ATR_LDS int16_t *jv;               // [64] the pair's overlap length
    ATR_LDS uint32_t *tail;            // [1]
    ATR_LDS uint32_t *ptail;           // [1] entries of the tie list
    ATR_LDS unsigned long long *acc;   // [64] per pair: the two quality sums of its overlap, then which read wins its ties
    int qcap;                          // uint16 entries the queue's LDS holds (>= CORRECT_QUEUE_ENTRIES): the tie list grows down from its end
    ATR_LDS const uint8_t *comp;       // [256] complement table (block-wide)
    ATR_LDS const uint8_t *letter;     // [16] DNA15 code -> its byte (block-wide)
};
template <class T>
__device__ __forceinline__ ATR_LDS T *lds_view(T *p) { return (ATR_LDS T *)p; }
// the lanes of ONE wave exchange data through LDS: its LDS operations execute in order, the fences keep the compiler
// from moving (or forwarding) accesses across the point
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class T>
__device__ __forceinline__ T lds_add(ATR_LDS T *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct CorrectArgs {
    const uint4 *planes1, *planes2;    // plane64 batches of the two reads
    int nchunks;
    uint8_t *s1, *q1, *s2, *q2;        // ASCII matrices, corrected in place (q1 / q2 may be null)
    long long stride;
    int action, min_qual_diff;
    int32_t *changed, *newlen;         // [n][2]
};

// set bits of `mask` below `lane`
__device__ __forceinline__ int lane_rank_below(unsigned long long mask, int lane) {
    (void)lane;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void correct_letter_table(uint8_t *s_letter) {
    if (threadIdx.x < 16) s_letter[threadIdx.x] = (uint8_t)"\0ACMGRSVTWYHKDBN"[threadIdx.x];   // (A 1, C 2, G 4, T 8 and their unions)
}

// tile: the wave's tile of 64 pairs, p = tile * 64 + lane; todo: the pair has an insert match with errors and j is its
// overlap (read1[0:j] faces revcomp(read2[0:j])); mism[w]: the disagreeing positions of word w of read 1
// (facing_mismatches / insert_overlap_mismatches; 0 beyond nchunks).  The whole wave calls this (wave barriers inside).
template <int PW>
__device__ __forceinline__ void correct_wave_tail(const CorrectWaveLds &S, const CorrectArgs &A, long long tile, int lane,
                                                  bool live, bool todo, int j, int len1, int len2,
                                                  const uint32_t (&mism)[PW]) {
    const long long p = tile * 64 + lane;
    ATR_LDS uint16_t *queue = S.queue;
    const uint8_t *const comp = (const uint8_t *)S.comp;
    S.cnt[lane] = 0u;
    S.err[lane] = 0;
    S.jv[lane] = (int16_t)j;
    S.acc[lane] = 0ull;
    if (lane == 0) { *S.tail = 0u; *S.ptail = 0u; }
    wave_sync_lds();
    const uint32_t pend_cap = (uint32_t)(S.qcap - CORRECT_QUEUE_ENTRIES);
    const bool has_quals = A.q1 != nullptr && A.q2 != nullptr;
    const int nchunks = A.nchunks;
    // The tile's base pointers (wave-uniform: scalar registers) and 32-bit offsets from them (a pair's row, a position):
    // global_load / global_store with a scalar base and a vector offset, where 64-bit pointers per lane cost two to three
    // VALU instructions per access (the API bounds the row pitch: 64 rows of it fit 32 bits).
    const size_t tile_row0 = (size_t)(tile * 64) * (size_t)A.stride;
    uint8_t *const s1t = A.s1 + tile_row0, *const s2t = A.s2 + tile_row0;
    uint8_t *const q1t = has_quals ? A.q1 + tile_row0 : nullptr, *const q2t = has_quals ? A.q2 + tile_row0 : nullptr;
    const uint4 *const p1t = A.planes1 + (size_t)tile * nchunks * 64, *const p2t = A.planes2 + (size_t)tile * nchunks * 64;
    const uint32_t pitch = (uint32_t)A.stride;
    // A task in two halves, so that the loads of SEVERAL tasks can be in flight before the first one stores (byte
    // pointers alias: the compiler keeps every load behind the stores that precede it in program order).
    struct Task { int src, i, jx; uint32_t row; int qa, qb; uint8_t base1, raw2; bool valid; uint32_t entry; };
    const auto fetch = [&](bool valid, uint32_t t) {
        Task k;
        k.valid = valid; k.entry = t;
        k.src = (int)(t >> 9); k.i = (int)(t & 511u);
        k.jx = 0; k.row = 0; k.qa = k.qb = 0; k.base1 = k.raw2 = 0;
        if (valid) {
            k.jx = (int)S.jv[k.src] - 1 - k.i;
            k.row = (uint32_t)k.src * pitch;
            // The two BASES come from the bit planes, not from the ASCII matrices: the pair's chunks were streamed by this
            // wave a moment ago (L2), a DNA15 code names its byte (15 upper-case letters, aligner_host.hpp), and a byte
            // fetched from a matrix costs a 64-byte sector of HBM -- two of the four a task used to pull (round 4:
            // 1 068 B per pair counted).  Code 0 (a byte outside the table): read the matrix as before.
            const uint4 v1 = p1t[(uint32_t)((k.i >> 5) * 64 + k.src)];
            const uint4 v2 = p2t[(uint32_t)((k.jx >> 5) * 64 + k.src)];
            const uint32_t b1 = (uint32_t)(k.i & 31), b2s = (uint32_t)(k.jx & 31);
            const uint32_t code1 = ((v1.x >> b1) & 1u) | (((v1.y >> b1) & 1u) << 1) | (((v1.z >> b1) & 1u) << 2) | (((v1.w >> b1) & 1u) << 3);
            const uint32_t code2 = ((v2.x >> b2s) & 1u) | (((v2.y >> b2s) & 1u) << 1) | (((v2.z >> b2s) & 1u) << 2) | (((v2.w >> b2s) & 1u) << 3);
            k.qa = has_quals ? (int)q1t[k.row + (uint32_t)k.i] : 0;
            k.qb = has_quals ? (int)q2t[k.row + (uint32_t)k.jx] : 0;
            k.base1 = code1 ? S.letter[code1] : s1t[k.row + (uint32_t)k.i];
            k.raw2 = code2 ? S.letter[code2] : s2t[k.row + (uint32_t)k.jx];
        }
        return k;
    };
    const auto apply = [&](const Task &k) {
        if (!k.valid) return;
        uint32_t delta = 0u;
        const int e = correct_apply_delta(s1t, q1t, s2t, q2t, k.row + (uint32_t)k.i, k.row + (uint32_t)k.jx, k.base1, k.raw2, k.qa, k.qb,
                                          A.action, A.min_qual_diff, comp, delta);
        if (e) S.err[k.src] = e;
        else if (delta) {
            lds_add(&S.cnt[k.src], delta);
            if (delta >= CORRECT_NP) {                              // 'liberal', qualities too close: decided by the pair's mean qualities below
                const uint32_t slot = lds_add(S.ptail, 1u);
                if (slot < pend_cap) queue[S.qcap - 1 - (int)slot] = (uint16_t)k.entry;
            }
        }
    };
    const auto drain = [&](int from, int count) {               // tasks queue[from .. from + count), count <= 64, wave-uniform
        const bool v = lane < count;
        const Task k = fetch(v, v ? (uint32_t)queue[from + lane] : 0u);
        apply(k);
    };
    int total = 0;
#pragma unroll
    for (int w = 0; w < PW; ++w) total += __builtin_popcount(mism[w]);
    uint32_t at = total ? lds_add(S.tail, (uint32_t)total) : 0u;
    wave_sync_lds();
    const int qall = __builtin_amdgcn_readfirstlane((int)*S.tail);
    if (qall <= CORRECT_QUEUE_ENTRIES) {
        // (the common case) every lane lists all its positions at once -- one reservation instead of one per plane word
        // with three wave barriers each -- and the wave works the list off CORRECT_TASKS_PER_LANE tasks per lane and round trip
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            uint32_t m = mism[w];
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1u;
                queue[at++] = (uint16_t)((lane << 9) | (32 * w + b));
            }
        }
        wave_sync_lds();
        for (int base = 0; base < qall; base += 64 * CORRECT_TASKS_PER_LANE) {
            Task k[CORRECT_TASKS_PER_LANE];
#pragma unroll
            for (int u = 0; u < CORRECT_TASKS_PER_LANE; ++u) {
                const int idx = base + 64 * u + lane;
                const bool v = idx < qall;
                k[u] = fetch(v, v ? (uint32_t)queue[idx] : 0u);
            }
#pragma unroll
            for (int u = 0; u < CORRECT_TASKS_PER_LANE; ++u) apply(k[u]);
        }
    } else {
        // more positions than the queue holds (a wave of unrelated reads "matched" at a high error rate): word by word,
        // full rounds worked off as they fill
        wave_sync_lds();
        if (lane == 0) *S.tail = 0u;
        wave_sync_lds();
        int qsize = 0;                                            // wave-uniform
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            uint32_t m = mism[w];                                 // (0 beyond nchunks)
            const int cnt = __builtin_popcount(m);
            if (cnt) {
                uint32_t at2 = lds_add(S.tail, (uint32_t)cnt);
                while (m) {
                    const int b = __builtin_ctz(m);
                    m &= m - 1u;
                    queue[at2++] = (uint16_t)((lane << 9) | (32 * w + b));
                }
            }
            wave_sync_lds();
            qsize = __builtin_amdgcn_readfirstlane((int)*S.tail);
            while (qsize >= 64) {                                 // full rounds, off the end of the queue
                qsize -= 64;
                drain(qsize, 64);
            }
            wave_sync_lds();
            if (lane == 0) *S.tail = (uint32_t)qsize;
            wave_sync_lds();
        }
        if (qsize > 0) drain(0, qsize);
    }
    // the lanes' stores, before their pairs' owners (lanes of this same wave: one CU, one L1) read them back --
    // a device-scope fence would write back the L2 for the sake of other XCDs, once per wave
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    wave_sync_lds();
    // ---- 'liberal' ties (modifiers.py:301-322): the read with the better MEAN quality over the overlap wins every position
    // whose two qualities were too close.  A lane summing its own pair's two quality rows and then walking its positions
    // one memory round trip at a time cost a quarter of the fused kernel; here the WAVE sums a tied pair's rows (64 lanes
    // x 4 bytes per load, LDS atomics), and the tied positions -- listed by the first pass -- are decided 64 at a time.
    uint32_t packed_counts = S.cnt[lane];
    int err = S.err[lane];
    const uint32_t npend_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)*S.ptail);
    const bool tied = live && todo && !err && (packed_counts >> 20) > 0u;
    if (npend_all > 0u && npend_all <= pend_cap && has_quals) {
        // The tied pairs of the wave, listed (the main queue is worked off: its first 64 entries take the list), and their
        // quality rows cut into 16-byte pieces: work item = (pair, piece), 32 piece slots per pair (rows of up to 512
        // bytes), 64 items per round, CORRECT_PIECES_PER_LANE rounds' loads in flight at once -- every tied pair's rows in one or two
        // memory round trips, where a round per four pairs took four.
        const unsigned long long tm = __ballot(tied);
        const int ntied = (int)__popcll(tm);
        if (tied) queue[lane_rank_below(tm, lane)] = (uint16_t)lane;
        wave_sync_lds();
        const int nitems = 32 * ntied;
        const auto piece_sums = [&](int item) -> unsigned long long {
            if (item >= nitems) return 0ull;
            const int src = (int)queue[item >> 5], piece = item & 31;
            const int js = (int)S.jv[src], off = 16 * piece;
            if (off >= js) return 0ull;
            const uint32_t row = (uint32_t)src * pitch;
            const int nvalid = js - off;                          // bytes of this piece inside the overlap
            uint32_t w1[4], w2[4];
            if (nvalid >= 16 || js >= 16) {
                // a full piece, or the row's last piece read as the 16 bytes that END at the overlap's end (never a byte
                // behind it): its leading 16 - nvalid bytes belong to the piece before and are masked away
                const int at = nvalid >= 16 ? off : js - 16;
                const atr_u128_unaligned v1 = *(const atr_u128_unaligned *)&q1t[row + (uint32_t)at];
                const atr_u128_unaligned v2 = *(const atr_u128_unaligned *)&q2t[row + (uint32_t)at];
                w1[0] = v1.x; w1[1] = v1.y; w1[2] = v1.z; w1[3] = v1.w;
                w2[0] = v2.x; w2[1] = v2.y; w2[2] = v2.z; w2[3] = v2.w;
                if (nvalid < 16) {
                    const int lo = 16 - nvalid;                   // first byte that counts
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t m = 4 * d + 4 <= lo ? 0u : 4 * d >= lo ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (lo - 4 * d));
                        w1[d] &= m; w2[d] &= m;
                    }
                }
            } else {                                              // an overlap of fewer than 16 bases: byte by byte
                w1[0] = w1[1] = w1[2] = w1[3] = w2[0] = w2[1] = w2[2] = w2[3] = 0u;
                for (int b = 0; b < nvalid; ++b) {
                    w1[0] += (uint32_t)q1t[row + (uint32_t)b];    // (w1[0] / w2[0] are the sums themselves here)
                    w2[0] += (uint32_t)q2t[row + (uint32_t)b];
                }
                return (unsigned long long)w1[0] | ((unsigned long long)w2[0] << 32);
            }
            uint32_t sum1 = 0u, sum2 = 0u;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                sum1 = __builtin_amdgcn_sad_u8(w1[d], 0u, sum1);
                sum2 = __builtin_amdgcn_sad_u8(w2[d], 0u, sum2);
            }
            return (unsigned long long)sum1 | ((unsigned long long)sum2 << 32);
        };
        for (int base = 0; base < nitems; base += 64 * CORRECT_PIECES_PER_LANE) {          // wave-uniform
            unsigned long long part[CORRECT_PIECES_PER_LANE];
#pragma unroll
            for (int u = 0; u < CORRECT_PIECES_PER_LANE; ++u) part[u] = piece_sums(base + 64 * u + lane);
#pragma unroll
            for (int u = 0; u < CORRECT_PIECES_PER_LANE; ++u) {
                const int item = base + 64 * u + lane;
                if (part[u]) lds_add(&S.acc[(int)queue[item >> 5]], part[u]);
            }
        }
        wave_sync_lds();
        uint32_t dir = 0u;                                        // 1: read 1 is better (read 2 is corrected), 2: read 2 is better
        if (tied) {
            if (j <= 0) {
                err = -3;
                S.err[lane] = -3;
            } else {
                const unsigned long long a = S.acc[lane];
                const long long sum1 = (long long)(uint32_t)a, sum2 = (long long)(a >> 32);
                const double diff = (double)sum1 / (double)j - (double)sum2 / (double)j;
                dir = diff > 1.0 ? 1u : diff < -1.0 ? 2u : 0u;
            }
        }
        wave_sync_lds();
        S.acc[lane] = (unsigned long long)dir;
        wave_sync_lds();
        for (uint32_t base = 0u; base < npend_all; base += 64u) {
            if (base + (uint32_t)lane < npend_all) {
                const uint32_t t = queue[S.qcap - 1 - (int)(base + (uint32_t)lane)];
                const int src = (int)(t >> 9), i = (int)(t & 511u);
                const uint32_t d = (uint32_t)S.acc[src];
                if (d != 0u) {
                    const int jx = (int)S.jv[src] - 1 - i;
                    const uint32_t row = (uint32_t)src * pitch, o1 = row + (uint32_t)i, o2 = row + (uint32_t)jx;
                    const uint4 v1 = p1t[(uint32_t)((i >> 5) * 64 + src)];
                    const uint4 v2 = p2t[(uint32_t)((jx >> 5) * 64 + src)];
                    const uint32_t b1 = (uint32_t)(i & 31), b2s = (uint32_t)(jx & 31);
                    const uint32_t code1 = ((v1.x >> b1) & 1u) | (((v1.y >> b1) & 1u) << 1) | (((v1.z >> b1) & 1u) << 2) | (((v1.w >> b1) & 1u) << 3);
                    const uint32_t code2 = ((v2.x >> b2s) & 1u) | (((v2.y >> b2s) & 1u) << 1) | (((v2.z >> b2s) & 1u) << 2) | (((v2.w >> b2s) & 1u) << 3);
                    const int qa = (int)q1t[o1], qb = (int)q2t[o2];
                    const uint8_t base1 = code1 ? S.letter[code1] : s1t[o1], raw2 = code2 ? S.letter[code2] : s2t[o2];
                    const uint8_t base2 = comp[raw2];
                    const int qd = qa - qb;
                    if (!(base1 == base2 || base1 == 'N' || base2 == 'N' || qd >= A.min_qual_diff || qd <= -A.min_qual_diff)) {
                        if (d == 1u) {
                            const uint8_t cb = comp[base1];
                            if (cb == 0) S.err[src] = -1;
                            else { s2t[o2] = cb; q2t[o2] = (uint8_t)qa; lds_add(&S.cnt[src], CORRECT_C2); }
                        } else {
                            s1t[o1] = base2; q1t[o1] = (uint8_t)qb; lds_add(&S.cnt[src], CORRECT_C1);
                        }
                    }
                }
            }
        }
        wave_sync_lds();
        packed_counts = S.cnt[lane];
        err = S.err[lane];
    }
    if (!live) return;
    if (!todo) {
        A.changed[2 * p] = A.changed[2 * p + 1] = 0;
        A.newlen[2 * p] = len1; A.newlen[2 * p + 1] = len2;
        return;
    }
    int c1 = (int)(packed_counts & 1023u), c2 = (int)((packed_counts >> 10) & 1023u);
    if (npend_all > pend_cap && !err && (packed_counts >> 20) > 0u) {   // a tie list that did not fit (thousands of ties in one wave): lane by lane
        uint8_t *r1 = A.s1 + p * A.stride, *r2 = A.s2 + p * A.stride;
        correct_ties(r1, has_quals ? A.q1 + p * A.stride : nullptr, r2, has_quals ? A.q2 + p * A.stride : nullptr, j, mism, nchunks,
                     A.min_qual_diff, comp, c1, c2, err);
    }
    A.changed[2 * p] = err ? err : c1;
    A.changed[2 * p + 1] = err ? 0 : c2;
    A.newlen[2 * p] = (c1 > 0 && !err) ? min(len1, len2) : len1;   // the truncation quirk of correct_errors_one
    A.newlen[2 * p + 1] = len2;
}

}  // namespace atr
#endif
