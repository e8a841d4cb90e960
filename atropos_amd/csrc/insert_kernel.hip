// insert_kernel.hip -- gfx950 kernel around insert_core.hpp: one read pair per lane,
// 64 pairs per wavefront.  Each lane loads both plane64-packed reads completely (2 * 4W
// dwords, coalesced 1 KiB bursts per chunk), then runs the bit-sliced overlap sweep; the
// only other global traffic is the occasional double-precision table lookup for a hit
// (L2 resident, 2 x 528 KB) and three coalesced 16-byte result stores.
#include <hip/hip_runtime.h>
#include "insert_host.hpp"
#include "correct_wave.hpp"

namespace atr {

static __device__ __forceinline__ int wave_max_i32_ins(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// what the fused kernels (match + error correction of the overlap in one pass) get on top
struct InsertFuse {
    CorrectArgs A;
    CompTable ct;
};

template <int NCH, class IP, bool FUSE = false>
__device__ __forceinline__ void insert_body(const IP &ip, const uint4 *__restrict__ packed1,
                                            const int32_t *__restrict__ lens1, const uint4 *__restrict__ packed2,
                                            const int32_t *__restrict__ lens2, long long npairs, int max_len,
                                            uint4 *__restrict__ out, const InsertFuse *fz = nullptr) {
    constexpr int W = NCH;
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long r = tile * 64 + lane;
    const bool live = r < npairs;                    // (a wave behind the last tile runs along on empty pairs: the block meets at a barrier)
    const int len1 = live ? (lens1 ? lens1[r] : max_len) : 0;
    const int len2 = live ? (lens2 ? lens2[r] : max_len) : 0;

    uint32_t b1[4 * W], b2[4 * W];
    const long long ntiles = (npairs + 63) >> 6;
    const long long tile_ld = min(tile, ntiles - 1);                  // (an idle wave reads the last tile)
    const uint4 *t1 = packed1 + (size_t)tile_ld * NCH * 64 + lane;
    const uint4 *t2 = packed2 + (size_t)tile_ld * NCH * 64 + lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint4 a = t1[(size_t)c * 64], b = t2[(size_t)c * 64];
        b1[4 * c] = a.x; b1[4 * c + 1] = a.y; b1[4 * c + 2] = a.z; b1[4 * c + 3] = a.w;
        b2[4 * c] = b.x; b2[4 * c + 1] = b.y; b2[4 * c + 2] = b.z; b2[4 * c + 3] = b.w;
    }
    PairState<W> P;
    pair_init<W>(P, ip, len1, len2, b1, b2);
    const int jmax = wave_max_i32_ins(P.L);
    // read 2's planes per lane in LDS ([plane][word][lane]: a lane's words sit in its own bank), its list of
    // overlap lengths to cost exactly, and the per-length hit thresholds (pass 2 indexes them per lane)
    // (the fused kernels' task queue takes the planes' place afterwards: four-chunk reads keep a fifth word's room for it)
    constexpr int RL_WORDS = (FUSE && W < 5) ? 4 * (W + 1) * 64 : 4 * W * 64;
    __shared__ uint32_t s_rl[4][RL_WORDS];
    __shared__ __attribute__((aligned(16))) uint16_t s_cl[4][(FUSE && ins_list_cap(W) < 12 ? 12 : ins_list_cap(W)) * 64];   // (the fused kernels' counters take 1 536 B of it)
    __shared__ int32_t s_thr_hit[INS_MAX_LEN + 1];
    for (int i = threadIdx.x; i <= INS_MAX_LEN; i += 256) s_thr_hit[i] = ip.thr_hit[i];
    __shared__ uint8_t s_comp[FUSE ? 256 : 1], s_letter[16];
    if constexpr (FUSE) {
        s_comp[threadIdx.x] = fz->ct.c[threadIdx.x];
        correct_letter_table(s_letter);
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *rl = &s_rl[wv][lane];
    uint16_t *cl = &s_cl[wv][lane];
    planes_to_lds<W>(P, rl, 64);
    __syncthreads();
    sweep_probed<W>(P, ip, jmax, (const uint32_t *)t1, (const uint32_t *)t2, 64 * 4, rl, 64, cl, 64, s_thr_hit,
                    [](int n) { return wave_max_i32_ins(n); });
    // "stop after 100 hits" depends on the order of the hits: redo such (low-complexity) pairs
    // in the reference's order; the other lanes of the wave get the same result again
    if (__any(!unordered_is_exact<W>(P))) {
        planes_from_lds<W>(P, rl, 64);
        sweep_ordered<W>(P, ip, jmax);
    }
    if (live) {
        uint32_t rec[12];
        pair_result<W>(P, ip, rec);
        uint4 *o = out + 3 * r;
        o[0] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        o[1] = make_uint4(rec[4], rec[5], rec[6], rec[7]);
        o[2] = make_uint4(rec[8], rec[9], rec[10], rec[11]);
    }
    if constexpr (FUSE) {
        // The error correction of the overlap (modifiers.py:397-404, :219-350) while the pair is still here: the
        // disagreeing positions are the very words the match counted (read 1 in registers, read 2's reversed
        // complement in LDS); a second kernel streams both reads' planes again to find them (correct_planes_kernel).
        static_assert(sizeof(s_rl[0]) >= CORRECT_QUEUE_ENTRIES * sizeof(uint16_t), "task queue does not fit read 2's planes");
        static_assert(sizeof(s_cl[0]) >= 1024 + 64 * 8, "the per-pair counters do not fit the candidate lists");
        const bool todo = live && P.has_best && P.best_cost > 0;
        const int j = todo ? P.best_j : 0;
        uint32_t mism[W];
        insert_overlap_mismatches<W>(P, rl, 64, j, mism);
        wave_sync_lds();                                    // every lane has read its planes: the queue takes their place
        CorrectWaveLds S;
        S.queue = lds_view((uint16_t *)&s_rl[wv][0]);
        uint8_t *small = (uint8_t *)&s_cl[wv][0];
        S.cnt = lds_view((uint32_t *)small); S.err = lds_view((int32_t *)(small + 256)); S.jv = lds_view((int16_t *)(small + 512));
        S.tail = lds_view((uint32_t *)(small + 640)); S.ptail = lds_view((uint32_t *)(small + 644));
        S.acc = lds_view((unsigned long long *)(small + 1024)); S.qcap = (int)(sizeof(s_rl[0]) / (sizeof(uint16_t)));
        S.comp = lds_view((const uint8_t *)s_comp); S.letter = lds_view((const uint8_t *)s_letter);
        correct_wave_tail<W>(S, fz->A, tile, lane, live, todo, j, len1, len2, mism);
    }
}

// Two entry points around the same body: reads of up to five chunks (160 bases) and longer ones.  Round 6: read 2's
// planes in LDS lost their padding word and eight-chunk reads list 12 overlap lengths instead of 16 -- 40.5 KB per block
// instead of 46.6, a FOURTH block per CU for 250-base pairs (128 VGPRs; the 22-35 spilled dwords are all in the ordered
// redo of low-complexity pairs, behind a wave-uniform branch); the kernels are latency bound (one / two / three / four
// blocks per CU: 0.83 / 0.45 / 0.34 / 0.30 ms per 2 M pairs for the match alone), C5 7.1 -> 8.0-8.5 G reads/s.  Reads of
// up to five chunks: without the padding word and with eight listed lengths for five-chunk reads a block takes 25.9 KB --
// SIX blocks per CU, asked for here with waves_per_eu(6) (80 VGPRs, 52 spilled dwords at five chunks): C3 24.0 -> 26.0
// (five blocks, 96 VGPRs) -> 27.2-27.6 G reads/s.  The fused kernels' counters keep their 1.5 KB: five blocks.
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void insert_kernel_dense(
    const InsertParams ip, const uint4 *__restrict__ packed1, const int32_t *__restrict__ lens1,
    const uint4 *__restrict__ packed2, const int32_t *__restrict__ lens2, long long npairs, int max_len,
    uint4 *__restrict__ out) {
    insert_body<NCH>(ip, packed1, lens1, packed2, lens2, npairs, max_len, out);
}
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void insert_kernel(const InsertParams ip, const uint4 *__restrict__ packed1,
                                                     const int32_t *__restrict__ lens1,
                                                     const uint4 *__restrict__ packed2,
                                                     const int32_t *__restrict__ lens2, long long npairs,
                                                     int max_len, uint4 *__restrict__ out) {
    insert_body<NCH>(ip, packed1, lens1, packed2, lens2, npairs, max_len, out);
}

// match + correction in one pass (atr_insert_match_correct_batch): the same two register budgets
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void insert_correct_kernel_dense(
    const InsertParams ip, const uint4 *__restrict__ packed1, const int32_t *__restrict__ lens1,
    const uint4 *__restrict__ packed2, const int32_t *__restrict__ lens2, long long npairs, int max_len,
    uint4 *__restrict__ out, const InsertFuse fz) {
    insert_body<NCH, InsertParams, true>(ip, packed1, lens1, packed2, lens2, npairs, max_len, out, &fz);
}
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void insert_correct_kernel(
    const InsertParams ip, const uint4 *__restrict__ packed1, const int32_t *__restrict__ lens1,
    const uint4 *__restrict__ packed2, const int32_t *__restrict__ lens2, long long npairs, int max_len,
    uint4 *__restrict__ out, const InsertFuse fz) {
    insert_body<NCH, InsertParams, true>(ip, packed1, lens1, packed2, lens2, npairs, max_len, out, &fz);
}

// MiSeq-length reads (nine and ten chunks, up to 320 bases): 80 plane dwords per pair live in registers,
// two waves per SIMD.  The same register budget serves every read length when an ADAPTER has more than 64
// bases (LONGAD: the overhangs are compared in two halves, insert_core.hpp InsertParamsLong).
template <int NCH, bool LONGAD, bool CASED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void insert_kernel_long(
    const InsertParams ip, const uint4 *__restrict__ packed1, const int32_t *__restrict__ lens1,
    const uint4 *__restrict__ packed2, const int32_t *__restrict__ lens2, long long npairs, int max_len,
    uint4 *__restrict__ out) {
    if (LONGAD && CASED) insert_body<NCH>(static_cast<const InsertParamsLongCased &>(ip), packed1, lens1, packed2, lens2, npairs, max_len, out);
    else if (CASED) insert_body<NCH>(static_cast<const InsertParamsCased &>(ip), packed1, lens1, packed2, lens2, npairs, max_len, out);
    else if (LONGAD) insert_body<NCH>(static_cast<const InsertParamsLong &>(ip), packed1, lens1, packed2, lens2, npairs, max_len, out);
    else insert_body<NCH>(ip, packed1, lens1, packed2, lens2, npairs, max_len, out);
}

template <int NCH>
static int launch_nch(const atr_insert_aligner *a, const uint4 *p1, const int32_t *l1, const uint4 *p2,
                      const int32_t *l2, long long npairs, int max_len, uint4 *out, int cased, hipStream_t st) {
    const long long ntiles = (npairs + 63) / 64;
    if (cased) {                                          // soft-masked reads: kernels of their own, as for long adapters
        if (a->p.long_adapters)
            hipLaunchKernelGGL((insert_kernel_long<NCH, true, true>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p,
                               p1, l1, p2, l2, npairs, max_len, out);
        else
            hipLaunchKernelGGL((insert_kernel_long<NCH, false, true>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p,
                               p1, l1, p2, l2, npairs, max_len, out);
        return (int)hipGetLastError();
    }
    if (a->p.long_adapters) {
        hipLaunchKernelGGL((insert_kernel_long<NCH, true>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p, p1, l1,
                           p2, l2, npairs, max_len, out);
        return (int)hipGetLastError();
    }
    if constexpr (NCH == 4 || NCH == 5)
        hipLaunchKernelGGL((insert_kernel_dense<NCH>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p, p1, l1,
                           p2, l2, npairs, max_len, out);
    else if constexpr (NCH > 8)
        hipLaunchKernelGGL((insert_kernel_long<NCH, false>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p, p1, l1,
                           p2, l2, npairs, max_len, out);
    else
        hipLaunchKernelGGL((insert_kernel<NCH>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a->p, p1, l1,
                           p2, l2, npairs, max_len, out);
    return (int)hipGetLastError();
}

int launch_insert(const atr_insert_aligner *a, const uint4 *p1, const int32_t *l1, const uint4 *p2,
                  const int32_t *l2, long long npairs, int nchunks, int max_len, uint4 *out, int cased, hipStream_t st) {
    switch (nchunks) {
        case 0: case 1: return launch_nch<1>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 2: return launch_nch<2>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 3: return launch_nch<3>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 4: return launch_nch<4>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 5: return launch_nch<5>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 6: return launch_nch<6>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 7: return launch_nch<7>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 8: return launch_nch<8>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        case 9: return launch_nch<9>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
        default: return launch_nch<10>(a, p1, l1, p2, l2, npairs, max_len, out, cased, st);
    }
}

// The fused form: chunk counts 4 .. 8 with the plain parameter block; returns -1 when the batch is not one of those
// (the caller then runs the two kernels one after the other).
int launch_insert_correct(const atr_insert_aligner *a, const uint4 *p1, const int32_t *l1, const uint4 *p2, const int32_t *l2,
                          long long npairs, int nchunks, int max_len, uint4 *out, uint8_t *s1, uint8_t *q1, uint8_t *s2,
                          uint8_t *q2, long long stride, int action, int min_qual_diff, const uint8_t *comp,
                          int32_t *changed, int32_t *newlen, hipStream_t st) {
    if (a->p.long_adapters || nchunks < 4 || nchunks > 8) return -1;
    InsertFuse fz;
    fz.A.planes1 = p1; fz.A.planes2 = p2; fz.A.nchunks = nchunks;
    fz.A.s1 = s1; fz.A.q1 = q1; fz.A.s2 = s2; fz.A.q2 = q2; fz.A.stride = stride;
    fz.A.action = action; fz.A.min_qual_diff = min_qual_diff; fz.A.changed = changed; fz.A.newlen = newlen;
    memcpy(fz.ct.c, comp, 256);
    const long long ntiles = (npairs + 63) / 64;
    const dim3 grid((unsigned)((ntiles + 3) / 4));
#define ATR_FUSED(N) case N: hipLaunchKernelGGL((insert_correct_kernel<N>), grid, dim3(256), 0, st, a->p, p1, l1, p2, l2, npairs, max_len, out, fz); break
#define ATR_FUSED_DENSE(N) case N: hipLaunchKernelGGL((insert_correct_kernel_dense<N>), grid, dim3(256), 0, st, a->p, p1, l1, p2, l2, npairs, max_len, out, fz); break
    switch (nchunks) {
        ATR_FUSED_DENSE(4); ATR_FUSED_DENSE(5); ATR_FUSED(6); ATR_FUSED(7); ATR_FUSED(8);
    }
#undef ATR_FUSED_DENSE
#undef ATR_FUSED
    return (int)hipGetLastError();
}

}  // namespace atr
