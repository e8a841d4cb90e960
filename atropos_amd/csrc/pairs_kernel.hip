// pairs_kernel.hip -- atr_locate_pairs_batch: Aligner.locate with a per-pair reference
// (pairs_core.hpp).  One wave = 64 pairs; the wave's DP columns and staged references live in
// dynamic LDS ((max_m + 1) + ceil(max_m / 8) dwords per lane), so occupancy is LDS-bound:
// 3 waves per CU at 150 bp, 2 at 250 bp, 1 at 320 bp (92 KB of the CU's 160).  VALU-bound like every DP here (no MFMA: the
// recurrence is a min-plus chain along the column, not a contraction).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

#include "atropos_hip.h"
#include "pairs_core.hpp"
#include "pairs_fast_core.hpp"
#include "wave_core.hpp"

namespace atr {

int hip_fail(hipError_t e, const char *what);             // api.hip

// The pairs the fast pipeline (pairs_fast.hip) hands to the full sweep arrive as a slot range of its task list:
// order[range[0] .. range[1]) (both on the device; .x of a task = the pair).  order == nullptr: all pairs in turn.
struct PairIndex {
    const uint4 *order;
    const uint32_t *range;
};
__device__ __forceinline__ bool pair_of_slot(const PairIndex ix, long long slot, long long npairs, long long &r) {
    if (!ix.order) { r = slot; return slot < npairs; }
    const long long at = slot + (long long)ix.range[0];
    if (at >= (long long)ix.range[1]) { r = 0; return false; }
    r = (long long)ix.order[at].x;
    return true;
}

template <bool AND_MODE, bool XREP>
__global__ __launch_bounds__(64) void pairs_kernel(const PairParams p, const uint32_t *__restrict__ ref_packed,
                                                   const int32_t *__restrict__ ref_lens, int ref_chunks,
                                                   int ref_max_len, int revcomp,
                                                   const uint32_t *__restrict__ qry_packed,
                                                   const int32_t *__restrict__ qry_lens, int qry_chunks, int qry_max_len,
                                                   long long npairs, uint4 *__restrict__ out, const PairIndex ix) {
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pairs[];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 3; i += 64) s_thr[i] = p.thr[i];
    __syncthreads();
    const int lane = threadIdx.x;
    long long r;
    if (!pair_of_slot(ix, (long long)blockIdx.x * 64 + lane, npairs, r)) return;
    uint32_t *col = s_pairs + lane;                                   // (max_m + 1) x 64
    uint32_t *refw = s_pairs + (size_t)(ref_max_len + 1) * 64 + lane; // ceil(max_m / 8) x 64
    const int m = min(ref_lens ? ref_lens[r] : ref_max_len, ref_max_len);
    const int n = min(qry_lens ? qry_lens[r] : qry_max_len, qry_max_len);
    const uint32_t *rp = ref_packed + ((size_t)(r >> 6) * ref_chunks * 64 + (r & 63)) * 4;
    const uint32_t *qp = qry_packed + ((size_t)(r >> 6) * qry_chunks * 64 + (r & 63)) * 4;
    stage_reference(refw, 64, rp, m, revcomp != 0);
    uint32_t rec[4];
    locate_pair_one<AND_MODE, XREP>(col, 64, refw, 64, m, qp, n, p, s_thr, rec);
    out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

// Register-column variant (pairs_core.hpp): references of up to PAIRS_REG_MAX rows.
template <int MT, bool AND_MODE, bool XREP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void pairs_reg_kernel(const PairParams p, const uint32_t *__restrict__ ref_packed,
                                                        const int32_t *__restrict__ ref_lens, int ref_chunks,
                                                        int ref_max_len, int revcomp,
                                                        const uint32_t *__restrict__ qry_packed,
                                                        const int32_t *__restrict__ qry_lens, int qry_chunks,
                                                        int qry_max_len, long long npairs, uint4 *__restrict__ out,
                                                        const PairIndex ix) {
    constexpr int NW = (MT + 31) / 32;
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pairs[];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 3; i += 256) s_thr[i] = p.thr[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    long long r;
    const bool live = pair_of_slot(ix, tile * 64 + lane, npairs, r);
    if (__ballot(live) == 0ull) return;
    uint32_t *tab = s_pairs + (size_t)wave * 16 * NW * 64 + lane;     // [code][word][lane]
    const int m = live ? min(ref_lens ? ref_lens[r] : ref_max_len, ref_max_len) : 0;
    const int n = live ? min(qry_lens ? qry_lens[r] : qry_max_len, qry_max_len) : 0;
    const uint32_t *rp = ref_packed + ((size_t)(r >> 6) * ref_chunks * 64 + (r & 63)) * 4;
    const uint32_t *qp = qry_packed + ((size_t)(r >> 6) * qry_chunks * 64 + (r & 63)) * 4;
    build_match_masks<MT, AND_MODE>(tab, 64, rp, m, revcomp != 0);
    int mlo = live ? m : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mlo = min(mlo, __shfl_xor(mlo, o, 64));
    mlo = __builtin_amdgcn_readfirstlane(mlo);
    int mhi = m;                                        // (0 for the lanes past the end)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mhi = max(mhi, __shfl_xor(mhi, o, 64));
    mhi = __builtin_amdgcn_readfirstlane(mhi);
    uint32_t rec[4];
    locate_pair_reg<MT, AND_MODE, XREP>(tab, 64, m, mlo, mhi, qp, n, p, s_thr, rec);
    if (live) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

// Register strips (pairs_core.hpp, locate_pair_strips): references of more than PAIRS_REG_MAX rows.  A persistent
// grid; every wave owns one slot of the boundary buffer ((query_max_len + 1) x 64 words: the bottom row a strip
// leaves to the next one).
template <bool AND_MODE, bool XREP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void pairs_strip_kernel(
    const PairParams p, const uint32_t *__restrict__ ref_packed, const int32_t *__restrict__ ref_lens, int ref_chunks,
    int ref_max_len, int revcomp, const uint32_t *__restrict__ qry_packed, const int32_t *__restrict__ qry_lens,
    int qry_chunks, int qry_max_len, long long npairs, uint32_t *__restrict__ boundary, uint4 *__restrict__ out,
    const PairIndex ix) {
    constexpr int NW = PAIRS_STRIP_ROWS / 32;
    __shared__ int16_t s_thr[PAIRS_MAX_LEN + 3];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pairs[];
    for (int i = threadIdx.x; i < PAIRS_MAX_LEN + 3; i += 256) s_thr[i] = p.thr[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *tab = s_pairs + (size_t)wave * 16 * NW * 64 + lane;     // [code][word][lane]
    uint32_t *bnd = boundary + ((size_t)blockIdx.x * 4 + wave) * (size_t)(qry_max_len + 1) * 64 + lane;
    const long long nslots = ix.order ? (long long)ix.range[1] - (long long)ix.range[0] : npairs;
    const long long ntiles = (nslots + 63) >> 6;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        long long r;
        const bool live = pair_of_slot(ix, tile * 64 + lane, npairs, r);
        const int m = live ? min(ref_lens ? ref_lens[r] : ref_max_len, ref_max_len) : 0;
        const int n = live ? min(qry_lens ? qry_lens[r] : qry_max_len, qry_max_len) : 0;
        const uint32_t *rp = ref_packed + ((size_t)(r >> 6) * ref_chunks * 64 + (r & 63)) * 4;
        const uint32_t *qp = qry_packed + ((size_t)(r >> 6) * qry_chunks * 64 + (r & 63)) * 4;
        int mtop = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mtop = max(mtop, __shfl_xor(mtop, o, 64));
        mtop = __builtin_amdgcn_readfirstlane(mtop);
        int mlo_s[PAIRS_MAX_STRIPS], mhi_s[PAIRS_MAX_STRIPS];
#pragma unroll
        for (int s = 0; s < PAIRS_MAX_STRIPS; ++s) {
            const int loc = m - s * PAIRS_STRIP_ROWS;
            const bool here = live && loc >= 1 && loc <= PAIRS_STRIP_ROWS;
            int lo = here ? loc : 0x7fffffff, hi = here ? loc : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
            mlo_s[s] = __builtin_amdgcn_readfirstlane(lo);
            mhi_s[s] = __builtin_amdgcn_readfirstlane(hi);
        }
        uint32_t rec[4];
        locate_pair_strips<AND_MODE, XREP>(tab, 64, bnd, 64, rp, revcomp != 0, m, mlo_s, mhi_s, mtop, qp, n, p, s_thr, rec);
        if (live) out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

static hipError_t launch_pairs_strips(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                                      const uint32_t *qp, const int32_t *ql, int qmax, long long npairs, uint4 *out,
                                      hipStream_t st, const PairIndex ix = PairIndex{nullptr, nullptr}) {
    constexpr int NW = PAIRS_STRIP_ROWS / 32;
    const size_t lds = (size_t)4 * 16 * NW * 64 * 4;
    const long long ntiles = (npairs + 63) / 64;
    const unsigned blocks = (unsigned)std::min<long long>((ntiles + 3) / 4, 512);      // two blocks per CU are resident
    const size_t bytes = (size_t)blocks * 4 * (size_t)(qmax + 1) * 64 * 4;
    uint32_t *boundary = nullptr;
    hipError_t e = hipMallocAsync((void **)&boundary, bytes, st);                       // stream-ordered: one buffer per call
    if (e != hipSuccess) return e;
    const bool xrep = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;
#define ATR_LAUNCH_STRIPS(AND, XR)                                                                                     \
    do {                                                                                                               \
        e = hipFuncSetAttribute((const void *)pairs_strip_kernel<AND, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess)                                                                                           \
            hipLaunchKernelGGL((pairs_strip_kernel<AND, XR>), dim3(blocks), dim3(256), lds, st, p, rp, rl, (rmax + 31) / 32, rmax, \
                               revcomp, qp, ql, (qmax + 31) / 32, qmax, npairs, boundary, out, ix);                     \
    } while (0)
    if (p.and_mode) { if (xrep) ATR_LAUNCH_STRIPS(true, true); else ATR_LAUNCH_STRIPS(true, false); }
    else { if (xrep) ATR_LAUNCH_STRIPS(false, true); else ATR_LAUNCH_STRIPS(false, false); }
#undef ATR_LAUNCH_STRIPS
    const hipError_t launched = e == hipSuccess ? hipGetLastError() : e;
    const hipError_t freed = hipFreeAsync(boundary, st);
    return launched != hipSuccess ? launched : freed;
}

template <int MT>
static hipError_t launch_pairs_reg(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                                   const uint32_t *qp, const int32_t *ql, int qmax, long long npairs, uint4 *out,
                                   hipStream_t st, const PairIndex ix = PairIndex{nullptr, nullptr}) {
    constexpr int NW = (MT + 31) / 32;
    const size_t lds = (size_t)4 * 16 * NW * 64 * 4;
    const dim3 grid((unsigned)((npairs + 255) / 256)), block(256);
    hipError_t e = hipSuccess;
    const bool xrep = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;           // mismatch-counting payload (pairs_core.hpp)
#define ATR_LAUNCH_PAIRS(AND, XR)                                                                                      \
    do {                                                                                                               \
        e = hipFuncSetAttribute((const void *)pairs_reg_kernel<MT, AND, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((pairs_reg_kernel<MT, AND, XR>), grid, block, lds, st, p, rp, rl, (rmax + 31) / 32, rmax,   \
                           revcomp, qp, ql, (qmax + 31) / 32, qmax, npairs, out, ix);                                  \
    } while (0)
    if (p.and_mode) { if (xrep) ATR_LAUNCH_PAIRS(true, true); else ATR_LAUNCH_PAIRS(true, false); }
    else { if (xrep) ATR_LAUNCH_PAIRS(false, true); else ATR_LAUNCH_PAIRS(false, false); }
#undef ATR_LAUNCH_PAIRS
    return hipGetLastError();
}

// The full sweep: register-column kernel, register strips, or (without stream-ordered allocation) the LDS column.
static int pairs_full(const PairParams &p, const uint32_t *rp, const int32_t *rl, int ref_max_len, int revcomp_ref,
                      const uint32_t *qp, const int32_t *ql, int query_max_len, long long npairs, uint4 *out, hipStream_t st,
                      const PairIndex ix) {
    hipError_t e;
    if (ref_max_len <= PAIRS_REG_MAX) {                       // register-column kernel, smallest fitting size
        if (ref_max_len <= 64) e = launch_pairs_reg<64>(p, rp, rl, ref_max_len, revcomp_ref, qp, ql, query_max_len, npairs, out, st, ix);
        else if (ref_max_len <= 104) e = launch_pairs_reg<104>(p, rp, rl, ref_max_len, revcomp_ref, qp, ql, query_max_len, npairs, out, st, ix);
        else e = launch_pairs_reg<152>(p, rp, rl, ref_max_len, revcomp_ref, qp, ql, query_max_len, npairs, out, st, ix);
        return e == hipSuccess ? ATR_OK : hip_fail(e, "pairs_reg_kernel launch");
    }
    {   // more than PAIRS_REG_MAX rows: the register column in strips of 128 rows; the LDS-column kernel below
        // remains for a runtime without stream-ordered allocation
        e = launch_pairs_strips(p, rp, rl, ref_max_len, revcomp_ref, qp, ql, query_max_len, npairs, out, st, ix);
        if (e == hipSuccess) return ATR_OK;
        (void)hipGetLastError();
        if (e != hipErrorNotSupported && e != hipErrorOutOfMemory) return hip_fail(e, "pairs_strip_kernel launch");
    }
    const size_t lds = ((size_t)(ref_max_len + 1) + (size_t)(ref_max_len + 7) / 8) * 64 * 4;
    const dim3 grid((unsigned)((npairs + 63) / 64)), block(64);
    const bool xrep = (p.flags & ATR_STOP_WITHIN_SEQ2) != 0;           // mismatch-counting payload (pairs_core.hpp)
#define ATR_LAUNCH_PAIRS(MODE, XR)                                                                                     \
    do {                                                                                                               \
        e = hipFuncSetAttribute((const void *)pairs_kernel<MODE, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return hip_fail(e, "pairs_kernel LDS size");                                              \
        hipLaunchKernelGGL((pairs_kernel<MODE, XR>), grid, block, lds, st, p, rp, rl, (ref_max_len + 31) / 32, ref_max_len, \
                           revcomp_ref, qp, ql, (query_max_len + 31) / 32, query_max_len, npairs, out, ix);            \
    } while (0)
    if (p.and_mode) { if (xrep) ATR_LAUNCH_PAIRS(true, true); else ATR_LAUNCH_PAIRS(true, false); }
    else { if (xrep) ATR_LAUNCH_PAIRS(false, true); else ATR_LAUNCH_PAIRS(false, false); }
#undef ATR_LAUNCH_PAIRS
    e = hipGetLastError();
    return e == hipSuccess ? ATR_OK : hip_fail(e, "pairs_kernel launch");
}

// pairs_fast.hip
hipError_t launch_pairs_fast(const PairParams &p, double e_rate, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                             const uint32_t *qp, const int32_t *ql, int qmax, const int32_t *need, long long npairs,
                             uint4 *out, hipStream_t st);

hipError_t launch_pairs_wave(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                             const uint32_t *qp, const int32_t *ql, int qmax, const uint8_t *ref_codes, const uint8_t *qry_codes,
                             long long npairs, uint4 *out, hipStream_t st);

hipError_t launch_pairs_full_indexed(const PairParams &p, const uint32_t *rp, const int32_t *rl, int rmax, int revcomp,
                                     const uint32_t *qp, const int32_t *ql, int qmax, long long npairs, uint4 *out,
                                     const uint4 *order, const uint32_t *range, hipStream_t st) {
    const int rc = pairs_full(p, rp, rl, rmax, revcomp, qp, ql, qmax, npairs, out, st, PairIndex{order, range});
    return rc == ATR_OK ? hipSuccess : hipErrorUnknown;
}

}  // namespace atr

using namespace atr;

extern "C" int atr_locate_pairs_need_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len,
                                           int revcomp_ref, const uint8_t *d_query_packed, const int32_t *d_query_lens,
                                           int query_max_len, int64_t npairs, double max_error_rate, int flags,
                                           int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                                           const int32_t *d_need, atr_result *d_out, void *stream) {
    return atr_locate_pairs_path_batch(d_ref_packed, d_ref_lens, ref_max_len, revcomp_ref, d_query_packed, d_query_lens,
                                       query_max_len, npairs, max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap,
                                       indel_cost, d_need, ATR_PAIRS_AUTO, d_out, stream);
}

extern "C" int atr_locate_pairs_path_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len,
                                           int revcomp_ref, const uint8_t *d_query_packed, const int32_t *d_query_lens,
                                           int query_max_len, int64_t npairs, double max_error_rate, int flags,
                                           int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                                           const int32_t *d_need, int path, atr_result *d_out, void *stream) {
    if (npairs < 0 || path < ATR_PAIRS_AUTO || path > ATR_PAIRS_WAVE) return ATR_ERR_INVALID;
    PairParams p;
    const int rc = pairs_params(max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost, ref_max_len,
                                query_max_len, p);
    if (rc != ATR_OK) return rc;
    if (npairs == 0) return ATR_OK;
    if (!d_out || (ref_max_len > 0 && !d_ref_packed) || (query_max_len > 0 && !d_query_packed)) return ATR_ERR_INVALID;
    const uint32_t *rp = (const uint32_t *)d_ref_packed, *qp = (const uint32_t *)d_query_packed;
    if (path == ATR_PAIRS_WAVE && !wave_pairs_applies(ref_max_len, 0)) return ATR_ERR_UNSUPPORTED;
    if (path == ATR_PAIRS_WAVE || (path == ATR_PAIRS_AUTO && wave_pairs_applies(ref_max_len, (long long)npairs))) {
        // short batch: a wavefront per pair (pairs_wave.hip); a lane per pair is one chain of m x n dependent cells
        const hipError_t e = launch_pairs_wave(p, rp, d_ref_lens, ref_max_len, revcomp_ref, qp, d_query_lens, query_max_len,
                                               nullptr, nullptr, (long long)npairs, (uint4 *)d_out, (hipStream_t)stream);
        return e == hipSuccess ? ATR_OK : hip_fail(e, "pairs_wave_kernel launch");
    }
    // The cost / threat / band pipeline from a quarter of a million pairs on: its launches cost 0.65 ms (2 x 150 bp; the
    // latency of single lanes) whatever the batch holds, the full sweep 5.5 ns per pair with a floor of 0.36 ms
    // (tools/micro/small_pairs.py, pairs_fixed_cost.py).  Reads of more than 160 bases have bands almost as wide as the
    // matrix (2 k + 1 diagonals on either side of the overlap's end, k = 50 at 250 bases): 82 M against 75 M pairs/s on
    // 500 k pairs 2 x 250 bp, so they need more pairs to pay -- unless d_need ends most pairs after the cost pass.
    const long long fast_min = (ref_max_len <= 160 || d_need != nullptr) ? PAIRS_FAST_MIN_PAIRS : 3 * PAIRS_FAST_MIN_PAIRS / 2;
    const bool worth = path == ATR_PAIRS_FAST || npairs >= fast_min;
    if (path != ATR_PAIRS_FULL && worth && pairs_fast_applies(max_error_rate, flags, wildcard_ref, wildcard_query, indel_cost, ref_max_len, query_max_len) &&
        npairs < (1ll << 32)) {
        // costs by bit-vector, threats, banded payload (pairs_fast_core.hpp); pairs outside its envelope take the
        // full sweep inside the same call
        const hipError_t e = launch_pairs_fast(p, max_error_rate, rp, d_ref_lens, ref_max_len, revcomp_ref, qp, d_query_lens,
                                               query_max_len, d_need, (long long)npairs, (uint4 *)d_out, (hipStream_t)stream);
        if (e == hipSuccess) return ATR_OK;
        (void)hipGetLastError();
        if (e != hipErrorNotSupported && e != hipErrorOutOfMemory) return hip_fail(e, "pairs fast pipeline");
    }
    return pairs_full(p, rp, d_ref_lens, ref_max_len, revcomp_ref, qp, d_query_lens, query_max_len, (long long)npairs,
                      (uint4 *)d_out, (hipStream_t)stream, PairIndex{nullptr, nullptr});
}

extern "C" int atr_locate_pairs_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len,
                                      int revcomp_ref, const uint8_t *d_query_packed, const int32_t *d_query_lens,
                                      int query_max_len, int64_t npairs, double max_error_rate, int flags,
                                      int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                                      atr_result *d_out, void *stream) {
    return atr_locate_pairs_need_batch(d_ref_packed, d_ref_lens, ref_max_len, revcomp_ref, d_query_packed, d_query_lens,
                                       query_max_len, npairs, max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap,
                                       indel_cost, nullptr, d_out, stream);
}

// The full sweep of every pair, whatever the settings: the path of pairs_core.hpp alone (the fast pipeline's own
// fallback, and what the parity tests compare it against at size).
extern "C" int atr_locate_pairs_full_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len,
                                           int revcomp_ref, const uint8_t *d_query_packed, const int32_t *d_query_lens,
                                           int query_max_len, int64_t npairs, double max_error_rate, int flags,
                                           int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                                           atr_result *d_out, void *stream) {
    if (npairs < 0) return ATR_ERR_INVALID;
    PairParams p;
    const int rc = pairs_params(max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost, ref_max_len,
                                query_max_len, p);
    if (rc != ATR_OK) return rc;
    if (npairs == 0) return ATR_OK;
    if (!d_out || (ref_max_len > 0 && !d_ref_packed) || (query_max_len > 0 && !d_query_packed)) return ATR_ERR_INVALID;
    return pairs_full(p, (const uint32_t *)d_ref_packed, d_ref_lens, ref_max_len, revcomp_ref, (const uint32_t *)d_query_packed,
                      d_query_lens, query_max_len, (long long)npairs, (uint4 *)d_out, (hipStream_t)stream,
                      PairIndex{nullptr, nullptr});
}

// Aligner(reference, ...).locate(query) for ONE pair in host memory, synchronously (what MergeOverlapping does per read
// pair through the module swap when the reads are longer than an aligner handle's 128 bases): ref_codes / query_codes
// are the strings translated to 4-bit codes, one per byte, with the tables of atr_locate_pairs_batch.
namespace {
struct PairShot {
    uint8_t *base = nullptr, *dev = nullptr;
    int device = -1;
    bool ready() {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return false;
        if (base && cur == device) return true;
        if (base) { (void)hipHostFree(base); base = nullptr; }
        void *p = nullptr, *dp = nullptr;
        if (hipHostMalloc(&p, 1024, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return false;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) { (void)hipHostFree(p); return false; }
        base = (uint8_t *)p; dev = (uint8_t *)dp; device = cur;
        return true;
    }
};
}  // namespace

extern "C" int atr_locate_pair_one(const uint8_t *ref_codes, int m, int revcomp_ref, const uint8_t *query_codes, int n,
                                   double max_error_rate, int flags, int wildcard_ref, int wildcard_query, int min_overlap,
                                   int indel_cost, atr_result *out, void *stream) {
    if (m < 0 || n < 0 || (m > 0 && !ref_codes) || (n > 0 && !query_codes) || !out) return ATR_ERR_INVALID;
    PairParams p;
    const int rc = pairs_params(max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap, indel_cost, m, n, p);
    if (rc != ATR_OK) return rc;
    if (!wave_pairs_applies(m, 1)) return ATR_ERR_UNSUPPORTED;
    static_assert(2 * 320 + 64 <= 1024 && ATR_PAIRS_MAX_LEN <= 320, "staging layout of atr_locate_pair_one");
    static thread_local PairShot shot;
    if (!shot.ready()) return hip_fail(hipErrorOutOfMemory, "hipHostMalloc(pair staging)");
    std::memcpy(shot.base + 64, ref_codes, (size_t)m);                      // [record: 16 B, pad][reference: 320][query: 320]
    std::memcpy(shot.base + 64 + 320, query_codes, (size_t)n);
    const hipError_t e = launch_pairs_wave(p, nullptr, nullptr, m, revcomp_ref, nullptr, nullptr, n, shot.dev + 64, shot.dev + 64 + 320,
                                           1, (uint4 *)shot.dev, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "pairs_wave_kernel launch");
    const hipError_t s = hipStreamSynchronize((hipStream_t)stream);
    if (s != hipSuccess) return hip_fail(s, "hipStreamSynchronize");
    *out = *(const atr_result *)shot.base;
    return ATR_OK;
}
