// locate_kernel.hpp -- the gfx950 kernel around locate_core.hpp and its launcher.
//
// Mapping: one read per lane, 64 reads (one tile64 of the packed layout) per
// wavefront, 4 wavefronts per workgroup; a launch needs nreads/256 workgroups, i.e.
// tens of thousands for a 10 M-read batch on 256 CUs.  The only global traffic is
// one coalesced 1 KiB chunk load per wave per 32 columns (prefetched one chunk
// ahead) and one coalesced 16-byte result store per read; there is no reuse between
// workgroups, so no XCD-specific block mapping is needed.
#ifndef ATR_LOCATE_KERNEL_HPP
#define ATR_LOCATE_KERNEL_HPP

#include <hip/hip_runtime.h>
#include "aligner_host.hpp"
#include "fast_work.hpp"

namespace atr {

template <int W>
__device__ __forceinline__ void load_mask(uint32_t (&nm)[W], const uint32_t (*s_nm)[4], uint32_t q) {
#pragma unroll
    for (int w = 0; w < W; ++w) nm[w] = s_nm[q][w];
}

template <int MT, bool NOINDEL, bool XREP>
__global__ __launch_bounds__(256) void locate_kernel(const LocateParams p, const uint4 *__restrict__ packed,
                                                     const int32_t *__restrict__ lens, long long nreads,
                                                     int nchunks, int max_len, uint4 *__restrict__ out) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_init[ATR_MAX_REF_LEN + 1];        // by position
    __shared__ __attribute__((aligned(16))) uint32_t s_nm[16][4];

    const Uniform u = make_uniform(p, MT);

    // Stage the wave-uniform state in LDS: mismatch masks, column-init state (min_n == 0),
    // thresholds.
    if (threadIdx.x < 64) s_nm[threadIdx.x >> 2][threadIdx.x & 3] = p.nmask[threadIdx.x >> 2][threadIdx.x & 3];
    for (int i = threadIdx.x; i <= MT + 1; i += 256) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nreads + 63) >> 6;
    if (tile >= ntiles) return;                              // whole wave
    const long long r = tile * 64 + lane;
    const bool live = r < nreads;
    const int n = live ? (lens ? lens[r] : max_len) : 0;

    LaneState<MT> L;
    lane_init<MT, NOINDEL, XREP>(L, u, n, s_init, s_thr);

    // Wave-uniform column range (jlo, jhi]: the union of the lanes' windows.
    const bool has_window = live && L.max_n > L.min_n;
    const int jlo = wave_min_i32(has_window ? L.min_n : 0x7fffffff);
    const int jhi = wave_max_i32(has_window ? L.max_n : 0);
    const uint4 *tp = packed + (size_t)tile * nchunks * 64 + lane;

    if (jhi > jlo) {
        const int c0 = jlo >> 5, c1 = (jhi + 31) >> 5;       // chunks [c0, c1)
        uint4 nxt = tp[(size_t)c0 * 64];
        for (int c = c0; c < c1; ++c) {
            uint4 cur = nxt;
            if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];  // prefetch the next 1 KiB burst
            int j = c * 32;
#pragma unroll 1
            for (int d = 0; d < 4; ++d) {
                uint32_t w = cur.x;
                cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
#pragma unroll 1
                for (int b = 0; b < 8; ++b) {
                    ++j;
                    const uint32_t q = w & 15u;
                    w >>= 4;
                    if (j <= jlo || j > jhi) continue;       // wave-uniform (first / last chunk only)
                    uint32_t nm[(MT + 31) / 32];
                    load_mask(nm, s_nm, q);
                    lane_step<MT, NOINDEL, XREP>(L, u, j, nm, s_thr);
                }
            }
        }
    }

    if (live) {
        uint32_t rec[4];
        lane_result<MT>(L, u, rec);
        out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

// Reads of more than ATR_MAX_READ_LEN bases (up to ATR_MAX_LONG_READ_LEN): the same sweep with the rolling origin
// base of locate_core.hpp (long_base / lane_rebase).  The base is a function of the column alone, so it is the same
// in every lane; a lane's best match remembers the base it was found under.
template <int MT, bool NOINDEL, bool XREP>
__global__ __launch_bounds__(256) void locate_long_kernel(const LocateParams p, const uint4 *__restrict__ packed,
                                                          const int32_t *__restrict__ lens, long long nreads,
                                                          int nchunks, int max_len, uint4 *__restrict__ out) {
    __shared__ int16_t s_thr[ATR_MAX_REF_LEN + 2];
    __shared__ uint32_t s_init[ATR_MAX_REF_LEN + 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_nm[16][4];
    const Uniform u = make_uniform(p, MT);
    if (threadIdx.x < 64) s_nm[threadIdx.x >> 2][threadIdx.x & 3] = p.nmask[threadIdx.x >> 2][threadIdx.x & 3];
    for (int i = threadIdx.x; i <= MT + 1; i += 256) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nreads + 63) >> 6;
    if (tile >= ntiles) return;
    const long long r = tile * 64 + lane;
    const bool live = r < nreads;
    const int n = live ? (lens ? lens[r] : max_len) : 0;

    LaneState<MT> L;
    lane_init<MT, NOINDEL, XREP>(L, u, n, s_init, s_thr);
    const bool has_window = live && L.max_n > L.min_n;
    const int jlo = wave_min_i32(has_window ? L.min_n : 0x7fffffff);
    const int jhi = wave_max_i32(has_window ? L.max_n : 0);
    const uint4 *tp = packed + (size_t)tile * nchunks * 64 + lane;
    int obase = 0, best_base = 0;
    if (jhi > jlo) {
        obase = long_base(jlo + 1);
        // (lane_init wrote the initial column with origins counted from 0: an origin beyond the field ran into the
        // payload bits above it, which are zero in an initial column, and comes back into range here)
        if (!XREP) {
#pragma unroll
            for (int i = 0; i <= MT; ++i) L.col[i] -= (uint32_t)obase;
        }
        const int c0 = jlo >> 5, c1 = (jhi + 31) >> 5;
        uint4 nxt = tp[(size_t)c0 * 64];
        for (int c = c0; c < c1; ++c) {
            uint4 cur = nxt;
            if (c + 1 < c1) nxt = tp[(size_t)(c + 1) * 64];
            int j = c * 32;
            if (long_base(j + 1) != obase) {                 // wave-uniform; steps fall on chunk boundaries
                lane_rebase<MT>(L);
                obase += LONG_BASE_STEP;
            }
#pragma unroll 1
            for (int d = 0; d < 4; ++d) {
                uint32_t w = cur.x;
                cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
#pragma unroll 1
                for (int b = 0; b < 8; ++b) {
                    ++j;
                    const uint32_t q = w & 15u;
                    w >>= 4;
                    if (j <= jlo || j > jhi) continue;
                    uint32_t nm[(MT + 31) / 32];
                    load_mask(nm, s_nm, q);
                    const int key = L.best.key;
                    lane_step<MT, NOINDEL, XREP>(L, u, j, nm, s_thr, MT, obase);
                    if (L.best.key != key) best_base = obase;
                }
            }
        }
    }
    if (live) {
        uint32_t rec[4];
        lane_result<MT>(L, u, rec, best_base);
        out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

typedef int (*locate_launcher)(const atr_aligner *, const uint4 *, const int32_t *, long long, int, int,
                               uint4 *, hipStream_t);

template <int MT>
int launch_locate_mt(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads,
                     int nchunks, int max_len, uint4 *out, hipStream_t st) {
    const bool noindel = a->indel_cost > a->p.k;
    const long long ntiles = (nreads + 63) / 64;
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
    const bool xrep = (a->flags & ATR_STOP_WITHIN_SEQ2) != 0;     // mismatch-count payload (locate_core.hpp)
    if (max_len > ATR_MAX_READ_LEN) {
        if (a->p.m + a->p.k > LONG_MAX_SPAN) return (int)hipErrorInvalidValue;
        if (xrep) {
            if (noindel) hipLaunchKernelGGL((locate_long_kernel<MT, true, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
            else         hipLaunchKernelGGL((locate_long_kernel<MT, false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
        } else {
            if (noindel) hipLaunchKernelGGL((locate_long_kernel<MT, true, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
            else         hipLaunchKernelGGL((locate_long_kernel<MT, false, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
        }
        return (int)hipGetLastError();
    }
    if (xrep) {
        if (noindel) hipLaunchKernelGGL((locate_kernel<MT, true, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
        else         hipLaunchKernelGGL((locate_kernel<MT, false, true>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    } else {
        if (noindel) hipLaunchKernelGGL((locate_kernel<MT, true, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
        else         hipLaunchKernelGGL((locate_kernel<MT, false, false>), grid, block, 0, st, a->p, packed, lens, nreads, nchunks, max_len, out);
    }
    return (int)hipGetLastError();
}

constexpr int LOCATE_GROUPS = 8;                             // instantiation units (parallel compilation)
constexpr int LOCATE_PER_GROUP = ATR_MAX_REF_LEN / ROW_GRAN / LOCATE_GROUPS;   // 4 sizes each

}  // namespace atr
#endif
