// piece_kernels.hip -- the two-pass pre-pass on plane64 reads (piece_core.hpp) and its driver:
//
//   P1  piece_filter_kernel<NW>   pass A over every tile (pieces, 32 positions per op), pass B over the flagged
//                                 reads' windows (a per-wave LDS queue, one task per lane); resolved reads get their
//                                 record, the others go into the block's list of (read, window word); the few reads
//                                 whose hits need more than PIECE_NARROW columns are listed per block (with a copy of
//                                 their planes) and take the full bit-vector sweep at the end of the kernel, 64 to a wave
//   K2  scan_bins / scan_total    (locate_fast.hpp, unchanged)
//   P3  piece_scatter_kernel      the block's list -> the bins of `order`
//   K4a / K4                      band_kernel / window_kernel<.., PLANES>: the exact DP, reading plane64
//
// Same records as the one-pass pipeline (filter_kernels.hip) and as the full sweep; tests/test_gpu_locate.py
// compares all three on every size.
#include "locate_fast.hpp"
#include "piece_filter.hpp"
#include "jit.hpp"

namespace atr {

void launch_fast_scan(FastWork wk, hipStream_t st);                              // filter_kernels.hip
int launch_fast_dp(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                   int max_len, uint4 *out, FastWork wk, const LinkedArgs *la, int idx, int count, hipStream_t st, bool planes, bool one_stream);

template <int NW, bool RAGGED, int WW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ATR_PIECE_WAVES(NW), 8))) void piece_filter_kernel(
    const LocateParams p, const FilterParams fp, const PieceParams pp, const uint4 *__restrict__ planes,
    const int32_t *__restrict__ lens, long long nreads, int max_len, uint4 *__restrict__ out, FastWork wk) {
    piece_filter_body<NW, RAGGED, WW>(p, fp, pp, planes, lens, nreads, max_len, out, wk);
}

// P3: a block's list of (read, window word) -> the bins of `order` (offsets from K2's scan, an LDS cursor per bin)
__global__ __launch_bounds__(256) void piece_scatter_kernel(long long nreads, int m, FastWork wk) {
    __shared__ uint32_t s_cur[FILTER_BINS + 1], s_tmp[256];
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    block_tiles(ntiles, t0, t1, wk.nused);
    // the block is one chain of dependent round trips (totals -> offsets -> entry -> LDS cursor -> store): everything that
    // does not depend on the scan is requested before it -- the block's offsets, its list length, every thread's first entry
    const uint2 *list = wk.tmp + t0 * 64;
    const uint32_t count = wk.lcount[blockIdx.x];
    const uint32_t mine = threadIdx.x < FILTER_BINS ? wk.counts[(size_t)blockIdx.x * FILTER_BINS + threadIdx.x] : 0u;
    const uint2 first = threadIdx.x < count ? list[threadIdx.x] : make_uint2(0u, 0u);
    if (wk.fused) {
        fused_bin_bases(wk, s_cur, s_tmp);
        if (threadIdx.x < FILTER_BINS) s_cur[threadIdx.x] += mine;
    } else if (threadIdx.x < FILTER_BINS) s_cur[threadIdx.x] = fast_slot0(wk, threadIdx.x);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        const uint2 e = i == threadIdx.x ? first : list[i];
        const uint32_t slot = atomicAdd(&s_cur[window_bin(e.y & ~PIECE_NODENSE, m, true)], 1u);
        wk.order[slot] = e;
        wk.dref[slot] = (uint32_t)(t0 * 64) + i;                  // the entry's 64-code record in tdata
    }
}

// Blocks of P1 the device holds at once: P1 is a persistent grid -- a wave's task queue wants many tiles, and a grid
// of exactly the resident blocks has no partial last round (2048 blocks on 768 slots cost a third round: 0.72 ms
// instead of 0.5).  Per thread and device: the occupancy query is not free.
template <int NW, bool RAGGED, int WW>
static int piece_resident_blocks() {
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (dev != cached_dev) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, piece_filter_kernel<NW, RAGGED, WW>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached = std::min(FAST_BLOCKS, per_cu * cus);
        cached_dev = dev;
    }
    return cached;
}

template <int NW, int WW>
static void launch_piece_filter_w(const atr_aligner *a, const FilterParams &fp, const PieceParams &pp, const uint4 *planes,
                                  const int32_t *lens, long long nreads, int max_len, uint4 *out, FastWork &wk, hipStream_t st) {
    const long long want = ((nreads + 63) / 64 + 3) / 4;
    if (lens) {
        wk.nused = (int)std::max<long long>(1, std::min<long long>(piece_resident_blocks<NW, true, WW>(), want));
        hipLaunchKernelGGL((piece_filter_kernel<NW, true, WW>), dim3(wk.nused), dim3(256), 0, st, a->p, fp, pp, planes, lens, nreads, max_len, out, wk);
    } else {
        wk.nused = (int)std::max<long long>(1, std::min<long long>(piece_resident_blocks<NW, false, WW>(), want));
        hipLaunchKernelGGL((piece_filter_kernel<NW, false, WW>), dim3(wk.nused), dim3(256), 0, st, a->p, fp, pp, planes, lens, nreads, max_len, out, wk);
    }
}
template <int NW>
static void launch_piece_filter(const atr_aligner *a, const FilterParams &fp, const PieceParams &pp, const uint4 *planes,
                                const int32_t *lens, long long nreads, int max_len, uint4 *out, FastWork &wk, hipStream_t st) {
    // (the 96-column instantiation is also the one with accumulators for more than five body pieces and the two-word sweep)
    if (pp.window > PIECE_WINDOW || pp.nb > 5 || fp.rows > 32) launch_piece_filter_w<NW, 3>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st);
    else launch_piece_filter_w<NW, 2>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st);
}

// Does the two-pass pre-pass take this aligner on equal-length reads of max_len bases?  (+ its parameters)
// A ragged batch of reads of at most max_len bases is the equal-length case of 32 ceil(max_len / 32) bases
// (piece_filter_kernel<.., RAGGED> moves every read to the end of its words): see piece_ragged_len().
int piece_ragged_len(int max_len) { return 32 * ((max_len + 31) / 32); }
bool piece_applies(const atr_aligner *a, int max_len, FilterParams *fp_out, PieceParams *pp_out) {
    if (!a->filterable || max_len < 1) return false;
    const FilterParams &fp = aligner_piece_filter_params(a);
    PieceParams pp;
    if (!piece_params(a->codes, a->p.m, fp.rows, a->p.k, a->flags, a->wildcard_ref || a->wildcard_query,
                      a->table_kind == ATR_TABLE_CUSTOM, fp.thr_row, max_len, pp, a->p.thr, a->p.min_overlap)) return false;
    const int nw = (max_len + 31) / 32;
    if (nw < 3 || nw > 10) return false;                           // instantiated word counts: reads of 65 .. 320 bases
    if (fp_out) *fp_out = fp;
    if (pp_out) *pp_out = pp;
    return true;
}

// The run-time compiled pre-pass of this aligner for reads of max_len bases (jit.hpp), or nullptr: the generic kernel.
// force: compile now if no object exists yet (atr_aligner_prepare; ATR_JIT=1); otherwise jit.hpp's policy decides.
static const jit::SpecKernel *piece_spec_for(const atr_aligner *a, const FilterParams &fp, const PieceParams &pp, bool ragged,
                                             int max_len, long long nreads, bool force, int ascii_stride = 0) {
    const int pol = jit::policy();
    if (pol == 0) return nullptr;
    const int nw = (max_len + 31) / 32;
    // auto: once the handle has seen $ATR_JIT_MIN_READS reads in all (an aligner lives for a run: the 0.8 s of hiprtc are
    // then paid back within a few hundred million reads, and at once from the second run on -- the disk cache)
    a->planes_seen += nreads;
    const bool compile = force || pol == 1 || a->planes_seen >= jit::min_reads();
    return jit::spec_kernel(a, fp, pp, nw, ragged, ragged ? 32 * nw : max_len, compile, ascii_stride);
}

// 1: a specialised kernel is ready for (aligner, max_len, ragged) on the current device; 0: there is none (outside the
// two-pass envelope, ATR_JIT=0, no hiprtc, compile error) -- the generic kernel then serves the calls.
int prepare_locate_planes(const atr_aligner *a, int max_len, bool ragged) {
    FilterParams fp;
    PieceParams pp;
    if (!piece_applies(a, ragged ? piece_ragged_len(max_len) : max_len, &fp, &pp)) return 0;
    return piece_spec_for(a, fp, pp, ragged, max_len, 0, true) != nullptr ? 1 : 0;
}

// lens == nullptr: every read has max_len bases.  The call in two halves, so that a caller with several batches
// (linked_group.hip: a sub-batch per 3' adapter) can put all pre-passes on the device before the first DP tail:
//   launch_planes_prepass   P1 (run-time compiled when there is such a kernel); grid_div > 1: the persistent grid takes
//                           that share of the resident blocks
//   launch_planes_tail      scan (unless fused), P3, K4a || K4; one_stream: see launch_fast_dp
int launch_planes_prepass(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len,
                          uint4 *out, void *work, hipStream_t st, int grid_div, PlanesCall *pc) {
    FilterParams fp;
    PieceParams pp;
    if (!piece_applies(a, lens ? piece_ragged_len(max_len) : max_len, &fp, &pp)) return (int)hipErrorInvalidValue;
    FastWork wk = fast_carve(work, nreads);
    wk.lpw = nreads <= 8192 ? 0 : 64;
    const int nw = (max_len + 31) / 32;
    const jit::SpecKernel *sk = piece_spec_for(a, fp, pp, lens != nullptr, max_len, nreads, false);
    wk.fused = fast_fused_scan() ? 1 : 0;
    if (wk.fused) {   // the bins' totals the pre-pass blocks add to (fused_hist_flush)
        const hipError_t rc = hipMemsetAsync(wk.chunks, 0, (size_t)wk.nbins * 4, st);
        if (rc != hipSuccess) return (int)rc;
    }
    if (sk) {
        const long long want = ((nreads + 63) / 64 + 3) / 4;
        wk.nused = (int)std::max<long long>(1, std::min<long long>(std::max(1, sk->resident / std::max(1, grid_div)), want));
        const hipError_t rc = jit::spec_launch(sk, wk.nused, planes, lens, nreads, max_len, out, wk, st);
        if (rc != hipSuccess) return (int)rc;
    } else {
        switch (nw) {
        case 3: launch_piece_filter<3>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 4: launch_piece_filter<4>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 5: launch_piece_filter<5>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 6: launch_piece_filter<6>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 7: launch_piece_filter<7>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 8: launch_piece_filter<8>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        case 9: launch_piece_filter<9>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        default: launch_piece_filter<10>(a, fp, pp, planes, lens, nreads, max_len, out, wk, st); break;
        }
    }
    pc->wk = wk;
    pc->nw = nw;
    return (int)hipGetLastError();
}

int launch_planes_tail(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len, uint4 *out,
                       const PlanesCall &pc, hipStream_t st, bool one_stream) {
    FastWork wk = pc.wk;
    if (!wk.fused) launch_fast_scan(wk, st);
    hipLaunchKernelGGL(piece_scatter_kernel, dim3(wk.nused), dim3(256), 0, st, nreads, a->p.m, wk);
    return launch_fast_dp(a, planes, lens, nreads, pc.nw, max_len, out, wk, nullptr, 0, 1, st, /*planes=*/true, one_stream);
}

int launch_locate_planes_shared(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len,
                                uint4 *out, void *work, hipStream_t st, int grid_div, bool one_stream) {
    PlanesCall pc;
    const int rc = launch_planes_prepass(a, planes, lens, nreads, max_len, out, work, st, grid_div, &pc);
    if (rc != 0) return rc;
    return launch_planes_tail(a, planes, lens, nreads, max_len, out, pc, st, one_stream);
}

// The fused ASCII entry (piece_filter.hpp, ATR_SPEC_ASCII): rows of `row_stride` bytes in, the packed plane64 batch
// (`planes`, written) and the records out, in ONE pre-pass launch + the DP tail.  Exists only as a run-time compiled
// kernel (aligner, read length and row stride are constants of the build): returns 1 when there is none -- outside the
// envelope, no hiprtc, reads of more than 256 bases, ATR_JIT=0 -- and the caller packs and calls launch_locate_planes.
int launch_locate_ascii_fused(const atr_aligner *a, const uint8_t *ascii, long long row_stride, const int32_t *lens, long long nreads,
                              int max_len, const uint8_t table[256], uint4 *planes, uint4 *out, void *work, hipStream_t st) {
    FilterParams fp;
    PieceParams pp;
    if (!piece_applies(a, lens ? piece_ragged_len(max_len) : max_len, &fp, &pp)) return 1;
    const int nw = (max_len + 31) / 32;
    static const bool off = [] { const char *x = getenv("ATR_ASCII_FUSED"); return x && x[0] == '0'; }();      // (A/B switch)
    if (off || nw > 8 || row_stride < max_len || row_stride > 256 || ((uintptr_t)ascii & 1)) return 1;
    // (an entry for batches that are long by definition: compile at the first call unless the policy says never)
    const jit::SpecKernel *sk = piece_spec_for(a, fp, pp, lens != nullptr, max_len, nreads, jit::policy() != 0, (int)row_stride);
    if (!sk) return 1;
    FastWork wk = fast_carve(work, nreads);
    wk.lpw = nreads <= 8192 ? 0 : 64;
    wk.fused = fast_fused_scan() ? 1 : 0;
    if (wk.fused) {
        const hipError_t rc = hipMemsetAsync(wk.chunks, 0, (size_t)wk.nbins * 4, st);
        if (rc != hipSuccess) return (int)rc;
    }
    const long long want = ((nreads + 63) / 64 + 3) / 4;
    wk.nused = (int)std::max<long long>(1, std::min<long long>(sk->resident, want));
    const hipError_t rc = jit::spec_launch_ascii(sk, wk.nused, ascii, lens, nreads, max_len, out, wk, planes, table, st);
    if (rc != hipSuccess) return (int)rc;
    PlanesCall pc;
    pc.wk = wk;
    pc.nw = nw;
    return launch_planes_tail(a, planes, lens, nreads, max_len, out, pc, st, false);
}

int launch_locate_planes(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len, uint4 *out,
                         void *work, hipStream_t st) {
    return launch_locate_planes_shared(a, planes, lens, nreads, max_len, out, work, st, 1, false);
}

}  // namespace atr
