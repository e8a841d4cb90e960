// filter_kernels.hip -- K1..K3 of the filtered locate pipeline and its driver.
#define ATR_DEFINE_FILTER_KERNELS
#include "locate_fast.hpp"
#include "side_stream.hpp"

namespace atr {

constexpr long long FAST_SERIAL_READS = 65536;       // "short batch": no side stream, a wave per few tasks
constexpr long long FAST_SERIAL_READS_LINKED = 64;   // the same for the adapters of a linked set (their launches would overlap)

window_launcher window_group_0(int), window_group_1(int), window_group_2(int), window_group_3(int);

static BandParams band_params(const atr_aligner *a) {
    BandParams bp;
    memset(&bp, 0, sizeof(bp));
    for (int i = 0; i < a->p.m && i < FILTER_MAX_M; ++i) bp.rrep[i] = (uint32_t)(a->codes[i] & 15u) * 0x11111111u;
    bp.and_mode = (a->wildcard_ref || a->wildcard_query) ? 1 : 0;
    bp.noindel = a->indel_cost > a->p.k ? 1 : 0;
    return bp;
}

int launch_linked_band(const void *d_wave, int nad, bool and_mode, const uint4 *packed, const int32_t *lens, long long nreads,
                       int nchunks, int max_len, const uint4 *front, uint4 *out, FastWork wk, hipStream_t st);   // linked_kernels.hip

static LinkedArgs no_linked_args() {
    LinkedArgs la;
    memset(&la, 0, sizeof(la));
    return la;
}

typedef window_launcher (*window_group_fn)(int);
static const window_group_fn window_groups[4] = {window_group_0, window_group_1, window_group_2, window_group_3};

// K2a + K2b over wk.nbins bins
void launch_fast_scan(FastWork wk, hipStream_t st) {
    hipLaunchKernelGGL(scan_bins_kernel, dim3((wk.nused + SCAN_CHUNK - 1) / SCAN_CHUNK, wk.nbins / 256), dim3(256), 0, st, wk);
    hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(1024), 0, st, wk);
}

// K4a and K4 over the reads `order` lists for one aligner; la != nullptr: the 3' part of adapter
// `idx` of a linked set of `count` adapters (bins la->bin0 ..).  K4a and K4 work on disjoint slots of `order` and
// write disjoint records: they run side by side (K4a's waves are latency bound: three dependent gathers per task --
// they share the SIMDs with K4's instead of running before them), K4a on the caller's stream, the K4 launches of
// the adapters of a linked set on side streams.  idx == 0 forks the side streams off `st` (they wait for the scatter
// pass), idx == count - 1 joins them again; events only, legal inside a stream capture.
// (A band stream per adapter was tried in round 4: the four K4a launches of C4 then run side by side, each four
// times as long -- the DP phase is bound by its task throughput, not by launch latency.  Round 5, from the kernel
// timelines of profiles/round5_tail_timelines.txt: with five streams C4's last window launch waited for another
// stream to drain -- three side streams now, a fourth adapter's window launch queues behind the first's.)
constexpr int DP_STREAMS = 4;                            // >= LINKED_MAX: band stream + one per further adapter
// one_stream: both DP launches on `st` whatever the batch size (linked_group.hip runs several calls side by side, a stream
// each: the device has four hardware queues, a fifth stream waits for one of them to drain)
int launch_fast_dp(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                   int max_len, uint4 *out, FastWork wk, const LinkedArgs *la, int idx, int count, hipStream_t st, bool planes,
                   bool one_stream) {
    static thread_local SideStream side[DP_STREAMS];
    if (planes && la) return (int)hipErrorInvalidValue;                  // plane64: the single-aligner pipeline
    if (count < 1 || count > DP_STREAMS || idx < 0 || idx >= count) return (int)hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    // a short batch (the <= 1000 reads the unchanged trim command hands over per call) leaves most of the chip
    // idle anyway: the fork / join events would cost more than the overlap gives, both DP kernels go to `st`
    const bool serial = one_stream || nreads <= (count == 1 ? FAST_SERIAL_READS : FAST_SERIAL_READS_LINKED);
    if (serial) {
        const BandParams bp = band_params(a);
        const dim3 bgrid((unsigned)std::max<long long>(1, std::min<long long>((nreads + 3) / 4, 4096)));
        const LinkedArgs none = la ? *la : no_linked_args();
        if (la) {
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, true>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
            else             hipLaunchKernelGGL((band_kernel<false, true>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
        } else if (planes) {
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, false, true>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
            else             hipLaunchKernelGGL((band_kernel<false, false, true>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
        } else {
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, false>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
            else             hipLaunchKernelGGL((band_kernel<false, false>), bgrid, dim3(256), 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        const int mts = round_up_rows(a->p.m) / ROW_GRAN - 1;
        return window_groups[mts / 4](mts % 4)(a, packed, lens, nreads, nchunks, max_len, out, wk, la, st, planes);
    }
    const int nside = (la && la->win_count != 1) ? 1 : std::min(count, 3);      // (one window launch for the whole set: one side stream)
    hipStream_t bst = st;
    if (idx == 0) {
        static const bool urgent = [] { const char *x = getenv("ATR_WINDOW_PRIORITY"); return !(x && x[0] == '0'); }();   // (A/B switch)
        for (int k = 0; k < nside; ++k) if (!side[k].ready(urgent)) return (int)hipErrorInvalidValue;
        e = hipEventRecord(side[0].fork, st);
        for (int k = 0; k < nside && e == hipSuccess; ++k) e = hipStreamWaitEvent(side[k].stream, side[0].fork, 0);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 block(256);
    {   // K4a: banded DP over the band reads
        const BandParams bp = band_params(a);
        const dim3 bgrid((unsigned)std::min<long long>((nreads + 255) / 256, 4096));
        if (la && la->multi) {                                      // a linked set: all adapters' band reads at once
            if (idx == 0) {
                const int rc = launch_linked_band(la->multi, count, la->multi_and, packed, lens, nreads, nchunks, max_len, la->front,
                                                  out, wk, bst);
                if (rc != 0) return rc;
            }
        } else if (la) {
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, true>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, *la);
            else             hipLaunchKernelGGL((band_kernel<false, true>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, *la);
        } else if (planes) {
            const LinkedArgs none = no_linked_args();
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, false, true>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
            else             hipLaunchKernelGGL((band_kernel<false, false, true>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
        } else {
            const LinkedArgs none = no_linked_args();
            if (bp.and_mode) hipLaunchKernelGGL((band_kernel<true, false>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
            else             hipLaunchKernelGGL((band_kernel<false, false>), bgrid, block, 0, bst, a->p, bp, packed, lens, nreads, nchunks, max_len, out, wk, none);
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int mt = round_up_rows(a->p.m) / ROW_GRAN - 1;
    const int rc = window_groups[mt / 4](mt % 4)(a, packed, lens, nreads, nchunks, max_len, out, wk, la,
                                                 side[idx % nside].stream, planes);
    if (rc != 0) return rc;
    if (idx == count - 1) {
        for (int k = 0; k < nside && e == hipSuccess; ++k) {
            e = hipEventRecord(side[k].join, side[k].stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, side[k].join, 0);
        }
    }
    return (int)e;
}

int launch_locate_fast(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads,
                       int nchunks, int max_len, uint4 *out, void *work, hipStream_t st) {
    FastWork wk = fast_carve(work, nreads);
    wk.nused = fast_blocks_for((nreads + 63) / 64);
    wk.lpw = nreads <= 8192 ? 0 : 64;
    const FilterParams &fp = aligner_filter_params(a);
    const bool wide = fp.rows > 32, ragged = lens != nullptr;
    const dim3 grid(wk.nused), block(256);
    if (wide) {
        if (ragged) hipLaunchKernelGGL((filter_kernel<true, true>), grid, block, 0, st, a->p, fp, packed, lens, nreads, nchunks, max_len, out, wk);
        else        hipLaunchKernelGGL((filter_kernel<true, false>), grid, block, 0, st, a->p, fp, packed, lens, nreads, nchunks, max_len, out, wk);
    } else {
        if (ragged) hipLaunchKernelGGL((filter_kernel<false, true>), grid, block, 0, st, a->p, fp, packed, lens, nreads, nchunks, max_len, out, wk);
        else        hipLaunchKernelGGL((filter_kernel<false, false>), grid, block, 0, st, a->p, fp, packed, lens, nreads, nchunks, max_len, out, wk);
    }
    launch_fast_scan(wk, st);
    // the same bin rule as K1's histogram (filter_kernel): row-count bins also for ragged batches when tail mode applies
    hipLaunchKernelGGL(scatter_kernel, dim3(wk.nused), dim3(256), 0, st, nreads, a->p.m,
                       (!ragged || !(a->flags & ATR_START_WITHIN_SEQ1)) ? 1 : 0, wk);        // == ragged_rows_bins(u.sr)
    // K4a and K4 work on disjoint slots of `order` and run side by side (launch_fast_dp)
    return launch_fast_dp(a, packed, lens, nreads, nchunks, max_len, out, wk, nullptr, 0, 1, st, false, false);
}

int launch_prefix_band(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                       int max_len, uint4 *out, hipStream_t st) {
    const BandParams bp = band_params(a);
    const dim3 grid((unsigned)((nreads + 255) / 256)), block(256);
    if (bp.and_mode) hipLaunchKernelGGL((prefix_band_kernel<true>), grid, block, 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out);
    else             hipLaunchKernelGGL((prefix_band_kernel<false>), grid, block, 0, st, a->p, bp, packed, lens, nreads, nchunks, max_len, out);
    return (int)hipGetLastError();
}

}  // namespace atr
