// pairs_long.hip -- atr_locate_pairs_long_batch: Aligner.locate for pairs beyond ATR_PAIRS_MAX_LEN
// (pairs_long_core.hpp: 64-bit cells, the DP column in the caller's workspace).  One pair per lane.
#include <hip/hip_runtime.h>

#include "atropos_hip.h"
#include "pairs_long_core.hpp"

namespace atr {

int hip_fail(hipError_t e, const char *what);

template <bool AND_MODE>
__global__ __launch_bounds__(64) void pairs_long_kernel(const PairLongParams p, const uint4 *__restrict__ ref_packed,
                                                        const int32_t *__restrict__ ref_lens, int ref_chunks, int ref_max_len,
                                                        int revcomp, const uint4 *__restrict__ query_packed,
                                                        const int32_t *__restrict__ query_lens, int query_chunks,
                                                        int query_max_len, long long npairs, uint64_t *__restrict__ col,
                                                        uint8_t *__restrict__ refc, long long stride, uint4 *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
    if (r >= npairs) return;
    const int m = ref_lens ? ref_lens[r] : ref_max_len, n = query_lens ? query_lens[r] : query_max_len;
    const uint32_t *rp = (const uint32_t *)(ref_packed + ((size_t)(r >> 6) * ref_chunks) * 64 + (r & 63));
    const uint32_t *qp = (const uint32_t *)(query_packed + ((size_t)(r >> 6) * query_chunks) * 64 + (r & 63));
    uint32_t rec[4];
    locate_pair_long<AND_MODE>(col + r, (size_t)stride, refc + r, (size_t)stride, rp, m, revcomp != 0, qp, n, p, rec);
    out[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

}  // namespace atr

using namespace atr;

extern "C" {

size_t atr_locate_pairs_long_work_bytes(int64_t npairs, int ref_max_len) {
    if (npairs < 0 || ref_max_len < 0) return 0;
    const size_t stride = (size_t)((npairs + 63) / 64 * 64);
    return stride * ((size_t)ref_max_len + 1) * 8 + stride * (size_t)(ref_max_len + 1) + 256;
}

int atr_locate_pairs_long_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len, int revcomp_ref,
                                const uint8_t *d_query_packed, const int32_t *d_query_lens, int query_max_len,
                                int64_t npairs, double max_error_rate, int flags, int wildcard_ref, int wildcard_query,
                                int min_overlap, int indel_cost, atr_result *d_out, void *d_work, void *stream) {
    if (npairs < 0 || ref_max_len < 0 || query_max_len < 0 || flags < 0 || flags > 15 || min_overlap < 1 || indel_cost < 1)
        return ATR_ERR_INVALID;
    if (ref_max_len > ATR_MAX_LONG_READ_LEN || query_max_len > ATR_MAX_LONG_READ_LEN) return ATR_ERR_UNSUPPORTED;
    if (!(max_error_rate >= 0.0) || max_error_rate * (double)ref_max_len > (double)PAIRS_LONG_MAX_K) return ATR_ERR_UNSUPPORTED;
    if (npairs == 0) return ATR_OK;
    if (!d_out || !d_work || (ref_max_len > 0 && !d_ref_packed) || (query_max_len > 0 && !d_query_packed)) return ATR_ERR_INVALID;
    PairLongParams p;
    p.e = max_error_rate; p.flags = flags; p.min_overlap = min_overlap; p.indel_cost = indel_cost;
    p.and_mode = (wildcard_ref || wildcard_query) ? 1 : 0;
    const long long stride = (npairs + 63) / 64 * 64;
    uint64_t *col = (uint64_t *)d_work;
    uint8_t *refc = (uint8_t *)(col + (size_t)stride * ((size_t)ref_max_len + 1));
    const dim3 grid((unsigned)(stride / 64)), block(64);
    const uint4 dummy_holder = make_uint4(0, 0, 0, 0);
    (void)dummy_holder;
    const uint4 *rp = (const uint4 *)(d_ref_packed ? d_ref_packed : (const uint8_t *)d_out);      // (an all-empty side reads nothing)
    const uint4 *qp = (const uint4 *)(d_query_packed ? d_query_packed : (const uint8_t *)d_out);
    if (p.and_mode)
        hipLaunchKernelGGL(pairs_long_kernel<true>, grid, block, 0, (hipStream_t)stream, p, rp, d_ref_lens, (ref_max_len + 31) / 32,
                           ref_max_len, revcomp_ref, qp, d_query_lens, (query_max_len + 31) / 32, query_max_len, (long long)npairs,
                           col, refc, stride, (uint4 *)d_out);
    else
        hipLaunchKernelGGL(pairs_long_kernel<false>, grid, block, 0, (hipStream_t)stream, p, rp, d_ref_lens, (ref_max_len + 31) / 32,
                           ref_max_len, revcomp_ref, qp, d_query_lens, (query_max_len + 31) / 32, query_max_len, (long long)npairs,
                           col, refc, stride, (uint4 *)d_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ATR_OK : hip_fail(e, "pairs_long_kernel launch");
}

}  // extern "C"
