// misc_kernels.hip -- gfx950 kernels around misc_core.hpp (general MultiAligner.locate,
// compare_prefixes/compare_suffixes), one pair per thread.
#include <hip/hip_runtime.h>
#include "aligner_host.hpp"
#include "misc_core.hpp"

namespace atr {

__global__ __launch_bounds__(256) void multi_locate_kernel(const uint8_t *__restrict__ refs, long long ref_stride,
                                                           const int32_t *__restrict__ ref_lens,
                                                           const uint8_t *__restrict__ queries, long long q_stride,
                                                           const int32_t *__restrict__ q_lens, long long npairs,
                                                           double e, int flags, int min_overlap, int max_matches,
                                                           int *__restrict__ work, int16_t *__restrict__ out,
                                                           int32_t *__restrict__ counts, int out_stride) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npairs) return;
    // column scratch is interleaved across pairs (element i of pair p at work[i*npairs + p]):
    // the threads of a wave touch consecutive words
    counts[p] = multi_locate_one(refs + p * ref_stride, ref_lens[p], queries + p * q_stride, q_lens[p], e, flags,
                                 min_overlap, max_matches, work + p, npairs, out + (size_t)p * out_stride * 8,
                                 out_stride);
}

struct CompareRef { uint8_t r[1024]; uint8_t tr[256]; uint8_t tq[256]; };

__global__ __launch_bounds__(256) void compare_kernel(const CompareRef cr, int m, const uint8_t *__restrict__ queries,
                                                      long long q_stride, const int32_t *__restrict__ q_lens,
                                                      long long n, int max_len, int use_tables, int suffix,
                                                      int16_t *__restrict__ out) {
    __shared__ uint8_t s_ref[1024], s_tr[256], s_tq[256];
    for (int i = threadIdx.x; i < 1024; i += 256) s_ref[i] = cr.r[i];
    s_tr[threadIdx.x] = cr.tr[threadIdx.x];
    s_tq[threadIdx.x] = cr.tq[threadIdx.x];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    compare_one(s_ref, m, queries + p * q_stride, q_lens ? q_lens[p] : max_len, use_tables ? s_tr : nullptr,
                use_tables ? s_tq : nullptr, suffix != 0, out + p * 8);
}

// compare_packed_one for a batch in tile64 layout: one read per lane, one tile per wave.
__global__ __launch_bounds__(256) void compare_packed_kernel(const LocateParams lp, const uint32_t *__restrict__ packed,
                                                             const int32_t *__restrict__ lens, long long n, int max_len,
                                                             int suffix, int16_t *__restrict__ out) {
    __shared__ uint32_t s_nmask[16][4];
    if (threadIdx.x < 64) s_nmask[threadIdx.x >> 2][threadIdx.x & 3] = lp.nmask[threadIdx.x >> 2][threadIdx.x & 3];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long p = tile * 64 + lane;
    if (p >= n) return;
    const int nchunks = (max_len + 31) / 32;
    const uint32_t *mine = packed + ((size_t)tile * nchunks * 64 + lane) * 4;
    const int len = max(0, min(lens ? lens[p] : max_len, max_len));
    int16_t rec[8];
    compare_packed_one(s_nmask, lp.m, [mine](int w) { return mine[(size_t)(w >> 2) * 256 + (w & 3)]; }, len, suffix != 0, rec);
    *(uint4 *)(out + 8 * p) = *(const uint4 *)rec;
}

int launch_compare_packed(const LocateParams &lp, const uint32_t *packed, const int32_t *lens, long long n, int max_len,
                          int suffix, int16_t *out, hipStream_t st) {
    hipLaunchKernelGGL(compare_packed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lp, packed, lens, n,
                       max_len, suffix, out);
    return (int)hipGetLastError();
}

struct CompTable { uint8_t c[256]; };

__global__ __launch_bounds__(256) void correct_kernel(uint8_t *__restrict__ s1, uint8_t *__restrict__ q1,
                                                      const int32_t *__restrict__ l1, uint8_t *__restrict__ s2,
                                                      uint8_t *__restrict__ q2, const int32_t *__restrict__ l2,
                                                      long long stride, const int16_t *__restrict__ im, int im_stride,
                                                      int gate_records,
                                                      const uint8_t *__restrict__ mask, long long n, int max_len,
                                                      int action, int min_qual_diff, int truncate,
                                                      const CompTable ct, int32_t *__restrict__ changed,
                                                      int32_t *__restrict__ newlen) {
    __shared__ uint8_t s_comp[256];
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int len1 = l1 ? l1[p] : max_len, len2 = l2 ? l2[p] : max_len;
    // gate_records: `im` are the records of atr_insert_match_batch; a pair is corrected when its insert
    // match exists and has errors (modifiers.py:397-404)
    const int16_t *imp = im + (size_t)im_stride * p;
    if ((mask && !mask[p]) || (gate_records && (imp[1] < 0 || imp[5] <= 0))) {
        changed[2 * p] = changed[2 * p + 1] = 0;
        newlen[2 * p] = len1; newlen[2 * p + 1] = len2;
        return;
    }
    correct_errors_one(s1 + p * stride, q1 ? q1 + p * stride : nullptr, len1, s2 + p * stride,
                       q2 ? q2 + p * stride : nullptr, len2, imp, action, min_qual_diff, truncate != 0, s_comp,
                       changed + 2 * p, newlen + 2 * p);
}

// Plane-guided correction (misc_core.hpp, correct_errors_planes_one): one pair per lane.  The lane
// loads the bit planes of both reads (coalesced tile64 chunk loads), parks read 2's in LDS (the 32 bases
// that face a word of read 1 start at a per-lane bit offset) and only then touches the ASCII
// matrices, at the few positions where the reads disagree.
// PW: plane words per read the kernel is built for -- 8 (reads of up to 256 bases: 33 KB of LDS per block, four
// blocks per CU) or 10 (ATR_INSERT_MAX_READ / 32).
template <int PW>
__global__ __launch_bounds__(256) void correct_planes_kernel(const int16_t *__restrict__ records,
                                                             const uint4 *__restrict__ planes1,
                                                             const uint4 *__restrict__ planes2, int nchunks,
                                                             uint8_t *__restrict__ s1, uint8_t *__restrict__ q1,
                                                             const int32_t *__restrict__ l1, uint8_t *__restrict__ s2,
                                                             uint8_t *__restrict__ q2, const int32_t *__restrict__ l2,
                                                             long long stride, long long n, int max_len, int action,
                                                             int min_qual_diff, const CompTable ct,
                                                             int32_t *__restrict__ changed, int32_t *__restrict__ newlen) {
    __shared__ uint8_t s_comp[256];
    __shared__ uint32_t s_b2[4][4 * PW][64];        // per wave: read 2's planes, [plane * W + word][lane]
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long p = tile * 64 + lane;
    if (tile * 64 >= n) return;
    const bool live = p < n;
    const int16_t *rec = records + 24 * (live ? p : 0);
    const int len1 = live ? (l1 ? l1[p] : max_len) : 0, len2 = live ? (l2 ? l2[p] : max_len) : 0;
    // a pair is corrected when its insert match exists and has errors (modifiers.py:397-404)
    const bool todo = live && rec[1] >= 0 && rec[5] > 0;
    const int j = todo ? (int)rec[3] : 0;                         // the overlap: read1[0:j] faces revcomp(read2[0:j])
    uint32_t a[PW][4];
    const uint4 *t1 = planes1 + (size_t)tile * nchunks * 64 + lane, *t2 = planes2 + (size_t)tile * nchunks * 64 + lane;
    uint32_t (*b2)[64] = s_b2[wave];
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        if (c < nchunks) {
            const uint4 v1 = t1[(size_t)c * 64], v2 = t2[(size_t)c * 64];
            a[c][0] = v1.x; a[c][1] = v1.y; a[c][2] = v1.z; a[c][3] = v1.w;
            b2[0 * PW + c][lane] = v2.x; b2[1 * PW + c][lane] = v2.y;
            b2[2 * PW + c][lane] = v2.z; b2[3 * PW + c][lane] = v2.w;
        } else {
            a[c][0] = a[c][1] = a[c][2] = a[c][3] = 0u;
        }
    }
    uint32_t mism[PW];
    const auto b2word = [b2, lane](int plane, int idx) { return b2[plane * PW + idx][lane]; };
#pragma unroll
    for (int w = 0; w < PW; ++w) mism[w] = (todo && w < nchunks) ? facing_mismatches(a[w], b2word, nchunks, j, w) : 0u;
    if (!live) return;
    if (!todo) {
        changed[2 * p] = changed[2 * p + 1] = 0;
        newlen[2 * p] = len1; newlen[2 * p + 1] = len2;
        return;
    }
    const auto code1 = [&a](int w, int b) {            // read 1: its planes are still in registers (w is a literal after unrolling)
        return ((a[w][0] >> b) & 1u) | (((a[w][1] >> b) & 1u) << 1) | (((a[w][2] >> b) & 1u) << 2) | (((a[w][3] >> b) & 1u) << 3);
    };
    const auto code2 = [b2, lane](int pos) {           // read 2: by position out of LDS
        const int w = pos >> 5, b = pos & 31;
        return ((b2[0 * PW + w][lane] >> b) & 1u) | (((b2[1 * PW + w][lane] >> b) & 1u) << 1) |
               (((b2[2 * PW + w][lane] >> b) & 1u) << 2) | (((b2[3 * PW + w][lane] >> b) & 1u) << 3);
    };
    correct_errors_planes_one<PW>(s1 + p * stride, q1 ? q1 + p * stride : nullptr, len1, s2 + p * stride,
                                                q2 ? q2 + p * stride : nullptr, len2, j, mism, nchunks, action, min_qual_diff,
                                                s_comp, changed + 2 * p, newlen + 2 * p, code1, code2);
}

int launch_correct_planes(const int16_t *records, const uint4 *planes1, const uint4 *planes2, int nchunks, uint8_t *s1,
                          uint8_t *q1, const int32_t *l1, uint8_t *s2, uint8_t *q2, const int32_t *l2, long long stride,
                          long long n, int max_len, int action, int min_qual_diff, const uint8_t *comp, int32_t *changed,
                          int32_t *newlen, hipStream_t st) {
    CompTable ct;
    memcpy(ct.c, comp, 256);
    if (nchunks <= 8)
        hipLaunchKernelGGL(correct_planes_kernel<8>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, records, planes1,
                           planes2, nchunks, s1, q1, l1, s2, q2, l2, stride, n, max_len, action, min_qual_diff, ct, changed,
                           newlen);
    else
        hipLaunchKernelGGL(correct_planes_kernel<10>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, records, planes1,
                           planes2, nchunks, s1, q1, l1, s2, q2, l2, stride, n, max_len, action, min_qual_diff, ct, changed,
                           newlen);
    return (int)hipGetLastError();
}

// Reads whose first min(len, other_len) bases hold a byte the pack table gave no code (all four
// plane bits clear): InsertAligner.match_insert only ever looks at that common prefix of read 2.
__global__ __launch_bounds__(256) void planes_uncoded_kernel(const uint4 *__restrict__ planes, int nchunks,
                                                             const int32_t *__restrict__ lens,
                                                             const int32_t *__restrict__ other, long long n, int max_len,
                                                             int32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long p = tile * 64 + lane;
    if (tile * 64 >= n) return;
    int len = 0;
    if (p < n) len = min(lens ? lens[p] : max_len, other ? other[p] : max_len);
    len = max(0, min(len, max_len));
    const uint4 *t = planes + (size_t)tile * nchunks * 64 + lane;
    bool bad = false;
    for (int c = 0; c < nchunks; ++c) {
        const int left = len - 32 * c;
        if (__builtin_amdgcn_readfirstlane(__any(left > 0)) == 0) break;
        const uint4 v = t[(size_t)c * 64];
        const uint32_t coded = v.x | v.y | v.z | v.w;
        const uint32_t want = left >= 32 ? 0xFFFFFFFFu : left > 0 ? (1u << left) - 1u : 0u;
        bad = bad || (~coded & want) != 0u;
    }
    const unsigned long long votes = __ballot(bad);
    if (lane == 0 && votes) atomicAdd(count, (int)__popcll(votes));
}

int launch_planes_uncoded(const uint4 *planes, int nchunks, const int32_t *lens, const int32_t *other, long long n,
                          int max_len, int32_t *count, hipStream_t st) {
    hipLaunchKernelGGL(planes_uncoded_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, planes, nchunks, lens,
                       other, n, max_len, count);
    return (int)hipGetLastError();
}

int launch_correct(uint8_t *s1, uint8_t *q1, const int32_t *l1, uint8_t *s2, uint8_t *q2, const int32_t *l2,
                   long long stride, const int16_t *im, int im_stride, int gate_records, const uint8_t *mask, long long n,
                   int max_len, int action,
                   int min_qual_diff, int truncate, const uint8_t *comp, int32_t *changed, int32_t *newlen,
                   hipStream_t st) {
    CompTable ct;
    memcpy(ct.c, comp, 256);
    hipLaunchKernelGGL(correct_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s1, q1, l1, s2, q2, l2,
                       stride, im, im_stride, gate_records, mask, n, max_len, action, min_qual_diff, truncate, ct, changed,
                       newlen);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void postfilter_kernel(int16_t *__restrict__ rec, long long n, int m,
                                                         int min_overlap, double max_error_rate,
                                                         const double *__restrict__ rmp, int rmp_ld, double max_rmp,
                                                         int accept_full) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int16_t r[8];
    *(uint4 *)r = *(const uint4 *)(rec + 8 * p);
    adapter_postfilter_one(r, m, min_overlap, max_error_rate, rmp, rmp_ld, max_rmp, accept_full != 0);
    *(uint4 *)(rec + 8 * p) = *(const uint4 *)r;
}

int launch_postfilter(int16_t *rec, long long n, int m, int min_overlap, double max_error_rate, const double *rmp,
                      int rmp_ld, double max_rmp, int accept_full, hipStream_t st) {
    hipLaunchKernelGGL(postfilter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, n, m, min_overlap,
                       max_error_rate, rmp, rmp_ld, max_rmp, accept_full);
    return (int)hipGetLastError();
}

int launch_multi(const uint8_t *refs, long long ref_stride, const int32_t *ref_lens, const uint8_t *queries,
                 long long q_stride, const int32_t *q_lens, long long npairs, double e, int flags, int min_overlap,
                 int max_matches, int *work, int16_t *out, int32_t *counts, int out_stride, hipStream_t st) {
    hipLaunchKernelGGL(multi_locate_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, refs, ref_stride,
                       ref_lens, queries, q_stride, q_lens, npairs, e, flags, min_overlap, max_matches, work, out,
                       counts, out_stride);
    return (int)hipGetLastError();
}

int launch_compare(const uint8_t *ref, int m, const uint8_t *queries, long long q_stride, const int32_t *q_lens,
                   long long n, int max_len, int wildcard_ref, int wildcard_query, int suffix, int16_t *out,
                   hipStream_t st) {
    CompareRef cr;
    memset(&cr, 0, sizeof(cr));
    memcpy(cr.r, ref, (size_t)m);
    const Tables &T = tables();
    const int use_tables = (wildcard_ref || wildcard_query) ? 1 : 0;
    if (use_tables) {                                                  // _align.pyx:521-530
        memcpy(cr.tr, wildcard_ref ? T.iupac : T.acgt, 256);
        memcpy(cr.tq, wildcard_query ? T.iupac : T.acgt, 256);
    }
    hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cr, m, queries, q_stride,
                       q_lens, n, max_len, use_tables, suffix, out);
    return (int)hipGetLastError();
}

}  // namespace atr
