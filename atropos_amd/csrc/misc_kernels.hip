// misc_kernels.hip -- gfx950 kernels around misc_core.hpp (general MultiAligner.locate: one pair per wavefront;
// compare_prefixes/compare_suffixes: one pair per thread).
#include <hip/hip_runtime.h>
#include "aligner_host.hpp"
#include "misc_core.hpp"
#include "correct_wave.hpp"

namespace atr {

// MultiAligner.locate, one pair per WAVEFRONT: 64 candidates at a time, one Hamming distance per lane (misc_core.hpp,
// multi_locate_diag is the same thing candidate by candidate), the reference's order restored with ballots.  The
// lane-per-pair kernel above runs the column DP as one chain of m * n dependent byte operations: 2.6 ms for ONE pair
// of 150-base reads, which is what the module swap of INTEGRATION.md section 1 pays per read pair with the insert
// aligner.  Pairs of up to MULTI_LDS bytes a side are staged in LDS.
constexpr int MULTI_LDS = 4096;

__global__ __launch_bounds__(64) void multi_wave_kernel(const uint8_t *__restrict__ refs, long long ref_stride,
                                                        const int32_t *__restrict__ ref_lens,
                                                        const uint8_t *__restrict__ queries, long long q_stride,
                                                        const int32_t *__restrict__ q_lens, long long npairs, double e,
                                                        int flags, int min_overlap, int max_matches,
                                                        int16_t *__restrict__ out, int32_t *__restrict__ counts, int cap) {
    __shared__ uint8_t s_ref[MULTI_LDS], s_qry[MULTI_LDS];
    const int lane = threadIdx.x;
    const long long p = blockIdx.x;
    const int m = __builtin_amdgcn_readfirstlane(ref_lens[p]), n = __builtin_amdgcn_readfirstlane(q_lens[p]);
    const uint8_t *R = refs + p * ref_stride, *Q = queries + p * q_stride;
    if (m <= MULTI_LDS && n <= MULTI_LDS) {
        for (int i = lane; i < m; i += 64) s_ref[i] = R[i];
        for (int i = lane; i < n; i += 64) s_qry[i] = Q[i];
        __syncthreads();
        R = s_ref; Q = s_qry;
    }
    const MultiSetup S = multi_setup(m, n, e, flags, min_overlap);
    int16_t *o = out + (size_t)p * cap * 8;
    int nh = 0, exact = -1, exact_lane = -1;
    bool broke = false;
    int16_t rec[8], exact_rec[8];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (S.eq) {
        for (int j0 = S.min_n + 1; j0 <= S.max_n && !broke; j0 += 64) {     // row m, by column (:667-745)
            const int j = j0 + lane;
            bool perfect = false;
            const bool acc = j <= S.max_n && multi_candidate(S, R, Q, m, j, rec, perfect);
            const unsigned long long A = __ballot(acc), X = __ballot(acc && perfect);
            if (A == 0ull) continue;
            // where the reference's loop ends inside this chunk: at the first perfect hit, or at the hit that fills max_matches
            unsigned long long t = A;
            for (int c = 1; c < max_matches - nh && t; ++c) t &= t - 1ull;
            const int cut = t ? (int)__builtin_ctzll(t) : 64, ex = X ? (int)__builtin_ctzll(X) : 64;
            const int brk = ex <= cut ? ex : cut;
            const int slot = nh + (int)__popcll(A & below);
            if (acc && lane <= brk) {
                const bool is_exact = lane == ex && lane == brk;
                const int at = slot < cap ? slot : (is_exact && cap > 0 ? cap - 1 : -1);
                if (at >= 0)
#pragma unroll
                    for (int w = 0; w < 8; ++w) o[8 * at + w] = rec[w];
                if (is_exact)
#pragma unroll
                    for (int w = 0; w < 8; ++w) exact_rec[w] = rec[w];
            }
            if (brk < 64) {
                const int taken = (int)__popcll(A & ((2ull << brk) - 1ull));
                if (brk == ex) { exact = nh + taken - 1; exact_lane = brk; }
                nh += taken;
                broke = true;
            } else {
                nh += (int)__popcll(A);
            }
        }
    }
    if (!broke && S.max_n == n) {                                            // the last column, by row (:746-763)
        for (int i0 = S.er ? 0 : m; i0 <= m; i0 += 64) {
            const int i = i0 + lane;
            bool perfect = false;
            const bool acc = i <= m && multi_candidate(S, R, Q, i, n, rec, perfect);
            const unsigned long long A = __ballot(acc);
            const int slot = nh + (int)__popcll(A & below);
            if (acc && slot < cap)
#pragma unroll
                for (int w = 0; w < 8; ++w) o[8 * slot + w] = rec[w];
            nh += (int)__popcll(A);
        }
    }
    if (exact >= 0) {                                                        // only the exact hit (:767-768)
        __syncthreads();                                                     // (the stores above have landed)
        if (exact != 0 && cap > 0 && lane == exact_lane)
#pragma unroll
            for (int w = 0; w < 8; ++w) o[w] = exact_rec[w];
        nh = 1;
    }
    if (lane == 0) counts[p] = nh;
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

// Aligner.enable_debug(): one pair, one lane (a debugging aid, not a throughput path)
__global__ void locate_debug_kernel(const LocateParams lp, double e, int flags, int min_overlap, int indel,
                                    const uint32_t *__restrict__ packed, int n, DebugCell *col, int32_t *matrix,
                                    int16_t *rec) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int p0 = round_up_rows_dev(lp.m) - lp.m;
    locate_debug_one(lp.m, n, e, flags, min_overlap, indel,
                     [&lp, p0](int i, uint32_t qc) { const int b = p0 + i - 1; return ((lp.nmask[qc][b >> 5] >> (b & 31)) & 1u) != 0u; },
                     [packed](int j) { return (packed[(size_t)((j - 1) >> 5) * 256 + (((j - 1) >> 3) & 3)] >> (4 * ((j - 1) & 7))) & 15u; },
                     col, matrix, rec);
}

int launch_locate_debug(const LocateParams &lp, double e, int flags, int min_overlap, int indel, const uint32_t *packed, int n,
                        void *col, int32_t *matrix, int16_t *rec, hipStream_t st) {
    hipLaunchKernelGGL(locate_debug_kernel, dim3(1), dim3(64), 0, st, lp, e, flags, min_overlap, indel, packed, n,
                       (DebugCell *)col, matrix, rec);
    return (int)hipGetLastError();
}


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
// PW: plane words per read the kernel is built for -- 8 (reads of up to 256 bases) or 10 (ATR_INSERT_MAX_READ / 32).
//
// The disagreeing positions of the wave's 64 pairs become (pair, position) TASKS in a per-wave LDS queue and are
// worked off 64 at a time, one task per lane: a task is four independent byte loads (both bases, both qualities),
// the reference's decision (correct_apply) and up to two byte stores -- one memory round trip for 64 positions,
// where a lane walking its own pair's ~8 positions one after the other paid one round trip EACH.  Positions of a
// pair touch distinct bytes, so their order does not matter; the per-pair counters are LDS atomics.  Only the
// 'liberal' pairs with positions of equal quality left (correct_ties) are finished by their own lane afterwards.
// The queue re-uses the LDS that staged read 2's planes for facing_mismatches.
template <int PW>
__global__ __launch_bounds__(256) void correct_planes_kernel(const int16_t *__restrict__ records,
                                                             const uint4 *__restrict__ planes1,
                                                             const uint4 *__restrict__ planes2, int nchunks,
                                                             uint8_t *s1, uint8_t *q1, const int32_t *__restrict__ l1,
                                                             uint8_t *s2, uint8_t *q2, const int32_t *__restrict__ l2,
                                                             long long stride, long long n, int max_len, int action,
                                                             int min_qual_diff, const CompTable ct,
                                                             int32_t *__restrict__ changed, int32_t *__restrict__ newlen) {
    // uint16 tasks that fit the plane staging area; at most 63 left over + 64 x 32 new ones are ever queued
    static_assert(4 * PW * 64 * 2 >= CORRECT_QUEUE_ENTRIES, "task queue does not fit the plane staging area");
    __shared__ uint8_t s_comp[256];
    __shared__ uint8_t s_letter[16];                              // DNA15 code -> its byte
    __shared__ uint32_t s_b2[4][4 * PW][64];                      // per wave: read 2's planes, [plane * W + word][lane]; then the queue
    __shared__ uint32_t s_cnt[4][64];                             // per pair: c1 | c2 << 10 | npend << 20
    __shared__ int32_t s_err[4][64];
    __shared__ int16_t s_j[4][64];
    __shared__ uint32_t s_tail[4], s_ptail[4];
    __shared__ unsigned long long s_acc[4][64];
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    correct_letter_table(s_letter);
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
    uint32_t mism[PW];
    {
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
        const auto b2word = [b2, lane](int plane, int idx) { return b2[plane * PW + idx][lane]; };
#pragma unroll
        for (int w = 0; w < PW; ++w) mism[w] = (todo && w < nchunks) ? facing_mismatches(a[w], b2word, nchunks, j, w) : 0u;
    }
    // ---- the queue (the planes in LDS are not needed any more; one wave, so no block barrier): correct_wave.hpp ----
    CorrectWaveLds S;
    S.queue = lds_view((uint16_t *)&s_b2[wave][0][0]);
    S.cnt = lds_view(&s_cnt[wave][0]); S.err = lds_view(&s_err[wave][0]); S.jv = lds_view(&s_j[wave][0]);
    S.tail = lds_view(&s_tail[wave]); S.ptail = lds_view(&s_ptail[wave]); S.acc = lds_view(&s_acc[wave][0]);
    S.qcap = 4 * PW * 64 * 2;
    S.comp = lds_view((const uint8_t *)s_comp); S.letter = lds_view((const uint8_t *)s_letter);
    CorrectArgs A;
    A.planes1 = planes1; A.planes2 = planes2; A.nchunks = nchunks;
    A.s1 = s1; A.q1 = q1; A.s2 = s2; A.q2 = q2; A.stride = stride;
    A.action = action; A.min_qual_diff = min_qual_diff; A.changed = changed; A.newlen = newlen;
    correct_wave_tail<PW>(S, A, tile, lane, live, todo, j, len1, len2, mism);
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
    (void)work;                                                       // (scratch of the lane-per-pair DP kernel, kept in the ABI)
    hipLaunchKernelGGL(multi_wave_kernel, dim3((unsigned)npairs), dim3(64), 0, st, refs, ref_stride, ref_lens, queries, q_stride,
                       q_lens, npairs, e, flags, min_overlap, max_matches, out, counts, out_stride);
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
