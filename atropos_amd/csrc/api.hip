// api.hip -- the C ABI of libatropos_hip.so (include/atropos_hip.h) and the
// ASCII -> 4-bit tile64 pack kernel.
#include <hip/hip_runtime.h>
#include <string>

#include "atropos_hip.h"
#include "aligner_host.hpp"
#include "locate_kernel.hpp"
#include "insert_host.hpp"
#include "locate_fast.hpp"
#include "wave_core.hpp"

#include "pack_fast.hpp"

namespace atr {

int launch_locate_fast(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads,
                       int nchunks, int max_len, uint4 *out, void *work, hipStream_t st);
bool piece_applies(const atr_aligner *a, int max_len, FilterParams *fp_out, PieceParams *pp_out);
int piece_ragged_len(int max_len);
int prepare_locate_planes(const atr_aligner *a, int max_len, bool ragged);
int launch_locate_ascii_fused(const atr_aligner *a, const uint8_t *ascii, long long row_stride, const int32_t *lens, long long nreads,
                              int max_len, const uint8_t table[256], uint4 *planes, uint4 *out, void *work, hipStream_t st);
int launch_locate_planes(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len, uint4 *out,
                         void *work, hipStream_t st);
int launch_prefix_band(const atr_aligner *a, const uint4 *packed, const int32_t *lens, long long nreads, int nchunks,
                       int max_len, uint4 *out, hipStream_t st);
int launch_insert(const atr_insert_aligner *a, const uint4 *p1, const int32_t *l1, const uint4 *p2,
                  const int32_t *l2, long long npairs, int nchunks, int max_len, uint4 *out, int cased, hipStream_t st);
int launch_locate_wave(const atr_aligner *a, const uint4 *packed, const uint8_t *ascii, long long stride, const int32_t *lens,
                       long long nreads, int nchunks, int max_len, uint4 *out, hipStream_t st);

int launch_correct(uint8_t *s1, uint8_t *q1, const int32_t *l1, uint8_t *s2, uint8_t *q2, const int32_t *l2,
                   long long stride, const int16_t *im, int im_stride, int gate_records, const uint8_t *mask, long long n,
                   int max_len, int action,
                   int min_qual_diff, int truncate, const uint8_t *comp, int32_t *changed, int32_t *newlen,
                   hipStream_t st);
int launch_locate_debug(const LocateParams &lp, double e, int flags, int min_overlap, int indel, const uint32_t *packed, int n,
                        void *col, int32_t *matrix, int16_t *rec, hipStream_t st);
int launch_compare_packed(const LocateParams &lp, const uint32_t *packed, const int32_t *lens, long long n, int max_len,
                          int suffix, int16_t *out, hipStream_t st);
int launch_planes_uncoded(const uint4 *planes, int nchunks, const int32_t *lens, const int32_t *other, long long n,
                          int max_len, int32_t *count, hipStream_t st);
int launch_insert_correct(const atr_insert_aligner *a, const uint4 *p1, const int32_t *l1, const uint4 *p2, const int32_t *l2,
                          long long npairs, int nchunks, int max_len, uint4 *out, uint8_t *s1, uint8_t *q1, uint8_t *s2,
                          uint8_t *q2, long long stride, int action, int min_qual_diff, const uint8_t *comp,
                          int32_t *changed, int32_t *newlen, hipStream_t st);
int launch_correct_planes(const int16_t *records, const uint4 *planes1, const uint4 *planes2, int nchunks, uint8_t *s1,
                          uint8_t *q1, const int32_t *l1, uint8_t *s2, uint8_t *q2, const int32_t *l2, long long stride,
                          long long n, int max_len, int action, int min_qual_diff, const uint8_t *comp, int32_t *changed,
                          int32_t *newlen, hipStream_t st);
int launch_postfilter(int16_t *rec, long long n, int m, int min_overlap, double max_error_rate, const double *rmp,
                      int rmp_ld, double max_rmp, int accept_full, hipStream_t st);
int launch_multi(const uint8_t *refs, long long ref_stride, const int32_t *ref_lens, const uint8_t *queries,
                 long long q_stride, const int32_t *q_lens, long long npairs, double e, int flags, int min_overlap,
                 int max_matches, int *work, int16_t *out, int32_t *counts, int out_stride, hipStream_t st);
int launch_compare(const uint8_t *ref, int m, const uint8_t *queries, long long q_stride, const int32_t *q_lens,
                   long long n, int max_len, int wildcard_ref, int wildcard_query, int suffix, int16_t *out,
                   hipStream_t st);

locate_launcher locate_group_0(int), locate_group_1(int), locate_group_2(int), locate_group_3(int),
    locate_group_4(int), locate_group_5(int), locate_group_6(int), locate_group_7(int);

static thread_local std::string g_err;

int hip_fail(hipError_t e, const char *what) {             // shared with fastq_kernels.hip
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return ATR_ERR_HIP;
}

struct PackTable { uint8_t t[256]; };

// The reference does bytes.translate(table) per read (_align.pyx:243-248, :292-297);
// here one lane translates and packs its read, 32 bases (one 16-byte chunk) at a time,
// and the wavefront stores each chunk as one contiguous 1 KiB burst.
//
// STAGED: the 64 rows of a tile are contiguous in memory (64 * row_stride bytes), so the
// wave first copies that whole region into LDS with coalesced 16-byte loads and the lanes
// then read their own rows from LDS -- instead of 64 lanes issuing byte loads 150 bytes
// apart.  Needs 64 * row_stride + 32 bytes of LDS per wave (row_stride <= PACK_STAGE_MAX).
constexpr int PACK_STAGE_MAX = 256;

template <bool STAGED, bool PLANES>
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii, long long row_stride,
                                                   const int32_t *__restrict__ lens,
                                                   const int32_t *__restrict__ starts, long long nreads,
                                                   int max_len, int nchunks, const PackTable tab,
                                                   uint4 *__restrict__ packed, int32_t *__restrict__ invalid) {
    __shared__ uint8_t s_tab[256];
    __shared__ uint32_t s_spread[PLANES ? 256 : 1];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];
    s_tab[threadIdx.x] = tab.t[threadIdx.x];
    if (PLANES) s_spread[threadIdx.x] = spread_code((uint32_t)tab.t[threadIdx.x] & 15u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long ntiles = (nreads + 63) >> 6;
    if (tile >= ntiles) return;
    const long long r = tile * 64 + lane;
    const int start = (r < nreads && starts) ? starts[r] : 0;
    const int n = (r < nreads) ? max(0, min((lens ? lens[r] : max_len) - start, max_len)) : 0;
    const uint8_t *row = ascii + (r < nreads ? r : 0) * row_stride + start;
    if (STAGED) {
        const size_t wave_bytes = (size_t)64 * row_stride + PACK_STAGE_SLACK;
        uint8_t *stage = s_stage + (size_t)wave * ((wave_bytes + 15) & ~(size_t)15);
        const uint32_t mis = pack_stage_tile(stage, ascii, row_stride, nreads, tile, lane);
        row = stage + mis + (size_t)lane * row_stride + start;
    }
    uint4 *dst = packed + (size_t)tile * nchunks * 64 + lane;
    bool zero_seen = false;
    if constexpr (STAGED) {
        // four bases per step out of the staged tile (pack_fast.hpp; round 5: the byte-by-byte loop below took 0.875 ms per
        // 10 M x 150 bp as planes, 1.03 ms per 12.5 M as codes -- now 0.61 / 0.63 ms, 3.8 TB/s of a 2.3 GB stream)
        uint8_t *stage0 = s_stage + (size_t)wave * ((((size_t)64 * row_stride + PACK_STAGE_SLACK) + 15) & ~(size_t)15);
        const uint32_t mis0 = (uint32_t)((uintptr_t)(ascii + tile * 64 * row_stride) & 15);
        const uint32_t rowoff = mis0 + (uint32_t)lane * (uint32_t)row_stride + (uint32_t)start;
        if constexpr (PLANES) pack_planes_row_fast((const uint32_t *)stage0, rowoff >> 2, 0x7fffffffu, rowoff & 3u, n, nchunks, s_tab, dst, zero_seen);
        else pack_codes_row_fast((const uint32_t *)stage0, rowoff >> 2, 0x7fffffffu, rowoff & 3u, n, nchunks, s_tab, dst, zero_seen);
        if (invalid && zero_seen) atomicAdd(invalid, 1);
        return;
    }
    for (int c = 0; c < nchunks; ++c) {
        uint4 v;
        if (PLANES) {
            uint32_t pl[4];
            pack_planes_chunk(row, c * 32, n, s_spread, zero_seen, pl);
            v = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        } else {
            v.x = pack_word(row, c * 32, n, s_tab, zero_seen);
            v.y = pack_word(row, c * 32 + 8, n, s_tab, zero_seen);
            v.z = pack_word(row, c * 32 + 16, n, s_tab, zero_seen);
            v.w = pack_word(row, c * 32 + 24, n, s_tab, zero_seen);
        }
        dst[(size_t)c * 64] = v;
    }
    if (invalid && zero_seen) atomicAdd(invalid, 1);           // one atomic per wave after hipcc's coalescing
}

}  // namespace atr

using namespace atr;

template <bool PLANES>
static int pack_launch(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens, const int32_t *d_starts,
                       int64_t nreads, int max_len, const uint8_t table[256], uint8_t *d_packed, int32_t *d_invalid,
                       void *stream) {
    if (nreads < 0 || max_len < 0 || max_len > (PLANES ? ATR_MAX_READ_LEN : ATR_MAX_LONG_READ_LEN) || !table) return ATR_ERR_INVALID;
    if (nreads == 0 || max_len == 0) return ATR_OK;
    if (!d_ascii || !d_packed) return ATR_ERR_INVALID;
    PackTable tab;
    memcpy(tab.t, table, 256);
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
    const size_t per_wave = (((size_t)64 * (row_stride > 0 ? row_stride : 0) + PACK_STAGE_SLACK) + 15) & ~(size_t)15;
    bool staged = row_stride > 0 && row_stride <= PACK_STAGE_MAX;
    if (staged && 4 * per_wave + 2048 > 64 * 1024) {
        // more dynamic LDS than a launch gets by default (row_stride 250 .. 256): ask for it once; gfx950 has 160 KB per CU
        static thread_local size_t granted[2] = {0, 0};
        if (granted[PLANES ? 1 : 0] < 4 * per_wave) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pack_kernel<true, PLANES>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * per_wave));
            if (e == hipSuccess) granted[PLANES ? 1 : 0] = 4 * per_wave;
            else { (void)hipGetLastError(); staged = false; }      // the plain kernel needs no LDS
        }
    }
    if (staged) {
        hipLaunchKernelGGL((pack_kernel<true, PLANES>), grid, block, 4 * per_wave, (hipStream_t)stream, d_ascii,
                           (long long)row_stride, d_lens, d_starts, (long long)nreads, max_len, nchunks, tab,
                           (uint4 *)d_packed, d_invalid);
    } else {
        hipLaunchKernelGGL((pack_kernel<false, PLANES>), grid, block, 0, (hipStream_t)stream, d_ascii,
                           (long long)row_stride, d_lens, d_starts, (long long)nreads, max_len, nchunks, tab,
                           (uint4 *)d_packed, d_invalid);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ATR_OK : hip_fail(e, "pack_kernel launch");
}

extern "C" {

int atr_version(void) { return 100; }

int atr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    return n;
}

const char *atr_last_error(void) { return g_err.c_str(); }

int atr_translate_table(int kind, uint8_t table[256]) {
    if (!table) return ATR_ERR_INVALID;
    const Tables &T = tables();
    switch (kind) {
        case ATR_TABLE_DNA15: memcpy(table, T.dna15, 256); return ATR_OK;
        case ATR_TABLE_ACGT:  memcpy(table, T.acgt, 256);  return ATR_OK;
        case ATR_TABLE_IUPAC: memcpy(table, T.iupac, 256); return ATR_OK;
        default: return ATR_ERR_INVALID;
    }
}

int atr_case_sensitive_table(uint8_t table[256]) {
    if (!table) return ATR_ERR_INVALID;
    case_sensitive_table(tables().dna15, table);
    return ATR_OK;
}

size_t atr_packed_bytes(int64_t nreads, int max_len) { return packed_bytes(nreads, max_len); }

int atr_pack_reads(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens, const int32_t *d_starts,
                   int64_t nreads, int max_len, const uint8_t table[256], uint8_t *d_packed, int32_t *d_invalid,
                   void *stream) {
    return pack_launch<false>(d_ascii, row_stride, d_lens, d_starts, nreads, max_len, table, d_packed, d_invalid, stream);
}

int atr_pack_planes(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens, const int32_t *d_starts,
                    int64_t nreads, int max_len, const uint8_t table[256], uint8_t *d_packed, int32_t *d_invalid,
                    void *stream) {
    return pack_launch<true>(d_ascii, row_stride, d_lens, d_starts, nreads, max_len, table, d_packed, d_invalid, stream);
}

int atr_aligner_create(const char *ref, int m, double max_error_rate, int flags, int wildcard_ref,
                       int wildcard_query, int min_overlap, int indel_cost, atr_aligner **out) {
    return aligner_create(ref, m, max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap,
                          indel_cost, out);
}

void atr_aligner_destroy(atr_aligner *a) { delete a; }

int atr_aligner_set_min_overlap(atr_aligner *a, int min_overlap) { return aligner_set_min_overlap(a, min_overlap); }

int atr_aligner_set_indel_cost(atr_aligner *a, int indel_cost) { return aligner_set_indel_cost(a, indel_cost); }

int atr_aligner_query_table(const atr_aligner *a, uint8_t table[256]) {
    if (!a) return ATR_ERR_INVALID;
    if (table) memcpy(table, a->qtable, 256);
    return a->table_kind;
}

size_t atr_locate_work_bytes(int64_t nreads) { return nreads < 0 ? 0 : fast_work_bytes(nreads); }

int atr_locate_work_unresolved(const void *d_work, int64_t nreads, int n_adapters, void *stream, int64_t *out) {
    if (!d_work || !out || nreads < 0 || n_adapters < 1 || n_adapters > LINKED_MAX) return ATR_ERR_INVALID;
    const FastWork wk = fast_carve(const_cast<void *>(d_work), nreads, n_adapters * FILTER_BINS);
    uint32_t total = 0;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpy(&total, wk.total, sizeof(total), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "atr_locate_work_unresolved");
    *out = (int64_t)total;
    return ATR_OK;
}

int atr_locate_batch(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads,
                     int max_len, atr_result *d_out, void *d_work, void *stream) {
    return atr_locate_batch_path(a, d_packed, d_lens, nreads, max_len, d_out, d_work, ATR_LOCATE_AUTO, stream);
}

int atr_locate_batch_path(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads,
                          int max_len, atr_result *d_out, void *d_work, int path, void *stream) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_LONG_READ_LEN) return ATR_ERR_INVALID;
    if (path < ATR_LOCATE_AUTO || path > ATR_LOCATE_WAVE) return ATR_ERR_INVALID;
    if (path == ATR_LOCATE_WAVE && (a->p.m > WAVE_MAX_M || max_len > ATR_MAX_READ_LEN)) return ATR_ERR_UNSUPPORTED;
    if (nreads == 0) return ATR_OK;
    if (!d_out || (max_len > 0 && !d_packed)) return ATR_ERR_INVALID;
    if (max_len > ATR_MAX_READ_LEN) {
        // long reads: the full column sweep with a rolling origin base, whatever the path asked for
        if (a->p.m + a->p.k > LONG_MAX_SPAN) return ATR_ERR_UNSUPPORTED;
        path = ATR_LOCATE_FULL;
    }
    if (path == ATR_LOCATE_FULL) d_work = nullptr;
    const bool band = d_work && max_len > 0 && prefix_band_applies(a->flags, a->p.m, a->p.k);
    if (path == ATR_LOCATE_WAVE || (path == ATR_LOCATE_AUTO && !band && wave_applies(a->p.m, nreads))) {
        // short batch: a wavefront per read (wave_core.hpp)
        if (max_len == 0) d_packed = (const uint8_t *)d_out;          // (nothing is read; non-null = "packed input")
        const int rc = launch_locate_wave(a, (const uint4 *)d_packed, nullptr, 0, d_lens, nreads, (max_len + 31) / 32, max_len,
                                          (uint4 *)d_out, (hipStream_t)stream);
        return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "locate_wave_kernel launch");
    }
    if (band) {
        // anchored 5' adapter: banded DP over the 2k + 1 diagonals around the main one
        const int rc = launch_prefix_band(a, (const uint4 *)d_packed, d_lens, nreads, (max_len + 31) / 32, max_len,
                                          (uint4 *)d_out, (hipStream_t)stream);
        return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "prefix band launch");
    }
    if (d_work && a->filterable && max_len > 0) {
        const int rc = launch_locate_fast(a, (const uint4 *)d_packed, d_lens, nreads, (max_len + 31) / 32, max_len,
                                          (uint4 *)d_out, d_work, (hipStream_t)stream);
        return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "filtered locate launch");
    }
    typedef locate_launcher (*group_fn)(int);
    static const group_fn groups[LOCATE_GROUPS] = {
        locate_group_0, locate_group_1, locate_group_2, locate_group_3,
        locate_group_4, locate_group_5, locate_group_6, locate_group_7};
    const int idx = round_up_rows(a->p.m) / ROW_GRAN - 1;
    const locate_launcher fn = groups[idx / LOCATE_PER_GROUP](idx % LOCATE_PER_GROUP);
    const int nchunks = (max_len + 31) / 32;
    const int rc = fn(a, (const uint4 *)d_packed, d_lens, nreads, nchunks, max_len, (uint4 *)d_out,
                      (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "locate_kernel launch");
}

int atr_locate_planes_applies(const atr_aligner *a, int max_len, int ragged) {
    if (!a || max_len < 1 || max_len > ATR_MAX_READ_LEN) return 0;
    return piece_applies(a, ragged ? piece_ragged_len(max_len) : max_len, nullptr, nullptr) ? 1 : 0;
}

int atr_aligner_prepare(const atr_aligner *a, int max_len, int ragged) {
    if (!a || max_len < 1 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    return prepare_locate_planes(a, max_len, ragged != 0) ? ATR_OK : ATR_ERR_UNSUPPORTED;
}

int atr_locate_planes_batch(const atr_aligner *a, const uint8_t *d_planes, const int32_t *d_lens, int64_t nreads, int max_len,
                            atr_result *d_out, void *d_work, void *stream) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_out || !d_planes || !d_work) return ATR_ERR_INVALID;
    if (!piece_applies(a, d_lens ? piece_ragged_len(max_len) : max_len, nullptr, nullptr)) return ATR_ERR_UNSUPPORTED;
    const int rc = launch_locate_planes(a, (const uint4 *)d_planes, d_lens, nreads, max_len, (uint4 *)d_out, d_work, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "two-pass locate launch");
}

int atr_locate_ascii_planes_batch(const atr_aligner *a, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                                  int64_t nreads, int max_len, uint8_t *d_planes, atr_result *d_out, void *d_work, void *stream) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || row_stride < max_len) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_out || !d_ascii || !d_planes || !d_work) return ATR_ERR_INVALID;
    if (!piece_applies(a, d_lens ? piece_ragged_len(max_len) : max_len, nullptr, nullptr)) return ATR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_locate_ascii_fused(a, d_ascii, row_stride, d_lens, nreads, max_len, a->qtable, (uint4 *)d_planes, (uint4 *)d_out,
                                       d_work, st);
    if (rc == 1) {   // no fused kernel for this aligner / stride on this box: the two-kernel form, the same records
        rc = pack_launch<true>(d_ascii, row_stride, d_lens, nullptr, nreads, max_len, a->qtable, d_planes, nullptr, stream);
        if (rc != ATR_OK) return rc;
        rc = launch_locate_planes(a, (const uint4 *)d_planes, d_lens, nreads, max_len, (uint4 *)d_out, d_work, st);
    }
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "fused ASCII locate launch");
}

int atr_locate_ascii_batch(const atr_aligner *a, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                           int64_t nreads, int max_len, atr_result *d_out, void *stream) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || row_stride < max_len) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_out || !d_ascii) return ATR_ERR_INVALID;
    if (nreads > WAVE_MAX_READS || (row_stride & 3) || ((uintptr_t)d_ascii & 3)) return ATR_ERR_UNSUPPORTED;
    const int rc = launch_locate_wave(a, nullptr, d_ascii, row_stride, d_lens, nreads, 0, max_len, (uint4 *)d_out,
                                      (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "locate_wave_kernel launch");
}

// Page-locked, device-visible staging of the one-object calls (atr_locate_one, atr_multi_locate_one, atr_compare_one):
// the object goes in, the records come out; per host thread, kept for its life, grown on demand.
struct OneShot {
    uint8_t *base = nullptr, *dev = nullptr;
    size_t cap = 0;
    int device = -1;
    bool ready(size_t bytes) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return false;
        if (base && cur == device && bytes <= cap) return true;
        if (base) { (void)hipHostFree(base); base = nullptr; cap = 0; }
        const size_t want = bytes < 4096 ? 4096 : (bytes + 4095) / 4096 * 4096;
        void *p = nullptr, *dp = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return false;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) { (void)hipHostFree(p); return false; }
        base = (uint8_t *)p; dev = (uint8_t *)dp; cap = want; device = cur;
        return true;
    }
};
static OneShot &one_shot() {
    static thread_local OneShot shot;
    return shot;
}

int atr_locate_one(const atr_aligner *a, const char *query, int n, atr_result *out, void *stream) {
    if (!a || n < 0 || (n > 0 && !query) || !out) return ATR_ERR_INVALID;
    if (n > ATR_MAX_READ_LEN) return ATR_ERR_UNSUPPORTED;
    static_assert(ATR_MAX_READ_LEN <= 1024, "staging layout of atr_locate_one");
    OneShot &shot = one_shot();
    if (!shot.ready(1024 + 64)) return hip_fail(hipErrorOutOfMemory, "hipHostMalloc(one-object staging)");
    memcpy(shot.base, query, (size_t)n);
    const int rc = launch_locate_wave(a, nullptr, shot.dev, 1024, nullptr, 1, 0, n, (uint4 *)(shot.dev + 1024), (hipStream_t)stream);
    if (rc != 0) return hip_fail((hipError_t)rc, "locate_wave_kernel launch");
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    *out = *(const atr_result *)(shot.base + 1024);
    return ATR_OK;
}

int atr_multi_locate_one(const char *ref, int m, const char *query, int n, double max_error_rate, int flags, int min_overlap,
                         int max_matches, atr_result *out, int cap, int32_t *count, void *stream) {
    if (m < 0 || n < 0 || (m > 0 && !ref) || (n > 0 && !query) || flags < 0 || flags > 15 || max_matches < 1 || cap < 1 || !out ||
        !count)
        return ATR_ERR_INVALID;
    if (m > 20000 || n > 32000) return ATR_ERR_UNSUPPORTED;           // int16 coordinates
    // layout: [lens: 2 x int32][count: int32, pad][records: cap x 16][reference][query]
    const size_t rec_off = 16, ref_off = rec_off + (size_t)cap * 16, qry_off = ref_off + (((size_t)m + 15) & ~(size_t)15);
    OneShot &shot = one_shot();
    if (!shot.ready(qry_off + (size_t)n + 16)) return hip_fail(hipErrorOutOfMemory, "hipHostMalloc(one-object staging)");
    int32_t *head = (int32_t *)shot.base;
    head[0] = m; head[1] = n; head[2] = 0;
    memcpy(shot.base + ref_off, ref, (size_t)m);
    memcpy(shot.base + qry_off, query, (size_t)n);
    const int rc = launch_multi(shot.dev + ref_off, 0, (const int32_t *)shot.dev, shot.dev + qry_off, 0,
                                (const int32_t *)shot.dev + 1, 1, max_error_rate, flags, min_overlap, max_matches, nullptr,
                                (int16_t *)(shot.dev + rec_off), (int32_t *)shot.dev + 2, cap, (hipStream_t)stream);
    if (rc != 0) return hip_fail((hipError_t)rc, "multi_wave_kernel launch");
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    *count = head[2];
    const int stored = head[2] < cap ? head[2] : cap;
    memcpy(out, shot.base + rec_off, (size_t)(stored > 0 ? stored : 0) * 16);
    return ATR_OK;
}

int atr_compare_one(const char *ref, int m, const char *query, int n, int wildcard_ref, int wildcard_query, int suffix,
                    atr_result *out, void *stream) {
    if (m < 0 || n < 0 || (m > 0 && !ref) || (n > 0 && !query) || !out) return ATR_ERR_INVALID;
    if (m > 1024 || n > 32000) return ATR_ERR_UNSUPPORTED;
    OneShot &shot = one_shot();
    if (!shot.ready(32 + (size_t)n + 16)) return hip_fail(hipErrorOutOfMemory, "hipHostMalloc(one-object staging)");
    *(int32_t *)(shot.base + 16) = n;
    memcpy(shot.base + 32, query, (size_t)n);
    const int rc = launch_compare((const uint8_t *)(m ? ref : ""), m, shot.dev + 32, 0, (const int32_t *)(shot.dev + 16), 1, n,
                                  wildcard_ref, wildcard_query, suffix, (int16_t *)shot.dev, (hipStream_t)stream);
    if (rc != 0) return hip_fail((hipError_t)rc, "compare_kernel launch");
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    *out = *(const atr_result *)shot.base;
    return ATR_OK;
}

// InsertAligner.match_insert(seq1, seq2) for ONE pair in host memory (align/__init__.py:250-377): both reads -- upper-case
// IUPAC letters only, the caller checks -- go through the staging buffer, are packed to bit planes and matched; three
// launches, one synchronisation.  out: the three records of atr_insert_match_batch.
struct InsertShot {
    uint8_t *packed = nullptr;                // two plane64 batches of one pair (device memory)
    size_t half = 0;
    int device = -1;
    bool ready() {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return false;
        if (packed && cur == device) return true;
        if (packed) { (void)hipFree(packed); packed = nullptr; }
        half = atr_packed_bytes(1, ATR_INSERT_MAX_READ);
        if (hipMalloc((void **)&packed, 2 * half) != hipSuccess) { packed = nullptr; return false; }
        device = cur;
        return true;
    }
};

int atr_insert_match_one(const atr_insert_aligner *a, const char *seq1, int n1, const char *seq2, int n2, atr_result *out,
                         void *stream) {
    if (!a || n1 < 0 || n2 < 0 || (n1 > 0 && !seq1) || (n2 > 0 && !seq2) || !out) return ATR_ERR_INVALID;
    if (n1 > ATR_INSERT_MAX_READ || n2 > ATR_INSERT_MAX_READ) return ATR_ERR_UNSUPPORTED;
    static_assert(ATR_INSERT_MAX_READ <= 512, "staging layout of atr_insert_match_one");
    OneShot &shot = one_shot();
    static thread_local InsertShot dev;
    if (!shot.ready(64 + 2 * 512) || !dev.ready()) return hip_fail(hipErrorOutOfMemory, "staging of atr_insert_match_one");
    // layout: [records: 48 B][lens: 2 x int32, pad to 64][read 1: 512][read 2: 512]
    int32_t *lens = (int32_t *)(shot.base + 48);
    lens[0] = n1; lens[1] = n2;
    memcpy(shot.base + 64, seq1, (size_t)n1);
    memcpy(shot.base + 64 + 512, seq2, (size_t)n2);
    const int max_len = n1 > n2 ? n1 : n2;
    const Tables &T = tables();
    int rc = ATR_OK;
    if (max_len > 0) {
        rc = pack_launch<true>(shot.dev + 64, 512, (const int32_t *)(shot.dev + 48), nullptr, 1, max_len, T.dna15, dev.packed, nullptr, stream);
        if (rc == ATR_OK)
            rc = pack_launch<true>(shot.dev + 64 + 512, 512, (const int32_t *)(shot.dev + 52), nullptr, 1, max_len, T.dna15, dev.packed + dev.half,
                                   nullptr, stream);
        if (rc != ATR_OK) return rc;
    }
    rc = atr_insert_match_batch(a, dev.packed, (const int32_t *)(shot.dev + 48), dev.packed + dev.half, (const int32_t *)(shot.dev + 52), 1,
                                max_len, (atr_result *)shot.dev, stream);
    if (rc != ATR_OK) return rc;
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    memcpy(out, shot.base, 48);
    return ATR_OK;
}

int atr_insert_aligner_create(const atr_insert_config *cfg, atr_insert_aligner **out) {
    if (!out) return ATR_ERR_INVALID;
    *out = nullptr;
    atr_insert_aligner *h = new (std::nothrow) atr_insert_aligner();
    if (!h) return ATR_ERR_NOMEM;
    h->d_tables = nullptr;
    int rc = insert_fill(h, cfg);
    if (rc != ATR_OK) { delete h; return rc; }
    const size_t bytes = h->rmp_insert.size() * sizeof(double);
    hipError_t e = hipMalloc(&h->d_tables, 2 * bytes);
    if (e != hipSuccess) { delete h; return e == hipErrorOutOfMemory ? ATR_ERR_NOMEM : hip_fail(e, "hipMalloc(rmp tables)"); }
    e = hipMemcpy(h->d_tables, h->rmp_insert.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy((char *)h->d_tables + bytes, h->rmp_adapter.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->d_tables); delete h; return hip_fail(e, "hipMemcpy(rmp tables)"); }
    h->p.rmp_insert = (const double *)h->d_tables;
    h->p.rmp_adapter = (const double *)((char *)h->d_tables + bytes);
    *out = h;
    return ATR_OK;
}

void atr_insert_aligner_destroy(atr_insert_aligner *a) {
    if (!a) return;
    if (a->d_tables) (void)hipFree(a->d_tables);
    delete a;
}

int atr_insert_match_batch(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                           const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                           atr_result *d_out, void *stream) {
    return atr_insert_match_batch_coded(a, d_packed1, d_lens1, d_packed2, d_lens2, npairs, max_len, ATR_READ_CODES_DNA15,
                                        d_out, stream);
}

int atr_insert_match_batch_coded(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                                 const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                                 int read_codes, atr_result *d_out, void *stream) {
    if (!a || npairs < 0 || max_len < 0) return ATR_ERR_INVALID;
    if (read_codes != ATR_READ_CODES_DNA15 && read_codes != ATR_READ_CODES_CASED) return ATR_ERR_INVALID;
    if (read_codes == ATR_READ_CODES_CASED && !a->cased_ok) return ATR_ERR_UNSUPPORTED;
    if (max_len > ATR_INSERT_MAX_READ) return ATR_ERR_UNSUPPORTED;
    if (npairs == 0) return ATR_OK;
    if (!d_out || (max_len > 0 && (!d_packed1 || !d_packed2))) return ATR_ERR_INVALID;
    const int nchunks = (max_len + 31) / 32;
    const int rc = launch_insert(a, (const uint4 *)d_packed1, d_lens1, (const uint4 *)d_packed2, d_lens2,
                                 npairs, nchunks, max_len, (uint4 *)d_out, read_codes == ATR_READ_CODES_CASED ? 1 : 0,
                                 (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "insert_kernel launch");
}

size_t atr_multi_locate_work_bytes(int64_t npairs, int max_ref_len) {
    if (npairs < 0 || max_ref_len < 0) return 0;
    return (size_t)npairs * 3 * ((size_t)max_ref_len + 1) * sizeof(int32_t);
}

int atr_multi_locate_batch(const uint8_t *d_refs, int64_t ref_stride, const int32_t *d_ref_lens,
                           const uint8_t *d_queries, int64_t query_stride, const int32_t *d_query_lens,
                           int64_t npairs, double max_error_rate, int flags, int min_overlap, int max_matches,
                           int max_ref_len, void *d_work, atr_result *d_out, int32_t *d_counts, int out_stride,
                           void *stream) {
    if (npairs < 0 || flags < 0 || flags > 15 || max_matches < 1 || out_stride < 1 || max_ref_len < 0)
        return ATR_ERR_INVALID;
    if (max_ref_len > 20000) return ATR_ERR_UNSUPPORTED;           // int16 coordinates / overhang costs
    if (npairs == 0) return ATR_OK;
    if (!d_refs || !d_ref_lens || !d_queries || !d_query_lens || !d_work || !d_out || !d_counts) return ATR_ERR_INVALID;
    const int rc = launch_multi(d_refs, ref_stride, d_ref_lens, d_queries, query_stride, d_query_lens, npairs,
                                max_error_rate, flags, min_overlap, max_matches, (int *)d_work, (int16_t *)d_out,
                                d_counts, out_stride, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "multi_locate_kernel launch");
}

int atr_compare_batch(const char *ref, int m, const uint8_t *d_queries, int64_t query_stride,
                      const int32_t *d_lens, int64_t n, int max_len, int wildcard_ref, int wildcard_query,
                      int suffix, atr_result *d_out, void *stream) {
    if (!ref || m < 0 || n < 0 || max_len < 0) return ATR_ERR_INVALID;
    if (m > 1024 || max_len > 32000) return ATR_ERR_UNSUPPORTED;
    if (n == 0) return ATR_OK;
    if (!d_out || (!d_queries && max_len > 0)) return ATR_ERR_INVALID;
    const int rc = launch_compare((const uint8_t *)ref, m, d_queries, query_stride, d_lens, n, max_len, wildcard_ref,
                                  wildcard_query, suffix, (int16_t *)d_out, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "compare_kernel launch");
}

size_t atr_locate_debug_bytes(const atr_aligner *a, int n) {
    if (!a || n < 0) return 0;
    return ((size_t)(a->p.m + 1) * (size_t)(n + 1)) * 4 + (size_t)(a->p.m + 1) * 12;
}

int atr_locate_debug(const atr_aligner *a, const uint8_t *d_packed, int n, void *d_matrix, atr_result *d_out, void *stream) {
    if (!a || n < 0 || n > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (!d_matrix || !d_out || (n > 0 && !d_packed)) return ATR_ERR_INVALID;
    int32_t *matrix = (int32_t *)d_matrix;
    void *col = (void *)(matrix + (size_t)(a->p.m + 1) * (size_t)(n + 1));
    const int rc = launch_locate_debug(a->p, a->max_error_rate, a->flags, a->min_overlap, a->indel_cost,
                                       (const uint32_t *)d_packed, n, col, matrix, (int16_t *)d_out, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "locate_debug_kernel launch");
}

int atr_compare_packed(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads, int max_len,
                       int suffix, atr_result *d_out, void *stream) {
    if (!a || nreads < 0 || max_len < 0) return ATR_ERR_INVALID;
    if (max_len > ATR_MAX_READ_LEN) return ATR_ERR_UNSUPPORTED;
    if (nreads == 0) return ATR_OK;
    if (!d_out || (!d_packed && max_len > 0)) return ATR_ERR_INVALID;
    const int rc = launch_compare_packed(a->p, (const uint32_t *)d_packed, d_lens, nreads, max_len, suffix,
                                         (int16_t *)d_out, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "compare_packed_kernel launch");
}

int atr_adapter_postfilter(atr_result *d_records, int64_t n, int adapter_len, int min_overlap,
                           double max_error_rate, const double *d_rmp, int rmp_ld, double max_rmp, int accept_full,
                           void *stream) {
    if (n < 0 || adapter_len < 1 || (d_rmp && rmp_ld < 1)) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_records) return ATR_ERR_INVALID;
    const int rc = launch_postfilter((int16_t *)d_records, n, adapter_len, min_overlap, max_error_rate, d_rmp, rmp_ld,
                                     max_rmp, accept_full, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "postfilter_kernel launch");
}

int atr_correct_errors_batch(uint8_t *d_seq1, uint8_t *d_qual1, const int32_t *d_lens1, uint8_t *d_seq2,
                             uint8_t *d_qual2, const int32_t *d_lens2, int64_t stride, const int16_t *d_insert,
                             const uint8_t *d_mask, int64_t n, int max_len, int action, int min_qual_difference,
                             int truncate_seqs, const uint8_t comp[256], int32_t *d_changed, int32_t *d_newlen,
                             void *stream) {
    if (n < 0 || max_len < 0 || action < 0 || action > 2 || !comp) return ATR_ERR_INVALID;
    if ((d_qual1 == nullptr) != (d_qual2 == nullptr)) return ATR_ERR_INVALID;
    if (action != ATR_CORRECT_N && !d_qual1) return ATR_ERR_INVALID;      /* modifiers.py:245-248 */
    if (n == 0) return ATR_OK;
    if (!d_seq1 || !d_seq2 || !d_insert || !d_changed || !d_newlen) return ATR_ERR_INVALID;
    const int rc = launch_correct(d_seq1, d_qual1, d_lens1, d_seq2, d_qual2, d_lens2, stride, d_insert, 4, 0, d_mask, n,
                                  max_len, action, min_qual_difference, truncate_seqs, comp, d_changed, d_newlen,
                                  (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "correct_kernel launch");
}

int atr_planes_count_uncoded(const uint8_t *d_planes, const int32_t *d_lens, const int32_t *d_other_lens, int64_t nreads,
                             int max_len, int32_t *d_count, void *stream) {
    if (nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0 || max_len == 0) return ATR_OK;
    if (!d_planes || !d_count) return ATR_ERR_INVALID;
    const int rc = launch_planes_uncoded((const uint4 *)d_planes, (max_len + 31) / 32, d_lens, d_other_lens, nreads, max_len,
                                         d_count, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "planes_uncoded_kernel launch");
}

int atr_insert_correct_batch(const atr_result *d_insert_records, const uint8_t *d_planes1, const uint8_t *d_planes2,
                             int planes_max_len, uint8_t *d_seq1, uint8_t *d_qual1,
                             const int32_t *d_lens1, uint8_t *d_seq2, uint8_t *d_qual2, const int32_t *d_lens2,
                             int64_t stride, int64_t n, int max_len, int action, int min_qual_difference,
                             const uint8_t comp[256], int32_t *d_changed, int32_t *d_newlen, void *stream) {
    if (n < 0 || max_len < 0 || action < 0 || action > 2 || !comp) return ATR_ERR_INVALID;
    if ((d_planes1 == nullptr) != (d_planes2 == nullptr)) return ATR_ERR_INVALID;
    if ((d_qual1 == nullptr) != (d_qual2 == nullptr)) return ATR_ERR_INVALID;
    if (action != ATR_CORRECT_N && !d_qual1) return ATR_ERR_INVALID;      /* modifiers.py:245-248 */
    if (n == 0) return ATR_OK;
    if (!d_insert_records || !d_seq1 || !d_seq2 || !d_changed || !d_newlen) return ATR_ERR_INVALID;
    if (d_planes1 && planes_max_len > 0 && planes_max_len <= ATR_INSERT_MAX_READ && stride < ((int64_t)1 << 26)) {
        // the reads as atr_insert_match_batch saw them: only the positions where they disagree are visited
        const int rcp = launch_correct_planes((const int16_t *)d_insert_records, (const uint4 *)d_planes1,
                                              (const uint4 *)d_planes2, (planes_max_len + 31) / 32, d_seq1, d_qual1, d_lens1,
                                              d_seq2, d_qual2, d_lens2, stride, n, max_len, action, min_qual_difference,
                                              comp, d_changed, d_newlen, (hipStream_t)stream);
        return rcp == 0 ? ATR_OK : hip_fail((hipError_t)rcp, "correct_planes_kernel launch");
    }
    const int rc = launch_correct(d_seq1, d_qual1, d_lens1, d_seq2, d_qual2, d_lens2, stride,
                                  (const int16_t *)d_insert_records, 24, 1, nullptr, n, max_len, action,
                                  min_qual_difference, 1, comp, d_changed, d_newlen, (hipStream_t)stream);
    return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "correct_kernel launch");
}

int atr_insert_match_correct_batch(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                                   const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                                   atr_result *d_out, uint8_t *d_seq1, uint8_t *d_qual1, uint8_t *d_seq2,
                                   uint8_t *d_qual2, int64_t stride, int action, int min_qual_difference,
                                   const uint8_t comp[256], int32_t *d_changed, int32_t *d_newlen, void *stream) {
    if (!a || npairs < 0 || max_len < 0 || action < 0 || action > 2 || !comp) return ATR_ERR_INVALID;
    if ((d_qual1 == nullptr) != (d_qual2 == nullptr)) return ATR_ERR_INVALID;
    if (action != ATR_CORRECT_N && !d_qual1) return ATR_ERR_INVALID;      /* modifiers.py:245-248 */
    if (max_len > ATR_INSERT_MAX_READ) return ATR_ERR_UNSUPPORTED;
    if (npairs == 0) return ATR_OK;
    if (!d_out || !d_seq1 || !d_seq2 || !d_changed || !d_newlen || (max_len > 0 && (!d_packed1 || !d_packed2)))
        return ATR_ERR_INVALID;
    const int nchunks = (max_len + 31) / 32;
    // (the wave-level correction addresses a tile's 64 rows with 32-bit offsets: a row pitch of 64 MB or more takes the two calls)
    const int rc = stride >= ((int64_t)1 << 26) ? -1 : launch_insert_correct(a, (const uint4 *)d_packed1, d_lens1, (const uint4 *)d_packed2, d_lens2, npairs,
                                         nchunks, max_len, (uint4 *)d_out, d_seq1, d_qual1, d_seq2, d_qual2, stride, action,
                                         min_qual_difference, comp, d_changed, d_newlen, (hipStream_t)stream);
    if (rc >= 0) return rc == 0 ? ATR_OK : hip_fail((hipError_t)rc, "insert_correct_kernel launch");
    // not a batch for the fused kernel: the two steps one after the other
    const int r1 = atr_insert_match_batch(a, d_packed1, d_lens1, d_packed2, d_lens2, npairs, max_len, d_out, stream);
    if (r1 != ATR_OK) return r1;
    return atr_insert_correct_batch(d_out, max_len > 0 ? d_packed1 : nullptr, max_len > 0 ? d_packed2 : nullptr, max_len, d_seq1,
                                    d_qual1, d_lens1, d_seq2, d_qual2, d_lens2, stride, npairs, max_len, action,
                                    min_qual_difference, comp, d_changed, d_newlen, stream);
}

}  // extern "C"
