// fastq_kernels.hip -- the device-resident FASTQ batch: line index, record descriptors,
// 4-bit pack straight from the file bytes, interval-update trimmers, filters and the
// formatter (see include/atropos_hip.h, "device-resident FASTQ batch").
//
// All of this is byte streaming: every kernel is bound by HBM traffic or by the number of
// scattered byte accesses, none has arithmetic worth speaking of.  Layout decisions:
//   * the file chunk stays as it came off the disk; records are 32-byte descriptors
//     (offsets into the chunk), modifier state is two int32 per read;
//   * line ends are found with 16-byte loads + a SWAR zero-byte test, positions come from
//     a two-level exclusive scan (no atomics, input order preserved);
//   * the packer stages the contiguous byte span of a wave's 64 records in LDS with
//     coalesced 16-byte loads before the lanes pick their own sequence lines apart;
//   * the formatter copies one record per wave iteration, a byte per lane, so that both the
//     reads and the writes of a segment are one contiguous burst.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <string.h>

#include "atropos_hip.h"
#include "locate_core.hpp"
#include "fastq_core.hpp"

namespace atr {

int hip_fail(hipError_t e, const char *what);             // api.hip

// ------------------------------------------------------------------------- scans
constexpr int SCAN_BLOCK = 1024;

// inclusive scan of one value per thread over a 1024-thread block; returns the inclusive
// prefix, *total = block sum
__device__ __forceinline__ unsigned long long block_scan_1024(unsigned long long v, unsigned long long *s_part,
                                                               unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_part[wave] = x;
    __syncthreads();
    if (wave == 0) {
        unsigned long long w = lane < (SCAN_BLOCK / 64) ? s_part[lane] : 0ull;
#pragma unroll
        for (int off = 1; off < SCAN_BLOCK / 64; off <<= 1) {
            const unsigned long long y = __shfl_up(w, off, 64);
            if (lane >= off) w += y;
        }
        if (lane < SCAN_BLOCK / 64) s_part[lane] = w;
    }
    __syncthreads();
    const unsigned long long base = wave ? s_part[wave - 1] : 0ull;
    *total = s_part[SCAN_BLOCK / 64 - 1];
    return x + base;
}

// level 1: out[i] = exclusive prefix of v inside its block of 1024, sums[b] = block total
__global__ __launch_bounds__(SCAN_BLOCK) void scan_local_kernel(const uint32_t *__restrict__ v, long long n,
                                                                 long long *__restrict__ out,
                                                                 unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long s_part[SCAN_BLOCK / 64];
    const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const unsigned long long x = i < n ? v[i] : 0u;
    unsigned long long total;
    const unsigned long long inc = block_scan_1024(x, s_part, &total);
    if (i < n) out[i] = (long long)(inc - x);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// level 2: exclusive scan of the nb block totals in place (one block; a thread owns a run)
__global__ __launch_bounds__(SCAN_BLOCK) void scan_sums_kernel(unsigned long long *__restrict__ sums, long long nb,
                                                                long long *__restrict__ total_out) {
    __shared__ unsigned long long s_part[SCAN_BLOCK / 64];
    const long long per = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const long long lo = min(nb, per * (long long)threadIdx.x), hi = min(nb, lo + per);
    unsigned long long mine = 0;
    for (long long i = lo; i < hi; ++i) mine += sums[i];
    unsigned long long total;
    const unsigned long long inc = block_scan_1024(mine, s_part, &total);
    unsigned long long run = inc - mine;
    for (long long i = lo; i < hi; ++i) { const unsigned long long t = sums[i]; sums[i] = run; run += t; }
    if (threadIdx.x == 0 && total_out) *total_out = (long long)total;
}

// level 3: add the block bases; out[n] = grand total
__global__ __launch_bounds__(SCAN_BLOCK) void scan_add_kernel(long long *__restrict__ out, long long n,
                                                               const unsigned long long *__restrict__ sums,
                                                               long long nb) {
    const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < n) out[i] += (long long)sums[blockIdx.x];
}

// ------------------------------------------------------------------------- line index
constexpr int NL_THREADS = 256, NL_BLOCK_BYTES = NL_THREADS * 16;

// bit i set <=> byte i of the 16 bytes at `off` is '\n' (bytes at or beyond nbytes masked out)
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *bytes, long long off, long long nbytes) {
    if (off >= nbytes) return 0u;
    const uint4 v = *(const uint4 *)(bytes + off);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t x = w[d] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);     // 0x80 in every zero byte
        const uint32_t m = z >> 7;
        mask |= ((m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu) << (4 * d);
    }
    const long long valid = nbytes - off;
    if (valid < 16) mask &= (1u << valid) - 1u;
    return mask;
}

__global__ __launch_bounds__(NL_THREADS) void count_newlines_kernel(const uint8_t *__restrict__ bytes, long long nbytes,
                                                                     uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t s_cnt[NL_THREADS / 64];
    const long long off = ((long long)blockIdx.x * NL_THREADS + threadIdx.x) * 16;
    uint32_t c = __popc(newline_mask16(bytes, off, nbytes));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(NL_THREADS) void line_ends_kernel(const uint8_t *__restrict__ bytes, long long nbytes,
                                                                const long long *__restrict__ block_base,
                                                                uint32_t *__restrict__ line_ends) {
    __shared__ uint32_t s_cnt[NL_THREADS / 64];
    const long long off = ((long long)blockIdx.x * NL_THREADS + threadIdx.x) * 16;
    uint32_t mask = newline_mask16(bytes, off, nbytes);
    const uint32_t c = __popc(mask);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_cnt[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    long long slot = block_base[blockIdx.x] + base + (x - c);
    while (mask) {
        const int b = __ffs((int)mask) - 1;
        mask &= mask - 1;
        line_ends[slot++] = (uint32_t)(off + b);
    }
}

__global__ __launch_bounds__(256) void records_kernel(const uint8_t *__restrict__ bytes,
                                                      const uint32_t *__restrict__ line_ends, long long nrec, int strip,
                                                      FastqRecord *__restrict__ records,
                                                      unsigned long long *__restrict__ error) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    FastqRecord rec;
    const int err = fastq_record_one(bytes, line_ends, r, strip, rec);
    records[r] = rec;
    if (err) atomicMin(error, (unsigned long long)r * 8ull + (unsigned long long)err);
}

__global__ void set_i64_kernel(long long *p, long long v) { *p = v; }

// ------------------------------------------------------------------------- pack from records
struct PackTable256 { uint8_t t[256]; };
constexpr int PACKREC_WAVES = 2;
constexpr int PACKREC_STAGE = 31 * 1024;                     // bytes of LDS per wave for the staged span

__global__ __launch_bounds__(64 * PACKREC_WAVES) void pack_records_kernel(
    const uint8_t *__restrict__ bytes, const FastqRecord *__restrict__ records, const int32_t *__restrict__ begin,
    const int32_t *__restrict__ end, long long nreads, int max_len, int nchunks, const PackTable256 tab,
    uint4 *__restrict__ packed, int32_t *__restrict__ lens, int32_t *__restrict__ invalid) {
    __shared__ uint8_t s_tab[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];
    for (int i = threadIdx.x; i < 256; i += 64 * PACKREC_WAVES) s_tab[i] = tab.t[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * PACKREC_WAVES + wave;
    const long long ntiles = (nreads + 63) >> 6;
    if (tile >= ntiles) return;
    const long long r = tile * 64 + lane;
    const bool live = r < nreads;
    uint32_t soff = 0;
    int n = 0;
    if (live) {
        const FastqRecord rec = records[r];
        int a = begin ? begin[r] : 0, b = end ? end[r] : (int)rec.seq_len;
        a = max(0, min(a, (int)rec.seq_len));
        b = max(a, min(b, (int)rec.seq_len));
        soff = rec.seq_off + (uint32_t)a;
        n = min(b - a, max_len);
        if (lens) lens[r] = n;
    }
    // contiguous span of the tile's sequence slices: [first lane's start, last live lane's end)
    const int last = (int)min<long long>(63, nreads - 1 - tile * 64);
    const uint32_t span_lo = __shfl(soff, 0, 64);
    const uint32_t span_hi = __shfl(soff + (uint32_t)n, last, 64);
    // the records of a file are in increasing offset order; anything else takes the slow path
    const bool ordered = __all(!live || (soff >= span_lo && soff + (uint32_t)n <= span_hi));
    const uint8_t *row = bytes + soff;
    const uint32_t mis = span_lo & 15u;
    if (ordered && span_hi - span_lo + mis <= (uint32_t)PACKREC_STAGE - 16u) {
        uint8_t *stage = s_stage + (size_t)wave * PACKREC_STAGE;
        const uint8_t *src_al = bytes + (span_lo - mis);
        const uint32_t need = span_hi - span_lo + mis;
        for (uint32_t o = (uint32_t)lane * 16u; o < need; o += 64u * 16u) *(uint4 *)(stage + o) = *(const uint4 *)(src_al + o);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);
        row = stage + mis + (soff - span_lo);
    }
    uint4 *dst = packed + (size_t)tile * nchunks * 64 + lane;
    bool zero_seen = false;
    for (int c = 0; c < nchunks; ++c) {
        uint4 v;
        v.x = pack_word(row, c * 32, n, s_tab, zero_seen);
        v.y = pack_word(row, c * 32 + 8, n, s_tab, zero_seen);
        v.z = pack_word(row, c * 32 + 16, n, s_tab, zero_seen);
        v.w = pack_word(row, c * 32 + 24, n, s_tab, zero_seen);
        dst[(size_t)c * 64] = v;
    }
    if (invalid && zero_seen) atomicAdd(invalid, 1);
}

// ------------------------------------------------------------------------- interval updates
__device__ __forceinline__ bool load_interval(const FastqRecord *records, const int32_t *begin, const int32_t *end,
                                              long long r, FastqRecord &rec, int &a, int &b) {
    rec = records[r];
    a = begin[r];
    b = end[r];
    return b > a;
}

__global__ __launch_bounds__(256) void clip_kernel(const FastqRecord *__restrict__ records, int32_t *__restrict__ begin,
                                                   int32_t *__restrict__ end, long long n, int front, int back) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int a = begin[r], b = end[r];
    if (b <= a || (front == 0 && back == 0)) return;          // Trimmer.clip: (front or back) and len(read) > 0
    int na, nb;
    py_clip(b - a, front, back, back < 0, na, nb);            // read[front:back] or read[front:]
    begin[r] = a + na;
    end[r] = a + nb;
}

__global__ __launch_bounds__(256) void quality_trim_kernel(const uint8_t *__restrict__ bytes,
                                                           const FastqRecord *__restrict__ records,
                                                           int32_t *__restrict__ begin, int32_t *__restrict__ end,
                                                           long long n, int cutoff_front, int cutoff_back, int base,
                                                           int nextseq) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    FastqRecord rec;
    int a, b;
    if (!load_interval(records, begin, end, r, rec, a, b)) return;       // if len(read) == 0: return read
    const uint8_t *qual = bytes + rec.qual_off + a;
    if (nextseq) {
        const int stop = nextseq_trim_one(bytes + rec.seq_off + a, qual, b - a, cutoff_back, base);
        end[r] = a + stop;                                                // subseq(read, end=stop)
    } else {
        int s, e;
        quality_trim_one(qual, b - a, cutoff_front, cutoff_back, base, s, e);
        begin[r] = a + s;
        end[r] = a + e;
    }
}

__global__ __launch_bounds__(256) void nend_trim_kernel(const uint8_t *__restrict__ bytes,
                                                        const FastqRecord *__restrict__ records,
                                                        int32_t *__restrict__ begin, int32_t *__restrict__ end,
                                                        const int32_t *__restrict__ ubegin,
                                                        const int32_t *__restrict__ uend, long long n) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    FastqRecord rec;
    int a, b;
    if (!load_interval(records, begin, end, r, rec, a, b)) return;
    const int ub = ubegin ? ubegin[r] - a : 0, ue = uend ? uend[r] - a : b - a;
    int s, e;
    nend_trim_one(bytes + rec.seq_off + a, b - a, ub, ue, s, e);
    begin[r] = a + s;
    end[r] = a + (e < s ? s : e);
}

__global__ __launch_bounds__(256) void match_trim_kernel(const int16_t *__restrict__ matches,
                                                         const uint8_t *__restrict__ front, int default_front,
                                                         int32_t *__restrict__ begin, int32_t *__restrict__ end,
                                                         uint8_t *__restrict__ active, uint8_t *__restrict__ matched,
                                                         long long n) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    if (active && !active[r]) return;
    const int16_t *m = matches + 8 * r;
    if (m[1] < 0) { if (active) active[r] = 0; return; }
    const int rstart = m[2], rstop = m[3];
    int f = front ? (int)front[r] : default_front;
    if (f > 1) f = rstart == 0 ? 1 : 0;                                   // Match: front guessed from rstart == 0
    const int a = begin[r], b = end[r];
    if (f) begin[r] = min(b, a + rstop);                                  // read[match.rstop:]
    else end[r] = max(a, min(b, a + rstart));                             // read[:match.rstart]
    if (matched) matched[r] = 1;
}

__global__ __launch_bounds__(256) void read_filter_kernel(const uint8_t *__restrict__ bytes,
                                                          const FastqRecord *__restrict__ records,
                                                          const int32_t *__restrict__ begin,
                                                          const int32_t *__restrict__ end,
                                                          const int32_t *__restrict__ ubegin,
                                                          const int32_t *__restrict__ uend,
                                                          const uint8_t *__restrict__ matched, long long n, int min_len,
                                                          int max_len, double max_n, int discard_trimmed,
                                                          int discard_untrimmed, uint8_t *__restrict__ dest) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const FastqRecord rec = records[r];
    const int a = begin[r], b = max(a, end[r]);
    const int ub = ubegin ? ubegin[r] - a : 0, ue = uend ? uend[r] - a : b - a;
    dest[r] = (uint8_t)read_filter_one(bytes + rec.seq_off + a, b - a, ub, ue, matched ? matched[r] != 0 : false,
                                       min_len, max_len, max_n, discard_trimmed, discard_untrimmed);
}

// ------------------------------------------------------------------------- formatter
__global__ __launch_bounds__(256) void emit_sizes_kernel(const FastqRecord *__restrict__ records,
                                                         const int32_t *__restrict__ begin,
                                                         const int32_t *__restrict__ end,
                                                         const uint8_t *__restrict__ dest, int which, long long n,
                                                         uint32_t *__restrict__ sizes) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t s = 0;
    if (!dest || dest[r] == which) s = fastq_record_bytes(records[r], max(0, end[r] - begin[r]));
    sizes[r] = s;
}

__global__ void emit_total_kernel(const uint32_t *__restrict__ sizes, long long *__restrict__ offsets, long long n) {
    offsets[n] = n ? offsets[n - 1] + (long long)sizes[n - 1] : 0;
}

__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t len, int lane) {
    for (uint32_t o = (uint32_t)lane; o < len; o += 64u) dst[o] = src[o];
}

// One wave formats 64 consecutive records, one after the other, a byte per lane.
__global__ __launch_bounds__(256) void emit_kernel(const uint8_t *__restrict__ bytes,
                                                   const FastqRecord *__restrict__ records,
                                                   const int32_t *__restrict__ begin, const int32_t *__restrict__ end,
                                                   const int32_t *__restrict__ ubegin, const int32_t *__restrict__ uend,
                                                   const uint8_t *__restrict__ dest, int which, long long n,
                                                   const long long *__restrict__ offsets, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r0 = tile * 64;
    if (r0 >= n) return;
    const int cnt = (int)min<long long>(64, n - r0);
    for (int i = 0; i < cnt; ++i) {
        const long long r = r0 + i;                               // wave-uniform
        if (dest && dest[r] != which) continue;
        const FastqRecord rec = records[r];
        const int a = begin[r], b = max(a, end[r]);
        const uint32_t kept = (uint32_t)(b - a);
        uint8_t *o = out + offsets[r];
        if (lane == 0) o[0] = '@';
        wave_copy(o + 1, bytes + rec.name_off, rec.name_len, lane);
        o += 1 + rec.name_len;
        if (lane == 0) o[0] = '\n';
        o += 1;
        if (ubegin) {
            const int ub = ubegin[r], ue = uend[r];
            for (uint32_t k = (uint32_t)lane; k < kept; k += 64u) {
                const int pos = a + (int)k;
                o[k] = (pos >= ub && pos < ue) ? bytes[rec.seq_off + pos] : (uint8_t)'N';
            }
        } else {
            wave_copy(o, bytes + rec.seq_off + a, kept, lane);
        }
        o += kept;
        if (lane == 0) { o[0] = '\n'; o[1] = '+'; }
        o += 2;
        if (rec.flags & 1u) { wave_copy(o, bytes + rec.name_off, rec.name_len, lane); o += rec.name_len; }
        if (lane == 0) o[0] = '\n';
        o += 1;
        wave_copy(o, bytes + rec.qual_off + a, kept, lane);
        o += kept;
        if (lane == 0) o[0] = '\n';
    }
}

static long long scan_blocks(long long n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK; }

// exclusive prefix sums of v[n] -> out[n] (int64); `sums` holds scan_blocks(n) uint64
static void launch_scan(const uint32_t *v, long long n, long long *out, unsigned long long *sums, long long *total,
                        hipStream_t st) {
    const long long nb = scan_blocks(n);
    if (n == 0) { if (total) hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, total, 0ll); return; }
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, v, n, out, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, nb, total);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, out, n, sums, nb);
}

}  // namespace atr

using namespace atr;

static inline unsigned grid256(long long n) { return (unsigned)((n + 255) / 256); }
static inline int launched(const char *what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ATR_OK : hip_fail(e, what);
}
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

// work layout: [block_counts u32 x nblk][block_base i64 x (nblk + 1)][sums u64 x scan_blocks(nblk)]
size_t atr_fastq_work_bytes(int64_t nbytes) {
    if (nbytes < 0) return 0;
    const long long nblk = (nbytes + NL_BLOCK_BYTES - 1) / NL_BLOCK_BYTES;
    return align256((size_t)nblk * 4) + align256((size_t)(nblk + 1) * 8) + align256((size_t)scan_blocks(nblk) * 8) + 256;
}

int atr_fastq_count_lines(const uint8_t *d_bytes, int64_t nbytes, void *d_work, int64_t *d_nlines, void *stream) {
    if (nbytes < 0 || nbytes >= (int64_t)0xFFFFFFF0ll || !d_nlines) return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (nbytes == 0) {
        hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_nlines, 0ll);
        return launched("fastq count launch");
    }
    if (!d_bytes || !d_work || ((uintptr_t)d_bytes & 15)) return ATR_ERR_INVALID;
    const long long nblk = (nbytes + NL_BLOCK_BYTES - 1) / NL_BLOCK_BYTES;
    uint32_t *counts = (uint32_t *)d_work;
    long long *base = (long long *)((char *)d_work + align256((size_t)nblk * 4));
    unsigned long long *sums = (unsigned long long *)((char *)base + align256((size_t)(nblk + 1) * 8));
    hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)nblk), dim3(NL_THREADS), 0, st, d_bytes, (long long)nbytes, counts);
    launch_scan(counts, nblk, base, sums, (long long *)d_nlines, st);
    return launched("fastq count launch");
}

int atr_fastq_index(const uint8_t *d_bytes, int64_t nbytes, int strip, const void *d_work, uint32_t *d_line_ends,
                    int64_t nlines, atr_fastq_record *d_records, int64_t *d_error, void *stream) {
    if (nbytes < 0 || nbytes >= (int64_t)0xFFFFFFF0ll || nlines < 0 || (strip != 1 && strip != 2) || !d_error)
        return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_error, (long long)LLONG_MAX);
    if (nlines == 0 || nbytes == 0) return launched("fastq index launch");
    if (!d_bytes || !d_work || !d_line_ends || ((uintptr_t)d_bytes & 15)) return ATR_ERR_INVALID;
    const long long nblk = (nbytes + NL_BLOCK_BYTES - 1) / NL_BLOCK_BYTES;
    const long long *base = (const long long *)((const char *)d_work + align256((size_t)nblk * 4));
    hipLaunchKernelGGL(line_ends_kernel, dim3((unsigned)nblk), dim3(NL_THREADS), 0, st, d_bytes, (long long)nbytes, base, d_line_ends);
    const long long nrec = nlines / 4;
    if (nrec > 0) {
        if (!d_records) return ATR_ERR_INVALID;
        hipLaunchKernelGGL(records_kernel, dim3(grid256(nrec)), dim3(256), 0, st, d_bytes, d_line_ends, nrec, strip,
                           (FastqRecord *)d_records, (unsigned long long *)d_error);
    }
    return launched("fastq index launch");
}

int atr_pack_records(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                     const int32_t *d_end, int64_t nreads, int max_len, const uint8_t table[256],
                     uint8_t *d_packed, int32_t *d_lens, int32_t *d_invalid, void *stream) {
    if (nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || !table) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_bytes || !d_records || (max_len > 0 && !d_packed)) return ATR_ERR_INVALID;
    PackTable256 tab;
    memcpy(tab.t, table, 256);
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((ntiles + PACKREC_WAVES - 1) / PACKREC_WAVES)),
                       dim3(64 * PACKREC_WAVES), (size_t)PACKREC_WAVES * PACKREC_STAGE, (hipStream_t)stream, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, (long long)nreads, max_len, nchunks, tab,
                       (uint4 *)d_packed, d_lens, d_invalid);
    return launched("pack_records_kernel launch");
}

int atr_clip_batch(const atr_fastq_record *d_records, int32_t *d_begin, int32_t *d_end, int64_t n, int front,
                   int back, void *stream) {
    if (n < 0 || front < 0 || back > 0) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_begin || !d_end) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(clip_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, (const FastqRecord *)d_records,
                       d_begin, d_end, (long long)n, front, back);
    return launched("clip_kernel launch");
}

int atr_quality_trim_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, int32_t *d_begin,
                           int32_t *d_end, int64_t n, int cutoff_front, int cutoff_back, int base, int nextseq,
                           void *stream) {
    if (n < 0) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_bytes || !d_records || !d_begin || !d_end) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(quality_trim_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, (long long)n, cutoff_front, cutoff_back, base, nextseq);
    return launched("quality_trim_kernel launch");
}

int atr_nend_trim_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, int32_t *d_begin,
                        int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end, int64_t n,
                        void *stream) {
    if (n < 0 || ((d_unmasked_begin == nullptr) != (d_unmasked_end == nullptr))) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_bytes || !d_records || !d_begin || !d_end) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(nend_trim_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, (long long)n);
    return launched("nend_trim_kernel launch");
}

int atr_match_trim_batch(const atr_result *d_matches, const uint8_t *d_front, int default_front, int32_t *d_begin,
                         int32_t *d_end, uint8_t *d_active, uint8_t *d_matched, int64_t n, void *stream) {
    if (n < 0) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_matches || !d_begin || !d_end) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(match_trim_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, (const int16_t *)d_matches,
                       d_front, default_front, d_begin, d_end, d_active, d_matched, (long long)n);
    return launched("match_trim_kernel launch");
}

int atr_read_filter_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                          const int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end,
                          const uint8_t *d_matched, int64_t n, int min_len, int max_len, double max_n,
                          int discard_trimmed, int discard_untrimmed, uint8_t *d_dest, void *stream) {
    if (n < 0 || ((d_unmasked_begin == nullptr) != (d_unmasked_end == nullptr))) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_bytes || !d_records || !d_begin || !d_end || !d_dest) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(read_filter_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_matched,
                       (long long)n, min_len, max_len, max_n, discard_trimmed, discard_untrimmed, d_dest);
    return launched("read_filter_kernel launch");
}

// work layout: [sizes u32 x n][sums u64 x scan_blocks(n)]
size_t atr_fastq_emit_work_bytes(int64_t n) {
    if (n < 0) return 0;
    return align256((size_t)n * 4) + align256((size_t)scan_blocks(n) * 8) + 256;
}

int atr_fastq_emit(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                   const int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end,
                   const uint8_t *d_dest, int dest, int64_t n, int64_t *d_offsets, void *d_work, uint8_t *d_out,
                   void *stream) {
    if (n < 0 || !d_offsets || ((d_unmasked_begin == nullptr) != (d_unmasked_end == nullptr))) return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (!d_out) hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_offsets, 0ll);
        return launched("fastq emit launch");
    }
    if (!d_bytes || !d_records || !d_begin || !d_end || !d_work) return ATR_ERR_INVALID;
    if (!d_out) {
        uint32_t *sizes = (uint32_t *)d_work;
        unsigned long long *sums = (unsigned long long *)((char *)d_work + align256((size_t)n * 4));
        hipLaunchKernelGGL(emit_sizes_kernel, dim3(grid256(n)), dim3(256), 0, st, (const FastqRecord *)d_records, d_begin,
                           d_end, d_dest, dest, (long long)n, sizes);
        launch_scan(sizes, n, (long long *)d_offsets, sums, nullptr, st);
        hipLaunchKernelGGL(emit_total_kernel, dim3(1), dim3(1), 0, st, sizes, (long long *)d_offsets, (long long)n);
        return launched("fastq emit sizes launch");
    }
    const long long ntiles = (n + 63) / 64;
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest, dest,
                       (long long)n, (const long long *)d_offsets, d_out);
    return launched("emit_kernel launch");
}

}  // extern "C"
