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
//   * the packer copies each sequence line into an LDS row with one direct-to-LDS load
//     (global_load_lds_dword) and lets every lane re-align its row in registers (v_alignbyte_b32);
//   * the formatter stages the contiguous byte span of a wave's 64 records in LDS with
//     coalesced 16-byte loads, lets every lane assemble its record in an LDS image of the
//     output span, and writes that image with aligned 16-byte stores.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <string.h>

#include "atropos_hip.h"
#include "locate_core.hpp"
#include "fastq_core.hpp"
#include "misc_core.hpp"

#include "pack_fast.hpp"

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

// one bit per byte of a dword: 0x80 in every byte equal to c, gathered into the low 4 bits
__device__ __forceinline__ uint32_t byte_eq_mask4(uint32_t w, uint32_t crep) {
    const uint32_t x = w ^ crep;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);     // 0x80 in every zero byte
    const uint32_t m = z >> 7;
    return (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
}

// bit i set <=> byte i of the 16 bytes at `off` ends a line (universal newlines: "\n", or a
// "\r" that is not followed by "\n"); bytes at or beyond nbytes are masked out
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *bytes, long long off, long long nbytes, bool &saw_cr) {
    saw_cr = false;
    if (off >= nbytes) return 0u;
    const uint4 v = *(const uint4 *)(bytes + off);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t nl = 0, anycr = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        nl |= byte_eq_mask4(w[d], 0x0A0A0A0Au) << (4 * d);
        const uint32_t x = w[d] ^ 0x0D0D0D0Du;
        anycr |= (x - 0x01010101u) & ~x & 0x80808080u;          // nonzero <=> some byte of the dword is '\r'
    }
    uint32_t cr = 0;
    if (anycr) {                                                // rare (no '\r' at all in Unix files)
        saw_cr = true;
#pragma unroll
        for (int d = 0; d < 4; ++d) cr |= byte_eq_mask4(w[d], 0x0D0D0D0Du) << (4 * d);
        // a '\r' ends a line unless a '\n' follows
        const uint32_t next_nl = (off + 16 < nbytes && bytes[off + 16] == '\n') ? 1u : 0u;
        cr &= ~((nl >> 1) | (next_nl << 15));
    }
    uint32_t mask = nl | cr;
    const long long valid = nbytes - off;
    if (valid < 16) mask &= (1u << valid) - 1u;
    return mask;
}

__global__ __launch_bounds__(NL_THREADS) void count_newlines_kernel(const uint8_t *__restrict__ bytes, long long nbytes,
                                                                     uint32_t *__restrict__ block_counts,
                                                                     uint32_t *__restrict__ has_cr) {
    __shared__ uint32_t s_cnt[NL_THREADS / 64];
    const long long off = ((long long)blockIdx.x * NL_THREADS + threadIdx.x) * 16;
    bool saw_cr;
    uint32_t c = __popc(newline_mask16(bytes, off, nbytes, saw_cr));
    if (__any(saw_cr) && (threadIdx.x & 63) == 0) atomicOr(has_cr, 1u);    // the chunk holds '\r': records check for "\r\n"
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
    bool saw_cr;
    uint32_t mask = newline_mask16(bytes, off, nbytes, saw_cr);
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
                                                      const uint32_t *__restrict__ line_ends, long long nrec,
                                                      const uint32_t *__restrict__ has_cr,
                                                      FastqRecord *__restrict__ records,
                                                      unsigned long long *__restrict__ error) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    FastqRecord rec;
    const int err = fastq_record_one(bytes, line_ends, r, *has_cr != 0, rec);
    records[r] = rec;
    if (err) atomicMin(error, (unsigned long long)r * 8ull + (unsigned long long)err);
}

__global__ void set_i64_kernel(long long *p, long long v) { *p = v; }

// ------------------------------------------------------------------------- pack from records
struct PackTable256 { uint8_t t[256]; };


// The sequence lines sit at arbitrary byte offsets of the file chunk, ~300 bytes apart.  A
// wave first copies the aligned dwords of its 64 lines into 64 LDS rows -- one
// global_load_lds_dword per line, lanes = consecutive dwords, so every line is fetched as one
// contiguous burst and the 64 copies are all in flight together (no VGPR round trip) -- and
// then every lane packs its own row, re-aligning the dwords with v_alignbyte_b32.
// Row stride is odd (in dwords): lanes reading dword d of their own rows hit 64 different banks.
__device__ __forceinline__ void glds_dword(const uint8_t *gsrc, uint32_t *lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst_wave_uniform, 4, 0, 0);
}

template <bool PLANES>
__global__ void pack_records_kernel(
    const uint8_t *__restrict__ bytes, const FastqRecord *__restrict__ records, const int32_t *__restrict__ begin,
    const int32_t *__restrict__ end, long long nreads, int max_len, int nchunks, int stride_dw, const PackTable256 tab,
    uint4 *__restrict__ packed, int32_t *__restrict__ lens, int32_t *__restrict__ invalid) {
    __shared__ uint8_t s_tab[256];
    __shared__ uint32_t s_spread[PLANES ? 256 : 1];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        s_tab[i] = tab.t[i];
        if (PLANES) s_spread[i] = spread_code((uint32_t)tab.t[i] & 15u);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const long long tile = (long long)blockIdx.x * waves + wave;
    const long long ntiles = (nreads + 63) >> 6;
    if (tile >= ntiles) return;
    const long long r = tile * 64 + lane;
    uint32_t soff = 0;
    int n = 0;
    if (r < nreads) {
        const FastqRecord rec = records[r];
        int a = begin ? begin[r] : 0, b = end ? end[r] : (int)rec.seq_len;
        a = max(0, min(a, (int)rec.seq_len));
        b = max(a, min(b, (int)rec.seq_len));
        soff = rec.seq_off + (uint32_t)a;
        n = min(b - a, max_len);
        if (lens) lens[r] = n;
    }
    const uint32_t sh = soff & 3u;
    const int nd = (n + (int)sh + 3) >> 2;                      // aligned dwords that hold the read
    uint32_t *rows = s_rows + (size_t)wave * 64 * stride_dw;
    const int cnt = (int)min<long long>(64, nreads - tile * 64);
    for (int i = 0; i < cnt; ++i) {                             // wave-uniform: row i <- line i
        const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)(soff - sh), i);
        const int nd_i = __builtin_amdgcn_readlane(nd, i);
        for (int d0 = 0; d0 < nd_i; d0 += 64)
            if (d0 + lane < nd_i) glds_dword(bytes + so + (size_t)(d0 + lane) * 4, rows + (size_t)i * stride_dw + d0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const uint32_t *p = rows + (size_t)lane * stride_dw;
    uint4 *dst = packed + (size_t)tile * nchunks * 64 + lane;
    bool zero_seen = false;
    // four bases per step out of the lane's LDS row (pack_fast.hpp); dwords from nd on are not read
    if (PLANES) pack_planes_row_fast(p, 0u, (uint32_t)nd, sh, n, nchunks, s_tab, dst, zero_seen);
    else pack_codes_row_fast(p, 0u, (uint32_t)nd, sh, n, nchunks, s_tab, dst, zero_seen);
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
        quality_trim_one(bytes, rec.qual_off + (uint32_t)a, b - a, cutoff_front, cutoff_back, base, s, e);
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
                                                          int discard_untrimmed, uint8_t *__restrict__ dest,
                                                          uint8_t *__restrict__ fail_mask) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const FastqRecord rec = records[r];
    const int a = begin[r], b = max(a, end[r]);
    const int ub = ubegin ? ubegin[r] - a : 0, ue = uend ? uend[r] - a : b - a;
    const uint32_t mask = read_filter_mask(bytes + rec.seq_off + a, b - a, ub, ue, matched ? matched[r] != 0 : false,
                                           min_len, max_len, max_n, discard_trimmed, discard_untrimmed);
    if (dest) dest[r] = (uint8_t)filter_destination(mask, 0u, false, 1);
    if (fail_mask) fail_mask[r] = (uint8_t)mask;
}

__global__ __launch_bounds__(256) void pair_filter_kernel(const uint8_t *__restrict__ mask1,
                                                          const uint8_t *__restrict__ mask2, long long n,
                                                          int min_affected, uint8_t *__restrict__ dest) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    dest[r] = (uint8_t)filter_destination(mask1[r], mask2[r], true, min_affected);
}

struct CompTable256 { uint8_t c[256]; };

__global__ __launch_bounds__(256) void insert_plan_kernel(
    const int16_t *__restrict__ ins, const int16_t *__restrict__ fb1, const int16_t *__restrict__ fb2,
    uint8_t *bytes1, const FastqRecord *__restrict__ records1, uint8_t *bytes2, const FastqRecord *__restrict__ records2,
    int32_t *__restrict__ begin1, int32_t *__restrict__ end1, int32_t *__restrict__ begin2, int32_t *__restrict__ end2,
    int32_t *__restrict__ uend1, int32_t *__restrict__ uend2, long long n, int min_insert_len, int symmetric,
    int trim_action, int correct_action, int min_qual_diff, const CompTable256 ct, uint8_t *__restrict__ matched1,
    uint8_t *__restrict__ matched2, int32_t *__restrict__ corrected, unsigned long long *__restrict__ error) {
    __shared__ uint8_t s_comp[256];
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    __syncthreads();
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int a1 = begin1[r], a2 = begin2[r];
    int len1 = max(0, end1[r] - a1), len2 = max(0, end2[r] - a2);
    InsertPlan P;
    insert_plan_matches(ins + 24 * r, fb1 + 8 * r, fb2 + 8 * r, len1, len2, min_insert_len, symmetric, correct_action >= 0, P);
    if (corrected) corrected[2 * r] = corrected[2 * r + 1] = 0;
    if (P.correct) {                                                   // correct_errors(..., truncate_seqs=True), :448-449
        const FastqRecord r1 = records1[r], r2 = records2[r];
        int32_t changed[2], newlen[2];
        correct_errors_one(bytes1 + r1.seq_off + a1, bytes1 + r1.qual_off + a1, len1, bytes2 + r2.seq_off + a2,
                           bytes2 + r2.qual_off + a2, len2, P.corr, correct_action, min_qual_diff, true, s_comp, changed,
                           newlen);
        if (changed[0] < 0) {
            atomicMin(error, (unsigned long long)r * 8ull + (unsigned long long)(-changed[0]));
        } else {
            len1 = newlen[0]; len2 = newlen[1];
            end1[r] = a1 + len1; end2[r] = a2 + len2;
            if (corrected) { corrected[2 * r] = changed[0]; corrected[2 * r + 1] = changed[1]; }
        }
    }
    int cut1, cut2;
    bool m1, m2;
    insert_plan_trim(P, len1, len2, trim_action, cut1, cut2, m1, m2);
    if (trim_action == 2) {                                            // mask: keep the length, remember the cut
        uend1[r] = a1 + cut1;
        uend2[r] = a2 + cut2;
    } else {
        end1[r] = a1 + cut1;
        end2[r] = a2 + cut2;
    }
    matched1[r] = m1 ? 1 : 0;
    matched2[r] = m2 ? 1 : 0;
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
        if (rec.flags & 1u) { wave_copy(o, bytes + fastq_name2_off(rec), fastq_name2_len(rec), lane); o += fastq_name2_len(rec); }
        if (lane == 0) o[0] = '\n';
        o += 1;
        wave_copy(o, bytes + rec.qual_off + a, kept, lane);
        o += kept;
        if (lane == 0) o[0] = '\n';
    }
}

// ---- staged formatter -------------------------------------------------------------------
// One wave formats a tile of EMIT_TILE consecutive records through LDS:
//   1. the tile's input span (first '@' .. end of the last quality line, ~10 KB for 150 bp
//      reads) is copied to LDS with coalesced 16-byte loads;
//   2. every lane pair assembles ITS record's output bytes at the record's place in an LDS image
//      of the tile's output span (LDS -> LDS, dword copies re-aligned in registers);
//   3. the image is written out with 16-byte stores aligned to the output buffer; the ragged
//      first / last bytes of the span with byte stores (neighbouring tiles never touch them).
// Output is never larger than input (same record layout, bases only removed), so one size
// bounds both stages.  Tiles that do not fit (very long names / reads) or whose records
// are not in file order take emit_kernel's path.
constexpr int EMIT_STAGE = 13 * 1024;                        // per stage; 2 stages per wave -> 6 waves per CU
// records per wave: 16; full-size stages for records of more than ~400 bytes, half-size stages (twice
// the waves per CU: the kernel is latency bound) for short-read files; 32 records per wave with full
// stages when the caller gives no hint

// LDS -> LDS copy of len bytes at arbitrary alignments: byte steps until dst is dword
// aligned, then one aligned ds_read_b32 + v_alignbyte_b32 + ds_write_b32 per 4 bytes (four
// dwords per batch so that the reads of a batch are in flight together), byte steps for the
// tail.  May read up to 3 bytes beyond src + len (the stages are padded).
__device__ __forceinline__ void lds_copy(uint8_t *dst, const uint8_t *src, uint32_t len) {
    uint32_t k = 0;
    const uint32_t head = min(len, (4u - ((uint32_t)(uintptr_t)dst & 3u)) & 3u);
    for (; k < head; ++k) dst[k] = src[k];
    const uint32_t sm = (uint32_t)(uintptr_t)(src + k) & 3u;
    const uint32_t *sp = (const uint32_t *)(src + k - sm);        // aligned dwords holding the source
    uint32_t *dp = (uint32_t *)(dst + k);
    const uint32_t nd = (len - k) >> 2;
    uint32_t prev = nd ? sp[0] : 0u, t = 0;
    for (; t + 4 <= nd; t += 4) {
        const uint32_t n1 = sp[t + 1], n2 = sp[t + 2], n3 = sp[t + 3], n4 = sp[t + 4];
        dp[t] = __builtin_amdgcn_alignbyte(n1, prev, sm);
        dp[t + 1] = __builtin_amdgcn_alignbyte(n2, n1, sm);
        dp[t + 2] = __builtin_amdgcn_alignbyte(n3, n2, sm);
        dp[t + 3] = __builtin_amdgcn_alignbyte(n4, n3, sm);
        prev = n4;
    }
    for (; t < nd; ++t) {
        const uint32_t nx = sp[t + 1];
        dp[t] = __builtin_amdgcn_alignbyte(nx, prev, sm);
        prev = nx;
    }
    for (k += nd * 4u; k < len; ++k) dst[k] = src[k];
}

template <int EMIT_TILE, int STAGE = EMIT_STAGE>
__global__ __launch_bounds__(64) void emit_staged_kernel(const uint8_t *__restrict__ bytes,
                                                         const FastqRecord *__restrict__ records,
                                                         const int32_t *__restrict__ begin, const int32_t *__restrict__ end,
                                                         const int32_t *__restrict__ ubegin, const int32_t *__restrict__ uend,
                                                         const uint8_t *__restrict__ dest, int which, long long n,
                                                         const long long *__restrict__ offsets, uint8_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_emit[];
    uint8_t *s_in = s_emit, *s_out = s_emit + STAGE;
    constexpr int LPR = 64 / EMIT_TILE;                       // lanes per record (the first two do the work)
    const int lane = threadIdx.x, slot = lane / LPR, part = lane % LPR;
    const long long r0 = (long long)blockIdx.x * EMIT_TILE;
    const int cnt = (int)min<long long>(EMIT_TILE, n - r0);
    const long long r = r0 + slot;
    const bool live = slot < cnt;
    FastqRecord rec = {0, 0, 0, 0, 0, 0, 0, 0};
    int a = 0, b = 0, ub = 0, ue = 0;
    long long off = 0;
    bool keep = false;
    if (live) {
        rec = records[r];
        a = begin[r];
        b = max(a, end[r]);
        ub = ubegin ? ubegin[r] : a;
        ue = uend ? uend[r] : b;
        off = offsets[r];
        keep = !dest || dest[r] == which;
    }
    const uint32_t rec_lo = rec.name_off - 1u, rec_hi = rec.qual_off + rec.qual_len;     // '@' .. last quality byte
    const uint32_t in_lo = __shfl(rec_lo, 0, 64), in_hi = __shfl(rec_hi, LPR * (cnt - 1), 64);
    const long long out_lo = offsets[r0], out_hi = offsets[r0 + cnt];
    const bool ordered = __all(!live || (rec_lo >= in_lo && rec_hi <= in_hi && rec.seq_off >= rec_lo &&
                                         rec.seq_off + rec.seq_len <= rec.qual_off && rec.name_off + rec.name_len <= rec.seq_off &&
                                         !(rec.flags & 2u)));
    const uint32_t mis_in = in_lo & 15u, mis_out = (uint32_t)(out_lo & 15);
    const bool fits = ordered && (in_hi - in_lo) + mis_in + 16u <= (uint32_t)STAGE &&
                      (uint32_t)(out_hi - out_lo) + mis_out + 16u <= (uint32_t)STAGE;
    if (out_hi == out_lo) return;
    if (!fits) {                                           // slow path: a byte per lane straight from / to global
        for (int i = 0; i < cnt; ++i) {
            const long long ri = r0 + i;
            if (dest && dest[ri] != which) continue;
            const FastqRecord rc = records[ri];
            const int ai = begin[ri], bi = max(ai, end[ri]);
            const uint32_t kept = (uint32_t)(bi - ai);
            uint8_t *o = out + offsets[ri];
            if (lane == 0) o[0] = '@';
            wave_copy(o + 1, bytes + rc.name_off, rc.name_len, lane);
            o += 1 + rc.name_len;
            if (lane == 0) o[0] = '\n';
            o += 1;
            const int ubi = ubegin ? ubegin[ri] : ai, uei = uend ? uend[ri] : bi;
            for (uint32_t k = (uint32_t)lane; k < kept; k += 64u) {
                const int pos = ai + (int)k;
                o[k] = (pos >= ubi && pos < uei) ? bytes[rc.seq_off + pos] : (uint8_t)'N';
            }
            o += kept;
            if (lane == 0) { o[0] = '\n'; o[1] = '+'; }
            o += 2;
            if (rc.flags & 1u) { wave_copy(o, bytes + fastq_name2_off(rc), fastq_name2_len(rc), lane); o += fastq_name2_len(rc); }
            if (lane == 0) o[0] = '\n';
            o += 1;
            wave_copy(o, bytes + rc.qual_off + ai, kept, lane);
            o += kept;
            if (lane == 0) o[0] = '\n';
        }
        return;
    }
    // 1. stage the input span
    {
        const uint8_t *src_al = bytes + (in_lo - mis_in);
        const uint32_t need = in_hi - in_lo + mis_in;
        for (uint32_t o = (uint32_t)lane * 16u; o < need; o += 64u * 16u) *(uint4 *)(s_in + o) = *(const uint4 *)(src_al + o);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // 2. two lanes format one record inside the output image: the even lane "@name\nSEQ\n",
    //    the odd lane "+[name]\nQUAL\n"
    if (keep) {
        const uint8_t *in = s_in + mis_in;                  // in[x - in_lo] = bytes[x]
        uint8_t *o = s_out + mis_out + (uint32_t)(off - out_lo);
        const uint32_t kept = (uint32_t)(b - a);
        if (part == 0) {
            *o++ = '@';
            lds_copy(o, in + (rec.name_off - in_lo), rec.name_len);
            o += rec.name_len;
            *o++ = '\n';
            const uint8_t *sq = in + (rec.seq_off - in_lo) + a;
            if (ub <= a && ue >= b) {
                lds_copy(o, sq, kept);
            } else {
                for (uint32_t k = 0; k < kept; ++k) {
                    const int pos = a + (int)k;
                    o[k] = (pos >= ub && pos < ue) ? sq[k] : (uint8_t)'N';
                }
            }
            o[kept] = '\n';
        } else if (part == 1) {
            o += 1u + rec.name_len + 1u + kept + 1u;
            *o++ = '+';
            if (rec.flags & 1u) { lds_copy(o, in + (rec.name_off - in_lo), rec.name_len); o += rec.name_len; }
            *o++ = '\n';
            lds_copy(o, in + (rec.qual_off - in_lo) + a, kept);
            o[kept] = '\n';
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // 3. write the image: aligned 16-byte blocks, byte stores at the ragged ends
    {
        uint8_t *dst_al = out + (out_lo - mis_out);          // 16-byte aligned when `out` is
        const uint32_t lo = mis_out, hi = mis_out + (uint32_t)(out_hi - out_lo);
        for (uint32_t o = (uint32_t)lane * 16u; o < hi; o += 64u * 16u) {
            if (o >= lo && o + 16u <= hi) {
                *(uint4 *)(dst_al + o) = *(const uint4 *)(s_out + o);
            } else {
                for (uint32_t k = max(o, lo); k < min(o + 16u, hi); ++k) dst_al[k] = s_out[k];
            }
        }
    }
}

// ------------------------------------------------------------------------- MergeOverlapping
__global__ __launch_bounds__(256) void merge_plan_kernel(const int16_t *__restrict__ align, const int32_t *__restrict__ need,
                                                         const FastqRecord *__restrict__ records1,
                                                         const int32_t *__restrict__ begin1, const int32_t *__restrict__ end1,
                                                         const int32_t *__restrict__ begin2, const int32_t *__restrict__ end2,
                                                         long long n, uint8_t *__restrict__ kind, uint32_t *__restrict__ sizes,
                                                         unsigned long long *__restrict__ error) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int len1 = max(0, end1[r] - begin1[r]), len2 = max(0, end2[r] - begin2[r]);
    const MergeShape m = merge_shape(align + 8 * r, len1, len2, need[r]);
    kind[r] = (uint8_t)m.kind;
    uint32_t s = 0;
    if (m.kind == MERGE_INVALID) atomicMin(error, (unsigned long long)r * 8ull + 4ull);
    else if (m.kind != MERGE_NONE) s = fastq_record_bytes(records1[r], m.len[0] + m.len[1]);
    sizes[r] = s;
}

// One wave writes the merged records of 64 consecutive pairs, one after the other (merge_emit_one).
__global__ __launch_bounds__(256) void merge_emit_kernel(const int16_t *__restrict__ align, const uint8_t *__restrict__ kind,
                                                         const uint8_t *__restrict__ bytes1,
                                                         const FastqRecord *__restrict__ records1,
                                                         const uint8_t *__restrict__ bytes2,
                                                         const FastqRecord *__restrict__ records2,
                                                         const int32_t *__restrict__ begin1, const int32_t *__restrict__ end1,
                                                         const int32_t *__restrict__ begin2, const int32_t *__restrict__ end2,
                                                         long long n, const CompTable256 ct, int mate_pass,
                                                         const long long *__restrict__ offsets, uint8_t *__restrict__ out) {
    __shared__ uint8_t s_comp[256];
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (r0 >= n) return;
    const int cnt = (int)min<long long>(64, n - r0);
    for (int i = 0; i < cnt; ++i) {
        const long long r = r0 + i;                               // wave-uniform
        const int k = kind[r];
        if (k == MERGE_NONE || k == MERGE_INVALID) continue;
        const int a1 = begin1[r], a2 = begin2[r];
        const int len1 = max(0, end1[r] - a1), len2 = max(0, end2[r] - a2);
        const MergeShape m = merge_shape(align + 8 * r, len1, len2, 0);
        merge_emit_one(out + offsets[r], m, records1[r], bytes1, a1, records2[r], bytes2, a2, len2, s_comp, mate_pass != 0,
                       lane, 64);
    }
}

// ErrorCorrectorMixin.correct_errors(read1, read2, alignment) of the pairs MergeOverlapping corrects
// (:900-902: a mismatch action is set, the alignment has errors, the insert aligner has not seen the pair).
__global__ __launch_bounds__(256) void merge_correct_kernel(const int16_t *__restrict__ align, const uint8_t *__restrict__ kind,
                                                            const uint8_t *__restrict__ insert_matched, uint8_t *bytes1,
                                                            const FastqRecord *__restrict__ records1, uint8_t *bytes2,
                                                            const FastqRecord *__restrict__ records2,
                                                            const int32_t *__restrict__ begin1, const int32_t *__restrict__ end1,
                                                            const int32_t *__restrict__ begin2, const int32_t *__restrict__ end2,
                                                            long long n, int action, int min_qual_diff, const CompTable256 ct,
                                                            int32_t *__restrict__ corrected,
                                                            unsigned long long *__restrict__ error) {
    __shared__ uint8_t s_comp[256];
    s_comp[threadIdx.x] = ct.c[threadIdx.x];
    __syncthreads();
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    if (corrected) corrected[2 * r] = corrected[2 * r + 1] = 0;
    const int k = kind[r];
    if (k == MERGE_NONE || k == MERGE_INVALID || align[8 * r + 5] <= 0 || (insert_matched && insert_matched[r])) return;
    const FastqRecord r1 = records1[r], r2 = records2[r];
    const int a1 = begin1[r], a2 = begin2[r];
    const int len1 = max(0, end1[r] - a1), len2 = max(0, end2[r] - a2);
    int32_t changed[2], newlen[2];
    correct_errors_one(bytes1 + r1.seq_off + a1, bytes1 + r1.qual_off + a1, len1, bytes2 + r2.seq_off + a2,
                       bytes2 + r2.qual_off + a2, len2, align + 8 * r, action, min_qual_diff, false, s_comp, changed, newlen);
    if (changed[0] < 0) atomicMin(error, (unsigned long long)r * 8ull + (unsigned long long)(-changed[0]));
    else if (corrected) { corrected[2 * r] = changed[0]; corrected[2 * r + 1] = changed[1]; }
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

// work layout: [block_counts u32 x nblk][block_base i64 x (nblk + 1)][sums u64 x scan_blocks(nblk)][has_cr u32]
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
    uint32_t *has_cr = (uint32_t *)((char *)sums + align256((size_t)scan_blocks(nblk) * 8));
    hipError_t me = hipMemsetAsync(has_cr, 0, 4, st);
    if (me != hipSuccess) return hip_fail(me, "hipMemsetAsync");
    hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)nblk), dim3(NL_THREADS), 0, st, d_bytes, (long long)nbytes, counts, has_cr);
    launch_scan(counts, nblk, base, sums, (long long *)d_nlines, st);
    return launched("fastq count launch");
}

int atr_fastq_index(const uint8_t *d_bytes, int64_t nbytes, const void *d_work, uint32_t *d_line_ends, int64_t nlines,
                    atr_fastq_record *d_records, int64_t *d_error, void *stream) {
    if (nbytes < 0 || nbytes >= (int64_t)0xFFFFFFF0ll || nlines < 0 || !d_error) return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_error, (long long)LLONG_MAX);
    if (nlines == 0 || nbytes == 0) return launched("fastq index launch");
    if (!d_bytes || !d_work || !d_line_ends || ((uintptr_t)d_bytes & 15)) return ATR_ERR_INVALID;
    const long long nblk = (nbytes + NL_BLOCK_BYTES - 1) / NL_BLOCK_BYTES;
    const long long *base = (const long long *)((const char *)d_work + align256((size_t)nblk * 4));
    const uint32_t *has_cr = (const uint32_t *)((const char *)base + align256((size_t)(nblk + 1) * 8) +
                                                align256((size_t)scan_blocks(nblk) * 8));
    hipLaunchKernelGGL(line_ends_kernel, dim3((unsigned)nblk), dim3(NL_THREADS), 0, st, d_bytes, (long long)nbytes, base, d_line_ends);
    const long long nrec = nlines / 4;
    if (nrec > 0) {
        if (!d_records) return ATR_ERR_INVALID;
        hipLaunchKernelGGL(records_kernel, dim3(grid256(nrec)), dim3(256), 0, st, d_bytes, d_line_ends, nrec, has_cr,
                           (FastqRecord *)d_records, (unsigned long long *)d_error);
    }
    return launched("fastq index launch");
}

int atr_pack_records(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                     const int32_t *d_end, int64_t nreads, int max_len, const uint8_t table[256], int planes,
                     uint8_t *d_packed, int32_t *d_lens, int32_t *d_invalid, void *stream) {
    if (nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || !table) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_bytes || !d_records || (max_len > 0 && !d_packed) || ((uintptr_t)d_bytes & 15)) return ATR_ERR_INVALID;
    PackTable256 tab;
    memcpy(tab.t, table, 256);
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    const int stride_dw = ((max_len + 6) / 4) | 1;              // dwords per LDS row (odd)
    const size_t per_wave = (size_t)64 * stride_dw * 4;
    const int waves = per_wave * 4 <= 65536 - 256 ? 4 : (per_wave * 2 <= 65536 - 256 ? 2 : 1);
    if (planes)
        hipLaunchKernelGGL(pack_records_kernel<true>, dim3((unsigned)((ntiles + waves - 1) / waves)), dim3(64 * waves),
                           per_wave * waves, (hipStream_t)stream, d_bytes, (const FastqRecord *)d_records, d_begin, d_end,
                           (long long)nreads, max_len, nchunks, stride_dw, tab, (uint4 *)d_packed, d_lens, d_invalid);
    else
        hipLaunchKernelGGL(pack_records_kernel<false>, dim3((unsigned)((ntiles + waves - 1) / waves)), dim3(64 * waves),
                           per_wave * waves, (hipStream_t)stream, d_bytes, (const FastqRecord *)d_records, d_begin, d_end,
                           (long long)nreads, max_len, nchunks, stride_dw, tab, (uint4 *)d_packed, d_lens, d_invalid);
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
    if (!d_bytes || !d_records || !d_begin || !d_end || ((uintptr_t)d_bytes & 15)) return ATR_ERR_INVALID;
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
                          int discard_trimmed, int discard_untrimmed, uint8_t *d_dest, uint8_t *d_fail_mask,
                          void *stream) {
    if (n < 0 || ((d_unmasked_begin == nullptr) != (d_unmasked_end == nullptr))) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_bytes || !d_records || !d_begin || !d_end || (!d_dest && !d_fail_mask)) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(read_filter_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_matched,
                       (long long)n, min_len, max_len, max_n, discard_trimmed, discard_untrimmed, d_dest, d_fail_mask);
    return launched("read_filter_kernel launch");
}

int atr_pair_filter_batch(const uint8_t *d_fail_mask1, const uint8_t *d_fail_mask2, int64_t n, int min_affected,
                          uint8_t *d_dest, void *stream) {
    if (n < 0 || (min_affected != 1 && min_affected != 2)) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_fail_mask1 || !d_fail_mask2 || !d_dest) return ATR_ERR_INVALID;
    hipLaunchKernelGGL(pair_filter_kernel, dim3(grid256(n)), dim3(256), 0, (hipStream_t)stream, d_fail_mask1, d_fail_mask2,
                       (long long)n, min_affected, d_dest);
    return launched("pair_filter_kernel launch");
}

int atr_insert_plan_batch(const atr_result *d_insert, const atr_result *d_fallback1, const atr_result *d_fallback2,
                          uint8_t *d_bytes1, const atr_fastq_record *d_records1, uint8_t *d_bytes2,
                          const atr_fastq_record *d_records2, int32_t *d_begin1, int32_t *d_end1, int32_t *d_begin2,
                          int32_t *d_end2, int32_t *d_unmasked_end1, int32_t *d_unmasked_end2, int64_t n,
                          int min_insert_len, int symmetric, int trim_action, int correct_action, int min_qual_difference,
                          const uint8_t comp[256], uint8_t *d_matched1, uint8_t *d_matched2, int32_t *d_corrected,
                          int64_t *d_error, void *stream) {
    if (n < 0 || trim_action < 0 || trim_action > 2 || correct_action < -1 || correct_action > 2) return ATR_ERR_INVALID;
    if (correct_action >= 0 && (!d_bytes1 || !d_records1 || !d_bytes2 || !d_records2 || !comp || !d_error))
        return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (d_error) hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_error, (long long)LLONG_MAX);
    if (n == 0) return launched("insert_plan_kernel launch");
    if (!d_insert || !d_fallback1 || !d_fallback2 || !d_begin1 || !d_end1 || !d_begin2 || !d_end2 || !d_matched1 ||
        !d_matched2 || (trim_action == 2 && (!d_unmasked_end1 || !d_unmasked_end2)))
        return ATR_ERR_INVALID;
    CompTable256 ct;
    memset(ct.c, 0, 256);
    if (comp) memcpy(ct.c, comp, 256);
    hipLaunchKernelGGL(insert_plan_kernel, dim3(grid256(n)), dim3(256), 0, st, (const int16_t *)d_insert,
                       (const int16_t *)d_fallback1, (const int16_t *)d_fallback2, d_bytes1,
                       (const FastqRecord *)d_records1, d_bytes2, (const FastqRecord *)d_records2, d_begin1, d_end1,
                       d_begin2, d_end2, d_unmasked_end1, d_unmasked_end2, (long long)n, min_insert_len, symmetric,
                       trim_action, correct_action, min_qual_difference, ct, d_matched1, d_matched2, d_corrected,
                       (unsigned long long *)d_error);
    return launched("insert_plan_kernel launch");
}

// work layout: [sizes u32 x n][sums u64 x scan_blocks(n)]
size_t atr_fastq_emit_work_bytes(int64_t n) {
    if (n < 0) return 0;
    return align256((size_t)n * 4) + align256((size_t)scan_blocks(n) * 8) + 256;
}

int atr_fastq_emit(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                   const int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end,
                   const uint8_t *d_dest, int dest, int64_t n, int record_bytes_hint, int64_t *d_offsets, void *d_work,
                   uint8_t *d_out, void *stream) {
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
    if (((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_bytes & 15) == 0) {
        if (record_bytes_hint > 400)
            hipLaunchKernelGGL(emit_staged_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(64), 2 * EMIT_STAGE, st, d_bytes,
                               (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest,
                               dest, (long long)n, (const long long *)d_offsets, d_out);
        else if (record_bytes_hint > 0 && record_bytes_hint <= 380)   // 8 records, 2 x 3.25 KB per wave: 24 waves per CU (-5 % more)
            hipLaunchKernelGGL((emit_staged_kernel<8, EMIT_STAGE / 4>), dim3((unsigned)((n + 7) / 8)), dim3(64), EMIT_STAGE / 2, st, d_bytes,
                               (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest,
                               dest, (long long)n, (const long long *)d_offsets, d_out);
        else if (record_bytes_hint > 0)                    // short records: half-size tiles and stages, 12 waves per CU (-15 %)
            hipLaunchKernelGGL((emit_staged_kernel<16, EMIT_STAGE / 2>), dim3((unsigned)((n + 15) / 16)), dim3(64), EMIT_STAGE, st, d_bytes,
                               (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest,
                               dest, (long long)n, (const long long *)d_offsets, d_out);
        else
            hipLaunchKernelGGL(emit_staged_kernel<32>, dim3((unsigned)((n + 31) / 32)), dim3(64), 2 * EMIT_STAGE, st, d_bytes,
                               (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest,
                               dest, (long long)n, (const long long *)d_offsets, d_out);
        return launched("emit_staged_kernel launch");
    }
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, d_bytes,
                       (const FastqRecord *)d_records, d_begin, d_end, d_unmasked_begin, d_unmasked_end, d_dest, dest,
                       (long long)n, (const long long *)d_offsets, d_out);
    return launched("emit_kernel launch");
}

size_t atr_merge_work_bytes(int64_t n) { return atr_fastq_emit_work_bytes(n); }

int atr_merge_plan_batch(const atr_result *d_align, const int32_t *d_need, const atr_fastq_record *d_records1,
                         const int32_t *d_begin1, const int32_t *d_end1, const int32_t *d_begin2, const int32_t *d_end2,
                         int64_t n, uint8_t *d_kind, int64_t *d_offsets, void *d_work, int64_t *d_error, void *stream) {
    if (n < 0 || !d_offsets || !d_error) return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_error, LLONG_MAX);
    if (n == 0) {
        hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, st, (long long *)d_offsets, 0ll);
        return launched("merge plan launch");
    }
    if (!d_align || !d_need || !d_records1 || !d_begin1 || !d_end1 || !d_begin2 || !d_end2 || !d_kind || !d_work)
        return ATR_ERR_INVALID;
    uint32_t *sizes = (uint32_t *)d_work;
    unsigned long long *sums = (unsigned long long *)((char *)d_work + align256((size_t)n * 4));
    hipLaunchKernelGGL(merge_plan_kernel, dim3(grid256(n)), dim3(256), 0, st, (const int16_t *)d_align, d_need,
                       (const FastqRecord *)d_records1, d_begin1, d_end1, d_begin2, d_end2, (long long)n, d_kind, sizes,
                       (unsigned long long *)d_error);
    launch_scan(sizes, n, (long long *)d_offsets, sums, nullptr, st);
    hipLaunchKernelGGL(emit_total_kernel, dim3(1), dim3(1), 0, st, sizes, (long long *)d_offsets, (long long)n);
    return launched("merge_plan_kernel launch");
}

int atr_merge_emit_batch(const atr_result *d_align, const uint8_t *d_kind, const uint8_t *d_insert_matched,
                         uint8_t *d_bytes1, const atr_fastq_record *d_records1, uint8_t *d_bytes2,
                         const atr_fastq_record *d_records2, const int32_t *d_begin1, const int32_t *d_end1,
                         const int32_t *d_begin2, const int32_t *d_end2, int64_t n, int correct_action,
                         int min_qual_difference, const uint8_t comp[256], const int64_t *d_offsets, int32_t *d_corrected,
                         int64_t *d_error, uint8_t *d_out, void *stream) {
    if (n < 0 || correct_action < -1 || correct_action > 2 || !comp) return ATR_ERR_INVALID;
    if (n == 0) return ATR_OK;
    if (!d_align || !d_kind || !d_bytes1 || !d_records1 || !d_bytes2 || !d_records2 || !d_begin1 || !d_end1 || !d_begin2 ||
        !d_end2 || !d_offsets || !d_error || !d_out)
        return ATR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    CompTable256 ct;
    memcpy(ct.c, comp, 256);
    const unsigned waves = (unsigned)(((n + 63) / 64 + 3) / 4);
    const FastqRecord *rec1 = (const FastqRecord *)d_records1, *rec2 = (const FastqRecord *)d_records2;
    // the mate's bases first: the reference reverse-complements read 2 before it corrects the pair (:887)
    hipLaunchKernelGGL(merge_emit_kernel, dim3(waves), dim3(256), 0, st, (const int16_t *)d_align, d_kind, d_bytes1, rec1,
                       d_bytes2, rec2, d_begin1, d_end1, d_begin2, d_end2, (long long)n, ct, 1, (const long long *)d_offsets,
                       d_out);
    if (correct_action >= 0)
        hipLaunchKernelGGL(merge_correct_kernel, dim3(grid256(n)), dim3(256), 0, st, (const int16_t *)d_align, d_kind,
                           d_insert_matched, d_bytes1, rec1, d_bytes2, rec2, d_begin1, d_end1, d_begin2, d_end2, (long long)n,
                           correct_action, min_qual_difference, ct, d_corrected, (unsigned long long *)d_error);
    hipLaunchKernelGGL(merge_emit_kernel, dim3(waves), dim3(256), 0, st, (const int16_t *)d_align, d_kind, d_bytes1, rec1,
                       d_bytes2, rec2, d_begin1, d_end1, d_begin2, d_end2, (long long)n, ct, 0, (const long long *)d_offsets,
                       d_out);
    return launched("merge_emit_kernel launch");
}

}  // extern "C"
