// prescan_c2.hip -- what would "pass A" of the two-pass C2 pre-pass cost?
// A stand-alone measurement, not product code: 10 M synthetic 150-base reads in the tile64 layout, one read per
// lane, and per column the exact-piece (PEX) scan of a 32-row adapter with k = 3: a Shift-And automaton over the
// four 8-base pieces (state bits = piece positions, one LDS mask per query code), hits folded per dword; then the
// read-end tests (exact overlaps of 3 .. 9 bases, position-constrained pieces for the longer ones).  Output per
// read: a block mask of the dwords with a piece hit and a tail flag.  The number to compare with: the bit-vector
// sweep this would replace for reads and columns without hits costs ~17 VALU ops per column (0.67 ms per 10 M
// reads on C2).
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/prescan_c2 tools/micro/prescan_c2.hip && /tmp/prescan_c2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Params {
    uint32_t mask[16];      // Shift-And masks by query code: bit 8p + i set <=> piece p has this code at position i
    uint32_t prefix[8];     // the adapter's first 9 bases, as nibbles (exact read-end overlaps)
    uint32_t piece5[6];     // 5-base pieces of the first 30 rows (read-end overlaps with errors), one nibble word each
};

__global__ __launch_bounds__(256) void prescan_kernel(const Params p, const uint4 *__restrict__ packed, long long nreads,
                                                      int nchunks, int n, uint32_t *__restrict__ out) {
    __shared__ uint32_t s_mask[16];
    if (threadIdx.x < 16) s_mask[threadIdx.x] = p.mask[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nreads + 63) >> 6;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const uint4 *tp = packed + (size_t)tile * nchunks * 64 + lane;
        const uint32_t START = 0x01010101u, END = 0x80808080u;
        uint32_t S = 0, blocks = 0, last[2] = {0, 0};
        int d_index = 0;
        for (int c = 0; c < nchunks; ++c) {
            const uint4 v = tp[(size_t)c * 64];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d, ++d_index) {
                const uint32_t w = w4[d];
                uint32_t ev = (w << 2) & 0x3C3C3C3Cu, od = (w >> 2) & 0x3C3C3C3Cu, H = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const uint32_t off = (((b & 1) ? od : ev) >> (8 * (b >> 1))) & 0xFFu;
                    const uint32_t m = *(const uint32_t *)((const char *)s_mask + off);
                    S = ((S << 1) | START) & m;
                    H |= S;
                }
                blocks |= ((H & END) ? 1u : 0u) << d_index;
                last[0] = last[1]; last[1] = w;
            }
        }
        // read end: the last 16 bases are in last[0], last[1] (n = 150: the read ends inside last[1])
        const int tail_nibbles = n & 7 ? n & 7 : 8;
        const uint64_t tail = (((uint64_t)last[1] << 32) | last[0]) >> (4 * (8 - (8 - tail_nibbles)));
        uint32_t flag = 0;
#pragma unroll
        for (int i = 3; i <= 9; ++i) {                                   // exact overlap of i bases: read[n - i:] == adapter[:i]
            const uint64_t have = (tail >> (4 * (16 - 8 + tail_nibbles - i))) & ((1ull << (4 * i)) - 1);
            const uint64_t want = (((uint64_t)p.prefix[1] << 32) | p.prefix[0]) & ((1ull << (4 * i)) - 1);
            flag |= have == want ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)                                      // 5-base pieces at their ~13 admissible positions each
#pragma unroll
            for (int pos = 0; pos < 13; ++pos) {
                const uint32_t have = (uint32_t)(tail >> (4 * ((pos + q) & 7))) & 0xFFFFFu;
                flag |= have == p.piece5[q] ? 2u : 0u;
            }
        const long long r = tile * 64 + lane;
        if (r < nreads) out[r] = blocks | (flag << 30);
    }
}

int main() {
    const long long nreads = 10000000;
    const int n = 150, nchunks = (n + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    const size_t words = (size_t)ntiles * nchunks * 64 * 4;
    std::vector<uint32_t> host(words);
    uint64_t x = 88172645463325252ull;
    const char *adapter = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC";
    auto code = [](char ch) { return ch == 'A' ? 1u : ch == 'C' ? 2u : ch == 'G' ? 4u : 8u; };
    for (long long r = 0; r < nreads; ++r) {
        uint8_t seq[160];
        for (int j = 0; j < n; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; seq[j] = (uint8_t)(1u << (x & 3)); }
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        if ((x & 127) < 60) {                                            // 47 %: the adapter from a random position on
            const int at = (int)((x >> 8) % 150);
            for (int j = at; j < n && j - at < 34; ++j) seq[j] = (uint8_t)code(adapter[j - at]);
        }
        const long long tile = r >> 6; const int lane = (int)(r & 63);
        for (int c = 0; c < nchunks; ++c)
            for (int d = 0; d < 4; ++d) {
                uint32_t w = 0;
                for (int b = 0; b < 8; ++b) { const int j = c * 32 + d * 8 + b; if (j < n) w |= (uint32_t)seq[j] << (4 * b); }
                host[(((size_t)tile * nchunks + c) * 64 + lane) * 4 + d] = w;
            }
    }
    Params p;
    memset(&p, 0, sizeof(p));
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i)
            for (uint32_t c = 0; c < 16; ++c)
                if (c == code(adapter[8 * q + i])) p.mask[c] |= 1u << (8 * q + i);
    for (int i = 0; i < 16; ++i) p.prefix[i >> 3] |= code(adapter[i]) << (4 * (i & 7));
    for (int q = 0; q < 6; ++q)
        for (int i = 0; i < 5; ++i) p.piece5[q] |= code(adapter[5 * q + i]) << (4 * i);
    uint32_t *d_packed, *d_out;
    CHECK(hipMalloc(&d_packed, words * 4));
    CHECK(hipMalloc(&d_out, nreads * 4));
    CHECK(hipMemcpy(d_packed, host.data(), words * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int grid : {2048, 4096, 8192}) {
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL(prescan_kernel, dim3(grid), dim3(256), 0, 0, p, (const uint4 *)d_packed, nreads, nchunks, n, d_out);
        CHECK(hipEventRecord(a, 0));
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep)
            hipLaunchKernelGGL(prescan_kernel, dim3(grid), dim3(256), 0, 0, p, (const uint4 *)d_packed, nreads, nchunks, n, d_out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("grid %5d: %.3f ms per 10 M reads (%.2f G reads/s)\n", grid, ms / reps, nreads / (ms / reps) / 1e6);
    }
    std::vector<uint32_t> res(nreads);
    CHECK(hipMemcpy(res.data(), d_out, nreads * 4, hipMemcpyDeviceToHost));
    long long hit = 0, tail = 0;
    for (long long r = 0; r < nreads; ++r) { hit += (res[r] & 0x3FFFFFFFu) != 0; tail += (res[r] >> 30) != 0; }
    printf("reads with a piece hit: %.3f, with a read-end flag: %.3f\n", (double)hit / nreads, (double)tail / nreads);
    return 0;
}
