// linked_group.hip -- linked adapters with the anchored 5' parts decided AT PACK TIME and adapter-uniform tiles (round 6).
//
// The fused pipeline (linked_kernels.hip) takes a tile64 batch of whole reads and runs every read's 3' part with the
// lane's OWN adapter: per-lane match masks, the one-pass pre-pass, 0.13 of the HBM roofline on C4.  The reference decides
// the 5' part first and then matches the 3' part of THAT adapter on read[front.rstop:] (LinkedAdapter.match_to,
// atropos/adapters/__init__.py:671-690, under AdapterCutter._best_match, commands/trim/modifiers.py:107-122).  An
// anchored 5' part only ever looks at the read's first m + k <= 32 bases, and every caller starts from ASCII -- so the
// 5' decision belongs where the read is first touched, and what is packed can already be what the 3' aligner wants:
//
//   G1  linked_front_ascii_kernel   the first 32 bases of every ASCII row -> nibble codes in registers -> the 5' stage
//                                   of linked_kernels.hip unchanged (literal compare, excl, exact pieces, the queued
//                                   anchored DP): `which`, the 5' record, a 2-byte (which, rstop) word; per-span totals
//   G2  linked_group_pack_kernel    rows staged through LDS like pack_kernel; read[rstop:] packed as bit planes into the
//                                   tile of ITS adapter's group: group g's reads, in batch order, fill consecutive slots
//                                   of one plane64 sub-batch (slots from the spans' totals: no atomics, deterministic),
//                                   with lens (n - rstop), the permutation and its inverse
//   3'  per group: atr_locate_planes_batch's pipeline (piece_kernels.hip) -- the two-pass pre-pass compiled at run time for
//                                   that ONE 3' aligner, ragged form -- on a side stream each; records land in slot order
//   G3  linked_group_finish_kernel  Adapter.match_to's acceptance test (adapters/__init__.py:386-398) per record and the
//                                   gather back into batch order
//
// The 3' records are relative to read[front.rstop:] as in the reference -- here because that IS what was packed.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "atropos_hip.h"
#include "linked_host.hpp"
#include "linked_blob.hpp"
#include "pack_fast.hpp"
#include "side_stream.hpp"

namespace atr {

int hip_fail(hipError_t e, const char *what);
bool piece_applies(const atr_aligner *a, int max_len, FilterParams *fp_out, PieceParams *pp_out);      // piece_kernels.hip
int piece_ragged_len(int max_len);
int launch_planes_prepass(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len,
                          uint4 *out, void *work, hipStream_t st, int grid_div, PlanesCall *pc);
int launch_planes_tail(const atr_aligner *a, const uint4 *planes, const int32_t *lens, long long nreads, int max_len, uint4 *out,
                       const PlanesCall &pc, hipStream_t st, bool one_stream);

struct GroupTable { uint8_t t[256]; };

constexpr int GROUP_SPAN_MAX = 1024;                 // tiles of a G2 block's span (its per-tile slot bases live in LDS)
constexpr int GROUP_SUB = 4;                         // G1 blocks per G2 span
constexpr uint32_t GROUP_NONE = 7u;                  // gmeta: no 5' match
constexpr size_t GROUP_HEAD = 96 * 1024;             // workspace behind the (which, rstop) words: span totals (4096 x 16 B), info

// gmeta[r] = which (0 .. 3, 7: none) | rstop << 3
__device__ __forceinline__ uint16_t group_meta(int which, int rstop) { return (uint16_t)((which < 0 ? GROUP_NONE : (uint32_t)which) | ((uint32_t)rstop << 3)); }

// tiles [t0, t1) of G2 block b; G1 block (b, q) takes the q-th of its GROUP_SUB parts
__device__ __forceinline__ void group_span(long long ntiles, int nb, int b, long long &t0, long long &t1) {
    const long long per = (ntiles + nb - 1) / nb;
    t0 = min(ntiles, per * (long long)b);
    t1 = min(ntiles, t0 + per);
}

// The first min(n, 32) bases of row r as four dwords of nibble codes (code 0 past the read's end): what chunk 0 of a
// tile64 batch holds.  Aligned dword loads around the row's start, four bases per step as pack_codes_row_fast.
__device__ __forceinline__ void front_w0_ascii(const uint8_t *ascii, const uint8_t *buf_end, long long row_stride, long long r,
                                               int n, const uint8_t *s_tab, const PackLetters &L, uint32_t (&w0)[4]) {
    // three aligned 16-byte pieces cover the 32 bases wherever the row starts (a wave's load touches 64 different lines
    // either way: what counts is the number of load instructions -- nine dword loads per lane measured 0.92 ms per 12.5 M)
    const uint8_t *row = ascii + r * row_stride;
    const uint32_t mis16 = (uint32_t)((uintptr_t)row & 15), mis = mis16 & 3u;
    const uint8_t *al = row - mis16;
    const int nb = min(n, 32);
    uint32_t p12[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint8_t *p = al + 16 * i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (16 * i < (int)mis16 + nb) {
            if (p >= ascii && p + 16 <= buf_end) v = *(const uint4 *)p;
            else {
                uint32_t t[4] = {0u, 0u, 0u, 0u};
                for (int b = 0; b < 16; ++b) if (p + b >= ascii && p + b < buf_end) t[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
        p12[4 * i] = v.x; p12[4 * i + 1] = v.y; p12[4 * i + 2] = v.z; p12[4 * i + 3] = v.w;
    }
    // the nine dwords from the row's own dword on: a per-lane start of 0 .. 3 dwords into the pieces
    const uint32_t d0 = mis16 >> 2;
    uint32_t raw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t v = p12[i];
        if (d0 == 1u) v = p12[i + 1];
        if (d0 == 2u) v = p12[i + 2];
        if (d0 == 3u) v = p12[i + 3];
        raw[i] = v;
    }
    w0[0] = w0[1] = w0[2] = w0[3] = 0u;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        uint32_t w = __builtin_amdgcn_alignbyte(raw[g + 1], raw[g], mis);
        const int left = nb - 4 * g;
        if (left < 4) w = left <= 0 ? 0u : (w & ((1u << (8 * left)) - 1u));
        const uint32_t idx = (w >> 1) & 0x03030303u;
        uint32_t nib = 0u;
        if (L.ok && __builtin_amdgcn_perm(0u, PACK_LETTERS_BY_INDEX, idx) == w) {
            uint32_t cc = __builtin_amdgcn_perm(0u, L.lutc, idx);
            cc = (cc | (cc >> 4)) & 0x00FF00FFu;
            nib = (cc | (cc >> 8)) & 0xFFFFu;
        } else if (left > 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < left) nib |= ((uint32_t)s_tab[(w >> (8 * b)) & 255u] & 15u) << (4 * b);
        }
        w0[g >> 1] |= nib << (16 * (g & 1));
    }
}

// ---- G1: the 5' stage on ASCII rows ---------------------------------------------------------------------------------
template <bool RAGGED, bool AND_MODE>
__global__ __launch_bounds__(256) void linked_front_ascii_kernel(const LinkedBlob *__restrict__ blob, const uint8_t *__restrict__ ascii,
                                                                 long long row_stride, const int32_t *__restrict__ lens,
                                                                 long long nreads, int max_len, const GroupTable tab,
                                                                 uint16_t *__restrict__ which_out, uint4 *__restrict__ front_out,
                                                                 uint16_t *__restrict__ gmeta, uint32_t *__restrict__ spantot, int nb) {
    __shared__ __attribute__((aligned(16))) LinkedBlob S;
    __shared__ uint8_t s_tab[256];
    __shared__ uint32_t s_stream[4][FRONT_STREAM][64];
    __shared__ uint32_t s_words[4][LINKED_ROUND * 64];
    __shared__ uint32_t s_counts[4][LINKED_ROUND * 16];
    __shared__ uint16_t s_queue[4][LINKED_TASKS];
    __shared__ uint16_t s_list[4][LINKED_ROUND * 64];
    __shared__ uint32_t s_tot[LINKED_MAX];
    for (int i = threadIdx.x; i < (int)(sizeof(LinkedBlob) / 4); i += 256) ((uint32_t *)&S)[i] = ((const uint32_t *)blob)[i];
    s_tab[threadIdx.x] = tab.t[threadIdx.x];
    if (threadIdx.x < LINKED_MAX) s_tot[threadIdx.x] = 0u;
    __syncthreads();

    const PackLetters L = pack_letters(s_tab);
    const int nad = rfl(S.p.n), ngroups = rfl(S.p.ngroups);
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    uint32_t *s_word = s_words[wave], *s_count = s_counts[wave];
    uint16_t *queue = s_queue[wave];
    uint32_t *ns = &s_stream[wave][0][lane];
    const long long ntiles = (nreads + 63) >> 6;
    const uint8_t *buf_end = ascii + nreads * row_stride;
    const int span = (int)blockIdx.x / GROUP_SUB, sub = (int)blockIdx.x % GROUP_SUB;
    long long s0, s1;
    group_span(ntiles, nb, span, s0, s1);
    const long long per = (s1 - s0 + GROUP_SUB - 1) / GROUP_SUB;
    const long long t0 = min(s1, s0 + per * sub), t1 = min(s1, t0 + per);
    const auto load = [&](long long tile, int l, int n, uint32_t (&w)[4]) {
        front_w0_ascii(ascii, buf_end, row_stride, tile * 64 + l, n, s_tab, L, w);
    };
    uint32_t mine[LINKED_MAX] = {0u, 0u, 0u, 0u};                            // this wave's reads per group (wave-uniform)
    for (long long tile_first = t0 + wave; tile_first < t1; tile_first += 4 * LINKED_ROUND) {
        const int slots = (int)min((long long)LINKED_ROUND, (t1 - tile_first + 3) / 4);
        // (1) the literal compare of every adapter on every read (linked_kernels.hip, the same steps)
        int nopen = 0;
        const uint32_t all_ad = (1u << nad) - 1u;
        for (int slot = 0; slot < slots; ++slot) {
            const long long tile = tile_first + 4 * slot;
            const long long r = tile * 64 + lane;
            const bool live = r < nreads;
            const int n = live ? (RAGGED ? min(max(lens[r], 0), max_len) : max_len) : 0;
            uint32_t w0[4] = {0u, 0u, 0u, 0u};
            if (live) load(tile, lane, n, w0);
            uint32_t word = FRONT_NONE, count = 0u, open = live ? all_ad : 0u;
            for (int a = 0; a < nad; ++a) {
                const FrontParams &fp = S.p.f[a];
                const int m = rfl(fp.m);
                const bool exact = rfl(fp.accept_full) != 0 && front_exact(fp.code, fp.code_mask, w0);
                if (exact) {
                    ++count;
                    word = min(word, front_word(a, m, m, 0));
                    open &= ~((1u << a) | (uint32_t)rfl((int)S.p.excl[a]));
                }
            }
            s_word[slot * 64 + lane] = live ? word : FRONT_NONE;
            uint32_t packed_counts = live ? count : 0u;
            packed_counts |= (uint32_t)__shfl_down((int)packed_counts, 1, 64) << 8;
            packed_counts |= (uint32_t)__shfl_down((int)packed_counts, 2, 64) << 16;
            if ((lane & 3) == 0) s_count[(slot * 64 + lane) >> 2] = packed_counts;
            const unsigned long long om = __ballot(open != 0u);
            if (open != 0u) s_list[wave][nopen + __popcll(om & ((1ull << lane) - 1ull))] = (uint16_t)((slot * 64 + lane) | (open << 9));
            nopen += (int)__popcll(om);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // (2) the exact-piece test of the open (read, adapter) pairs, the listed reads 64 at a time; DP tasks queued
        int ntasks = 0;
        for (int base = 0; base < nopen; base += 64) {
            if (ntasks + 64 * nad > LINKED_TASKS) {
                linked_drain_with<RAGGED, AND_MODE>(S, ngroups, queue, ntasks, lens, tile_first, max_len, s_word, s_count, ns, lane, load);
                ntasks = 0;
            }
            const bool valid = base + lane < nopen;
            const uint32_t entry = (uint32_t)s_list[wave][valid ? base + lane : base];
            const int cell = (int)(entry & 511u);
            const uint32_t open = valid ? entry >> 9 : 0u;
            const long long tile = tile_first + 4 * (cell >> 6);
            const long long r = tile * 64 + (cell & 63);
            const int n = RAGGED ? min(max(lens[r], 0), max_len) : max_len;
            uint32_t w0[4];
            load(tile, cell & 63, n, w0);
            bool pex_ok[LINKED_MAX];
            const bool shared = rfl(S.p.pex_shared) != 0;
            if (shared) front_pex_candidates_shared<AND_MODE>(S.p.f, nad, w0, pex_ok);
            for (int a = 0; a < nad; ++a) {
                const FrontParams &fp = S.p.f[a];
                bool hit;
                if (shared) {
                    hit = pex_ok[0];
#pragma unroll
                    for (int t = 1; t < LINKED_MAX; ++t) if (a == t) hit = pex_ok[t];
                } else {
                    hit = front_pex_candidate<AND_MODE>(fp.pex_code, fp.pex_mask, fp.pex_off, rfl(fp.npieces), rfl(fp.k), w0);
                }
                const bool cand = ((open >> a) & 1u) != 0u && hit;
                const unsigned long long votes = __ballot(cand);
                if (cand) queue[ntasks + __popcll(votes & ((1ull << lane) - 1ull))] = (uint16_t)((cell << 6) | a);
                ntasks += (int)__popcll(votes);
            }
        }
        linked_drain_with<RAGGED, AND_MODE>(S, ngroups, queue, ntasks, lens, tile_first, max_len, s_word, s_count, ns, lane, load);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // (3) what every read gets: 5' record, `which`, the (which, rstop) word of the pack pass
        for (int slot = 0; slot < slots; ++slot) {
            const long long tile = tile_first + 4 * slot;
            const long long r = tile * 64 + lane;
            const bool live = r < nreads;
            const uint32_t word = s_word[slot * 64 + lane];
            const int which = word == FRONT_NONE ? -1 : (int)(word >> 24);
            if (live) {
                uint32_t frec[4];
                front_word_record(word, S.p.f[which < 0 ? 0 : which].m, frec);
                front_out[r] = make_uint4(frec[0], frec[1], frec[2], frec[3]);
                const uint32_t count = (s_count[(slot * 64 + lane) >> 2] >> (8 * (lane & 3))) & 0xFFu;
                which_out[r] = (uint16_t)((uint32_t)(which & 0xFF) | (count << 8));
                gmeta[r] = group_meta(which, which < 0 ? 0 : (int)((word >> 16) & 0xFFu));
            }
#pragma unroll
            for (int g = 0; g < LINKED_MAX; ++g) mine[g] += (uint32_t)__popcll(__ballot(live && which == g));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) if (mine[g]) atomicAdd(&s_tot[g], mine[g]);
    }
    __syncthreads();
    if (threadIdx.x < LINKED_MAX && s_tot[threadIdx.x]) atomicAdd(&spantot[span * LINKED_MAX + threadIdx.x], s_tot[threadIdx.x]);
}

// ---- G2: read[rstop:] as bit planes into the tiles of its adapter's group -----------------------------------------
// info[g] = reads of group g, info[4 + g] = first tile of group g's sub-batch (block 0 writes them).
__global__ __launch_bounds__(256) void linked_group_pack_kernel(const uint8_t *__restrict__ ascii, long long row_stride,
                                                                const int32_t *__restrict__ lens, long long nreads, int max_len,
                                                                int nchunks, const GroupTable tab, const uint16_t *__restrict__ gmeta,
                                                                const uint32_t *__restrict__ spantot, int nb,
                                                                uint4 *__restrict__ grouped, int32_t *__restrict__ glens,
                                                                int32_t *__restrict__ perm, int32_t *__restrict__ slot_of,
                                                                long long *__restrict__ info) {
    __shared__ uint8_t s_tab[256];
    __shared__ uint32_t s_all[LINKED_MAX], s_before[LINKED_MAX];
    __shared__ uint16_t s_tile[GROUP_SPAN_MAX][LINKED_MAX];              // per tile of the span and group: first slot inside the span
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];
    s_tab[threadIdx.x] = tab.t[threadIdx.x];
    if (threadIdx.x < LINKED_MAX) s_all[threadIdx.x] = s_before[threadIdx.x] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nreads + 63) >> 6;
    long long t0, t1;
    group_span(ntiles, nb, (int)blockIdx.x, t0, t1);
    {   // reads of every group in all spans / in the spans before this one
        uint32_t all[LINKED_MAX] = {0u, 0u, 0u, 0u}, before[LINKED_MAX] = {0u, 0u, 0u, 0u};
        for (int b = threadIdx.x; b < nb; b += 256) {
            const uint4 v = *(const uint4 *)(spantot + (size_t)b * LINKED_MAX);
            const uint32_t c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int g = 0; g < LINKED_MAX; ++g) { all[g] += c[g]; if (b < (int)blockIdx.x) before[g] += c[g]; }
        }
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) {
            if (all[g]) atomicAdd(&s_all[g], all[g]);
            if (before[g]) atomicAdd(&s_before[g], before[g]);
        }
    }
    // per tile of the span: its reads of every group
    for (long long tile = t0 + wave; tile < t1; tile += 4) {
        const long long r = tile * 64 + lane;
        const uint32_t which = r < nreads ? ((uint32_t)gmeta[r] & 7u) : GROUP_NONE;
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) {
            const uint32_t c = (uint32_t)__popcll(__ballot(which == (uint32_t)g));
            if (lane == g) s_tile[tile - t0][g] = (uint16_t)c;
        }
    }
    __syncthreads();
    // exclusive scan over the span's tiles, wave g for group g, 64 tiles a step
    if (wave < LINKED_MAX) {
        uint32_t run = 0u;
        const int nt = (int)(t1 - t0);
        for (int base = 0; base < nt; base += 64) {
            const int i = base + lane;
            const uint32_t c = i < nt ? (uint32_t)s_tile[i][wave] : 0u;
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (i < nt) s_tile[i][wave] = (uint16_t)(run + inc - c);
            run += (uint32_t)__shfl((int)inc, 63, 64);
        }
    }
    __syncthreads();
    long long gbase[LINKED_MAX];                                        // first slot of group g
    {
        long long at = 0;
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) { gbase[g] = at; at += ((long long)s_all[g] + 63) & ~63ll; }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < LINKED_MAX) { info[threadIdx.x] = (long long)s_all[threadIdx.x]; info[LINKED_MAX + threadIdx.x] = gbase[threadIdx.x] >> 6; }
        // the slots behind a group's last read in its last tile: reads of length 0, code 0 everywhere
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) {
            const long long first = gbase[g] + (long long)s_all[g], last = (first + 63) & ~63ll;
            for (long long slot = first + threadIdx.x; slot < last; slot += 256) {
                glens[slot] = 0; perm[slot] = -1;
                for (int c = 0; c < nchunks; ++c) grouped[((size_t)(slot >> 6) * nchunks + c) * 64 + (slot & 63)] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    const size_t wave_bytes = (((size_t)64 * row_stride + PACK_STAGE_SLACK) + 15) & ~(size_t)15;
    uint8_t *stage = s_stage + (size_t)wave * wave_bytes;
    for (long long tile = t0 + wave; tile < t1; tile += 4) {
        const long long r = tile * 64 + lane;
        const bool live = r < nreads;
        const uint32_t meta = live ? (uint32_t)gmeta[r] : GROUP_NONE;
        const uint32_t which = meta & 7u;
        const int start = (int)(meta >> 3);
        const bool has = live && which != GROUP_NONE;
        const int n = has ? max(0, min((lens ? lens[r] : max_len), max_len) - start) : 0;
        const uint32_t mis = pack_stage_tile(stage, ascii, row_stride, nreads, tile, lane);
        long long slot = -1;
#pragma unroll
        for (int g = 0; g < LINKED_MAX; ++g) {
            const unsigned long long m = __ballot(has && which == (uint32_t)g);
            if (has && which == (uint32_t)g)
                slot = gbase[g] + (long long)s_before[g] + (long long)s_tile[tile - t0][g] + (long long)__popcll(m & ((1ull << lane) - 1ull));
        }
        if (live) slot_of[r] = (int32_t)slot;
        if (has) {
            glens[slot] = n;
            perm[slot] = (int32_t)r;
            const uint32_t rowoff = mis + (uint32_t)lane * (uint32_t)row_stride + (uint32_t)start;
            bool zero_seen = false;
            uint4 *dst = grouped + ((size_t)(slot >> 6) * nchunks) * 64 + (slot & 63);
            pack_planes_row_fast((const uint32_t *)stage, rowoff >> 2, 0x7fffffffu, rowoff & 3u, n, nchunks, s_tab, dst, zero_seen);
        }
        __builtin_amdgcn_wave_barrier();                                 // (the stage is rewritten by the next tile)
    }
}

// ---- G3: acceptance test + back into batch order ----------------------------------------------------------------
// One pass in batch order behind the groups' joins: the gather from the slab is cheap -- a tile's reads of one group sit
// on consecutive slots -- and every line of `back` is written once, whole.  (A pass per group in slot order, scattering
// 16-byte records into `back`, measured 35 - 135 us per group on the groups' streams: each line written four times,
// partially.)  The 3' adapters' acceptance tables in LDS.
__global__ __launch_bounds__(256) void linked_group_finish_kernel(const LinkedWaveBlob *__restrict__ blob, const uint4 *__restrict__ slab,
                                                                  const int32_t *__restrict__ slot_of, const uint16_t *__restrict__ which,
                                                                  long long nreads, uint4 *__restrict__ back) {
    __shared__ LinkedPost s_post[LINKED_MAX];
    for (int i = threadIdx.x; i < (int)(sizeof(s_post) / 4); i += 256) ((uint32_t *)s_post)[i] = ((const uint32_t *)blob->post)[i];
    __syncthreads();
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nreads; r += (long long)gridDim.x * 256) {
        const int slot = slot_of[r];
        uint32_t rec[4];
        rec_none(rec);
        if (slot >= 0) {
            const uint4 v = slab[slot];
            rec[0] = v.x; rec[1] = v.y; rec[2] = v.z; rec[3] = v.w;
            const LinkedPost &post = s_post[which[r] & 3u];
            linked_finish(rec, 0, post.m, post.min_overlap, post.pf_thr, post.accept_full != 0, post.rmp, post.rmp_ld, post.max_rmp);
        }
        back[r] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
}

static int group_blocks(size_t dyn_lds) {
    static thread_local int cached_dev = -1, cached = 0;
    static thread_local size_t cached_lds = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 512;
    if (dev != cached_dev || dyn_lds != cached_lds) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linked_group_pack_kernel, 256, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached = per_cu * cus;
        cached_dev = dev;
        cached_lds = dyn_lds;
    }
    return cached;
}

// Is the set inside the grouped pipeline's envelope for reads of at most max_len bases?  Every 3' aligner must be one the
// two-pass pre-pass takes in its ragged form.
static bool group_applies(const atr_linked_set *s, int max_len) {
    if (!s || max_len < 1 || max_len > ATR_MAX_READ_LEN) return false;
    for (int a = 0; a < s->p.n; ++a)
        if (!piece_applies(&s->back[a], piece_ragged_len(max_len), nullptr, nullptr)) return false;
    return true;
}

}  // namespace atr

using namespace atr;

extern "C" {

int atr_linked_group_applies(const atr_linked_set *s, int max_len) { return group_applies(s, max_len) ? 1 : 0; }

size_t atr_linked_group_bytes(int64_t nreads, int max_len) {
    if (nreads < 0 || max_len < 0) return 0;
    return (size_t)(((nreads + 63) / 64 + LINKED_MAX) * ((max_len + 31) / 32)) * 64 * 16;
}

size_t atr_linked_group_work_bytes(const atr_linked_set *s, int64_t nreads) {
    if (!s || nreads < 0) return 0;
    // G1 / G2: (which, rstop) words + span totals + the info block; the 3' calls: a workspace per group, carved by the
    // groups' sizes (fast_work_bytes is affine in the read count)
    return (size_t)nreads * 2 + GROUP_HEAD + 256 + fast_work_bytes(nreads) + (size_t)LINKED_MAX * (fast_work_bytes(64) + 4096);
}

int atr_linked_group_pack(const atr_linked_set *s, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                          int64_t nreads, int max_len, const uint8_t table[256], uint8_t *d_grouped, int32_t *d_glens,
                          int32_t *d_perm, int32_t *d_slot_of, int8_t *d_which, atr_result *d_front, int64_t info[8],
                          void *d_work, void *stream) {
    if (!s || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN || row_stride < max_len || !table || !info) return ATR_ERR_INVALID;
    for (int i = 0; i < 2 * LINKED_MAX; ++i) info[i] = 0;
    if (nreads == 0) return ATR_OK;
    if (!d_ascii || !d_grouped || !d_glens || !d_perm || !d_slot_of || !d_which || !d_front || !d_work) return ATR_ERR_INVALID;
    if (!group_applies(s, max_len) || row_stride > 256 || nreads > 0x7fffffffll - 512) return ATR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    GroupTable tab;
    memcpy(tab.t, table, 256);
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    const size_t per_wave = (((size_t)64 * row_stride + PACK_STAGE_SLACK) + 15) & ~(size_t)15;
    if (4 * per_wave + 16 * 1024 > 64 * 1024) {
        static thread_local size_t granted = 0;
        if (granted < 4 * per_wave) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linked_group_pack_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * per_wave));
            if (e != hipSuccess) { (void)hipGetLastError(); return ATR_ERR_UNSUPPORTED; }
            granted = 4 * per_wave;
        }
    }
    int nb = (int)std::max<long long>(1, std::min<long long>(group_blocks(4 * per_wave), (ntiles + 3) / 4));
    nb = std::min(nb, 4096);
    if ((ntiles + nb - 1) / nb > GROUP_SPAN_MAX) return ATR_ERR_UNSUPPORTED;
    // workspace: [gmeta: nreads x 2][spantot: 4096 x 4 x 4][info: 8 x 8]
    uint16_t *gmeta = (uint16_t *)d_work;
    uint32_t *spantot = (uint32_t *)(((uintptr_t)(gmeta + nreads) + 15) & ~(uintptr_t)15);
    long long *dinfo = (long long *)(spantot + 4096 * LINKED_MAX);
    hipError_t e = hipMemsetAsync(spantot, 0, (size_t)nb * LINKED_MAX * 4, st);
    if (e != hipSuccess) return hip_fail(e, "linked group memset");
    const LinkedBlob *blob = (const LinkedBlob *)s->d_params;
    const bool ragged = d_lens != nullptr, and_mode = s->p.and_mode != 0;
    const dim3 g1(nb * GROUP_SUB), block(256);
    uint16_t *which = (uint16_t *)d_which;
    uint4 *front = (uint4 *)d_front;
#define ATR_G1(R, A) hipLaunchKernelGGL((linked_front_ascii_kernel<R, A>), g1, block, 0, st, blob, d_ascii, (long long)row_stride, d_lens, \
                                        (long long)nreads, max_len, tab, which, front, gmeta, spantot, nb)
    if (ragged) { if (and_mode) ATR_G1(true, true); else ATR_G1(true, false); }
    else        { if (and_mode) ATR_G1(false, true); else ATR_G1(false, false); }
#undef ATR_G1
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "linked_front_ascii_kernel launch");
    hipLaunchKernelGGL(linked_group_pack_kernel, dim3(nb), block, 4 * per_wave, st, d_ascii, (long long)row_stride, d_lens,
                       (long long)nreads, max_len, nchunks, tab, (const uint16_t *)gmeta, (const uint32_t *)spantot, nb,
                       (uint4 *)d_grouped, d_glens, d_perm, d_slot_of, dinfo);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "linked_group_pack_kernel launch");
    long long host[2 * LINKED_MAX];
    e = hipMemcpyAsync(host, dinfo, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "linked group info");
    for (int i = 0; i < 2 * LINKED_MAX; ++i) info[i] = host[i];
    return ATR_OK;
}

int atr_linked_group_match(const atr_linked_set *s, const uint8_t *d_grouped, const int32_t *d_glens, const int64_t info[8],
                           int max_len, const int32_t *d_slot_of, const int8_t *d_which, int64_t nreads, atr_result *d_slab,
                           atr_result *d_back, void *d_work, void *stream) {
    if (!s || !info || nreads < 0 || max_len < 1 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!d_grouped || !d_glens || !d_slab || !d_work) return ATR_ERR_INVALID;
    if (d_back && (!d_slot_of || !d_which)) return ATR_ERR_INVALID;
    if (!group_applies(s, max_len)) return ATR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = (max_len + 31) / 32;
    static thread_local SideStream side[LINKED_MAX];
    long long total = 0;
    for (int g = 0; g < s->p.n; ++g) {
        if (info[g] < 0 || info[LINKED_MAX + g] < 0) return ATR_ERR_INVALID;
        total += info[g];
    }
    if (total > nreads) return ATR_ERR_INVALID;
    // the groups' workspaces behind the pack pass's words
    uint8_t *wbase = (uint8_t *)d_work + (((size_t)nreads * 2 + GROUP_HEAD + 255) & ~(size_t)255);
    // streams the groups are dealt over (1: all on the caller's stream, one after the other); A/B switch ATR_GROUP_STREAMS
    static const int nstreams = [] { const char *x = getenv("ATR_GROUP_STREAMS"); const int v = x ? atoi(x) : 0; return v >= 1 && v <= LINKED_MAX ? v : 2; }();
    const bool serial = nstreams == 1;
    static const int grid_div = [] { const char *x = getenv("ATR_GROUP_GRIDDIV"); return x ? atoi(x) : 1; }();
    static const bool one_stream = [] { const char *x = getenv("ATR_GROUP_ONE_STREAM"); return !(x && x[0] == '0'); }();
    hipError_t e = hipSuccess;
    int live_groups = 0;
    for (int g = 0; g < s->p.n; ++g) live_groups += info[g] > 0 ? 1 : 0;
    const bool side_by_side = !serial && live_groups > 1;
    bool waited[LINKED_MAX] = {false, false, false, false};
    if (side_by_side) {
        if (!side[0].ready()) return ATR_ERR_HIP;
        e = hipEventRecord(side[0].fork, st);
        if (e != hipSuccess) return hip_fail(e, "linked group fork");
    }
    // Every group's pre-pass first, dealt over the streams (slot 0 = the caller's stream), then every group's DP tail on
    // the stream of its pre-pass: the pre-passes keep the device full back to back, the tails -- few, short, latency-bound
    // launches -- run side by side behind them.  (A group's whole call one after the other measured 1.01 - 1.09 ms per
    // 12.5 M reads: each tail idled most of the device.)
    PlanesCall pcs[LINKED_MAX];
    hipStream_t gstream[LINKED_MAX] = {st, st, st, st};
    uint8_t *gwork[LINKED_MAX];
    for (int g = 0; g < s->p.n; ++g) {
        gwork[g] = wbase;
        wbase += (fast_work_bytes(info[g] > 64 ? info[g] : 64) + 4095) & ~(size_t)4095;
    }
    int live_seen = 0;
    for (int g = 0; g < s->p.n; ++g) {
        const long long n_g = info[g], tile0 = info[LINKED_MAX + g];
        if (n_g == 0) continue;
        const int slot_g = live_seen % nstreams;
        ++live_seen;
        if (side_by_side && slot_g != 0) {
            if (!side[slot_g].ready()) return ATR_ERR_HIP;
            if (!waited[slot_g]) {
                e = hipStreamWaitEvent(side[slot_g].stream, side[0].fork, 0);
                if (e != hipSuccess) return hip_fail(e, "linked group fork");
                waited[slot_g] = true;
            }
            gstream[g] = side[slot_g].stream;
        }
        const int rc = launch_planes_prepass(&s->back[g], (const uint4 *)d_grouped + (size_t)tile0 * nchunks * 64, d_glens + tile0 * 64, n_g,
                                             max_len, (uint4 *)d_slab + tile0 * 64, gwork[g], gstream[g],
                                             side_by_side ? std::max(1, std::min(std::min(live_groups, nstreams), grid_div)) : 1, &pcs[g]);
        if (rc != 0) return hip_fail((hipError_t)rc, "linked group 3' pre-pass launch");
    }
    for (int g = 0; g < s->p.n; ++g) {
        const long long n_g = info[g], tile0 = info[LINKED_MAX + g];
        if (n_g == 0) continue;
        const int rc = launch_planes_tail(&s->back[g], (const uint4 *)d_grouped + (size_t)tile0 * nchunks * 64, d_glens + tile0 * 64, n_g,
                                          max_len, (uint4 *)d_slab + tile0 * 64, pcs[g], gstream[g], side_by_side && one_stream);
        if (rc != 0) return hip_fail((hipError_t)rc, "linked group 3' DP launch");
    }
    for (int k = 1; k < LINKED_MAX; ++k)
        if (waited[k]) {
            e = hipEventRecord(side[k].join, side[k].stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, side[k].join, 0);
            if (e != hipSuccess) return hip_fail(e, "linked group join");
        }
    if (d_back) {
        const unsigned blocks = (unsigned)std::min<long long>((nreads + 255) / 256, 8192);
        hipLaunchKernelGGL(linked_group_finish_kernel, dim3(blocks), dim3(256), 0, st, (const LinkedWaveBlob *)s->d_wave,
                           (const uint4 *)d_slab, d_slot_of, (const uint16_t *)d_which, (long long)nreads, (uint4 *)d_back);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "linked_group_finish_kernel launch");
    }
    return ATR_OK;
}

}  // extern "C"
