// linked_blob.hpp -- device-side blocks of a linked set and the 5' stage's task drain, shared by linked_kernels.hip (the
// fused pipeline on tile64 reads) and linked_group.hip (round 6: 5' parts decided at pack time, adapter-uniform tiles).
#ifndef ATR_LINKED_BLOB_HPP
#define ATR_LINKED_BLOB_HPP

#include "linked_core.hpp"
#include "locate_fast.hpp"

namespace atr {

struct LinkedBlob {
    LinkedParams p;
    LinkedRmp rmp;
};

// The 3' aligners as the wavefront-per-read kernel, the band launch and the finish pass want them.
struct LinkedWaveBlob {
    LocateParams p[LINKED_MAX];
    LinkedPost post[LINKED_MAX];
    BandParams bp[LINKED_MAX];                      // linked_band_kernel
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool wave_any(bool v) { return __ballot(v) != 0ull; }

constexpr int LINKED_ROUND = 8;                      // tiles a wave takes through the 5' stage before their 3' stage
constexpr int LINKED_TASKS = 512;                    // capacity of a wave's queue of (read, 5' adapter) DP tasks

// The wave's queued DP tasks, 64 at a time, every lane against its own (read, adapter): the read's first chunk is
// fetched again (load(tile, lane in tile, n, w0): from the packed batch or the ASCII rows -- L2), the banded DP runs
// once per DP group that has a task in the pass, results are merged into the per-read words (smallest adapter index
// wins) and counters.
template <bool RAGGED, bool AND_MODE, class Load>
__device__ __forceinline__ void linked_drain_with(const LinkedBlob &S, int ngroups, const uint16_t *queue, int ntasks,
                                                  const int32_t *__restrict__ lens, long long tile_first, int max_len,
                                                  uint32_t *s_word, uint32_t *s_count, uint32_t *ns, int lane, Load load) {
    for (int base = 0; base < ntasks; base += 64) {                           // wave-uniform
        const bool valid = base + lane < ntasks;
        const uint32_t task = valid ? (uint32_t)queue[base + lane] : 0u;      // [14:6] slot * 64 + lane  [1:0] adapter
        const int a_l = (int)(task & 3u), cell = (int)(task >> 6);
        const long long tile = tile_first + 4 * (cell >> 6);
        const long long r = tile * 64 + (cell & 63);
        const int n = valid ? (RAGGED ? lens[r] : max_len) : 0;
        uint32_t w0[4] = {0u, 0u, 0u, 0u};
        if (valid) load(tile, cell & 63, n, w0);
        const FrontParams &mp = S.p.f[a_l];
        const int grp = valid ? mp.group : -1;
        for (int g = 0; g < ngroups; ++g) {
            if (!wave_any(grp == g)) continue;
            const FrontParams &gp = S.p.f[rfl(S.p.group_first[g])];
            const Uniform u = front_uniform(rfl(gp.m), rfl(gp.k), rfl(gp.indel), rfl(gp.min_overlap));
            front_stage(w0, u.k, ns, 64);
            uint32_t rec[4];
            const uint32_t *rr = (grp == g ? mp : gp).rrep;
            band_locate_prefix_rr<AND_MODE>(u, [rr](int i) { return rr[i - 1]; }, rfl(gp.noindel) != 0, ns, 64, n, gp.thr, rec);
            if (grp == g && front_accept(rec, u.m, u.min_overlap, mp.pf_thr, mp.accept_full != 0, S.rmp.front[a_l],
                                         S.rmp.front_ld[a_l], S.rmp.front_max[a_l])) {
                atomicMin(&s_word[cell], front_word_of(a_l, rec));
                atomicAdd(&s_count[cell >> 2], 1u << (8 * (cell & 3)));
            }
        }
    }
}

}  // namespace atr
#endif
