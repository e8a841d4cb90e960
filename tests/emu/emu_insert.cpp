// emu_insert.cpp -- TEST INFRASTRUCTURE.  CPU emulation of the gfx950 insert-match
// kernel (atropos_amd/csrc/insert_kernel.hip) from the same per-lane source
// (insert_core.hpp, -DATR_HOST_EMU) and the same host parameter builder
// (insert_host.hpp).  Lanes are independent in this kernel, so each pair is simply run
// through the block sweep with jmax = the tile's maximum overlap length, as on the GPU.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "insert_host.hpp"

using namespace atr;

namespace {

template <int NCH, class IP>
void emu_insert_tiles_ip(const IP &ip, const uint32_t *p1, const int32_t *l1, const uint32_t *p2,
                         const int32_t *l2, long long npairs, int max_len, uint32_t *out) {
    constexpr int W = NCH;
    const long long ntiles = (npairs + 63) / 64;
    for (long long tile = 0; tile < ntiles; ++tile) {
        int jmax = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const long long r = tile * 64 + lane;
            if (r < npairs) jmax = std::max(jmax, std::min(l1 ? l1[r] : max_len, l2 ? l2[r] : max_len));
        }
        for (int lane = 0; lane < 64; ++lane) {
            const long long r = tile * 64 + lane;
            if (r >= npairs) continue;
            uint32_t b1[4 * W], b2[4 * W];
            for (int c = 0; c < NCH; ++c)
                for (int d = 0; d < 4; ++d) {
                    b1[4 * c + d] = p1[(((size_t)tile * NCH + c) * 64 + lane) * 4 + d];
                    b2[4 * c + d] = p2[(((size_t)tile * NCH + c) * 64 + lane) * 4 + d];
                }
            PairState<W> P;
            pair_init<W>(P, ip, l1 ? l1[r] : max_len, l2 ? l2[r] : max_len, b1, b2);
            const uint32_t *g1 = p1 + (((size_t)tile * NCH) * 64 + lane) * 4, *g2 = p2 + (((size_t)tile * NCH) * 64 + lane) * 4;
            uint32_t rec_probed[12];
            {   // as the kernel: the probed two-pass sweep (read 2's planes and the list of overlap lengths in "LDS")
                uint32_t rl[4 * W];
                uint16_t cl[INS_LIST_CAP];
                planes_to_lds<W>(P, rl, 1);
                sweep_probed<W>(P, ip, jmax, g1, g2, 64 * 4, rl, 1, cl, 1, ip.thr_hit, [](int n) { return n; });
                if (!unordered_is_exact<W>(P)) { planes_from_lds<W>(P, rl, 1); sweep_ordered<W>(P, ip, jmax); }
                pair_result<W>(P, ip, rec_probed);
            }
            // and the one-pass unordered sweep it replaced: the same records, or the emulation aborts
            sweep_unordered<W>(P, ip, jmax, g1, g2, 64 * 4);
            if (!unordered_is_exact<W>(P)) sweep_ordered<W>(P, ip, jmax);
            pair_result<W>(P, ip, out + 12 * r);
            if (memcmp(rec_probed, out + 12 * r, sizeof(rec_probed)) != 0) abort();
        }
    }
}

// as the launcher: adapters of more than 64 bases take the code built with InsertParamsLong
template <int NCH>
void emu_insert_tiles(const atr_insert_aligner *a, const uint32_t *p1, const int32_t *l1, const uint32_t *p2,
                      const int32_t *l2, long long npairs, int max_len, uint32_t *out, int cased) {
    if (cased && a->p.long_adapters) emu_insert_tiles_ip<NCH>(static_cast<const InsertParamsLongCased &>(a->p), p1, l1, p2, l2, npairs, max_len, out);
    else if (cased) emu_insert_tiles_ip<NCH>(static_cast<const InsertParamsCased &>(a->p), p1, l1, p2, l2, npairs, max_len, out);
    else if (a->p.long_adapters) emu_insert_tiles_ip<NCH>(static_cast<const InsertParamsLong &>(a->p), p1, l1, p2, l2, npairs, max_len, out);
    else emu_insert_tiles_ip<NCH>(a->p, p1, l1, p2, l2, npairs, max_len, out);
}

}  // namespace

extern "C" {

int emu_insert_aligner_create(const atr_insert_config *cfg, atr_insert_aligner **out) {
    *out = nullptr;
    atr_insert_aligner *h = new atr_insert_aligner();
    h->d_tables = nullptr;
    int rc = insert_fill(h, cfg);
    if (rc != ATR_OK) { delete h; return rc; }
    h->p.rmp_insert = h->rmp_insert.data();
    h->p.rmp_adapter = h->rmp_adapter.data();
    *out = h;
    return ATR_OK;
}

void emu_insert_aligner_destroy(atr_insert_aligner *a) { delete a; }

int emu_case_sensitive_table(uint8_t table[256]) {
    case_sensitive_table(tables().dna15, table);
    return ATR_OK;
}

int emu_insert_match_batch(const atr_insert_aligner *a, const uint8_t *p1, const int32_t *l1, const uint8_t *p2,
                           const int32_t *l2, int64_t npairs, int max_len, int cased, int16_t *out) {
    if (!a || npairs < 0 || max_len < 0) return ATR_ERR_INVALID;
    if (cased && !a->cased_ok) return ATR_ERR_UNSUPPORTED;
    if (max_len > ATR_INSERT_MAX_READ) return ATR_ERR_UNSUPPORTED;
    if (npairs == 0) return ATR_OK;
    const uint32_t *a1 = (const uint32_t *)p1, *a2 = (const uint32_t *)p2;
    uint32_t *o = (uint32_t *)out;
    switch ((max_len + 31) / 32) {
        case 0: case 1: emu_insert_tiles<1>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 2: emu_insert_tiles<2>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 3: emu_insert_tiles<3>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 4: emu_insert_tiles<4>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 5: emu_insert_tiles<5>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 6: emu_insert_tiles<6>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 7: emu_insert_tiles<7>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 8: emu_insert_tiles<8>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        case 9: emu_insert_tiles<9>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
        default: emu_insert_tiles<10>(a, a1, l1, a2, l2, npairs, max_len, o, cased); break;
    }
    return ATR_OK;
}

}  // extern "C"
