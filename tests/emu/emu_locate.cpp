// emu_locate.cpp -- TEST INFRASTRUCTURE.  A lock-step CPU emulation of the gfx950
// locate kernel's control flow (atropos_amd/csrc/locate_kernel.hpp) built from the
// SAME per-lane source (locate_core.hpp, compiled with -DATR_HOST_EMU) and the same
// host parameter derivation (aligner_host.hpp).  It lets the CPU test-suite check the
// kernel arithmetic -- packed cell words, v_min3 tie-breaking, end-aligned rows,
// wave-uniform sweep with per-lane windows -- against the oracle without a GPU.
// It is never loaded by the product package.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "aligner_host.hpp"
#include "filter_core.hpp"
#include "piece_core.hpp"
#include "linked_host.hpp"
#include "wave_core.hpp"

using namespace atr;

namespace {

template <int MT, bool NOINDEL, bool XREP>
void emu_tiles(const atr_aligner *a, const uint32_t *packed, const int32_t *lens, long long nreads,
               int nchunks, int max_len, uint32_t *out) {
    const LocateParams &p = a->p;
    const Uniform u = make_uniform(p, MT);
    int16_t s_thr[ATR_MAX_REF_LEN + 2];
    uint32_t s_init[ATR_MAX_REF_LEN + 1];
    for (int i = 0; i <= MT + 1; ++i) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    const long long ntiles = (nreads + 63) / 64;
    std::vector<LaneState<MT>> L(64);
    for (long long tile = 0; tile < ntiles; ++tile) {
        int jlo = 0x7fffffff, jhi = 0;
        bool live[64];
        for (int lane = 0; lane < 64; ++lane) {
            const long long r = tile * 64 + lane;
            live[lane] = r < nreads;
            const int n = live[lane] ? (lens ? lens[r] : max_len) : 0;
            lane_init<MT, NOINDEL, XREP>(L[lane], u, n, s_init, s_thr);
            const bool has_window = live[lane] && L[lane].max_n > L[lane].min_n;
            jlo = std::min(jlo, has_window ? L[lane].min_n : 0x7fffffff);
            jhi = std::max(jhi, has_window ? L[lane].max_n : 0);
        }
        // reads of more than ATR_MAX_READ_LEN bases: locate_long_kernel's rolling origin base (locate_kernel.hpp)
        const bool lng = max_len > ATR_MAX_READ_LEN;
        int obase = 0, best_base[64] = {0};
        if (jhi > jlo) {
            if (lng) {
                obase = long_base(jlo + 1);
                if (!XREP)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i <= MT; ++i) L[lane].col[i] -= (uint32_t)obase;
            }
            const int c0 = jlo >> 5, c1 = (jhi + 31) >> 5;
            for (int c = c0; c < c1; ++c) {
                if (lng && long_base(c * 32 + 1) != obase) {
                    for (int lane = 0; lane < 64; ++lane) lane_rebase<MT>(L[lane]);
                    obase += LONG_BASE_STEP;
                }
                for (int d = 0; d < 4; ++d) {
                    for (int b = 0; b < 8; ++b) {
                        const int j = c * 32 + d * 8 + b + 1;
                        if (j <= jlo || j > jhi) continue;
                        for (int lane = 0; lane < 64; ++lane) {   // every lane, unmasked, like the wave
                            const uint32_t w = packed[(((size_t)tile * nchunks + c) * 64 + lane) * 4 + d];
                            const uint32_t q = (w >> (4 * b)) & 15u;
                            uint32_t nm[(MT + 31) / 32];
                            for (int w2 = 0; w2 < (MT + 31) / 32; ++w2) nm[w2] = p.nmask[q][w2];
                            const int key = L[lane].best.key;
                            lane_step<MT, NOINDEL, XREP>(L[lane], u, j, nm, s_thr, MT, obase);
                            if (L[lane].best.key != key) best_base[lane] = obase;
                        }
                    }
                }
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            if (!live[lane]) continue;
            lane_result<MT>(L[lane], u, out + (tile * 64 + lane) * 4, best_base[lane]);
        }
    }
}

// The filtered pipeline (locate_fast.hpp): K1 per read, K2/K3 as a stable bucket order by
// window-start bin, K4 in waves of 64 gathered reads, lock-step like the GPU wave.
static uint32_t read_code(const uint32_t *packed, int nchunks, long long r, int j /*1-based*/) {
    const long long tile = r >> 6;
    const int lane = (int)(r & 63), c = (j - 1) >> 5, d = ((j - 1) >> 3) & 3, b = (j - 1) & 7;
    return (packed[(((size_t)tile * nchunks + c) * 64 + lane) * 4 + d] >> (4 * b)) & 15u;
}

// What the LINKED variants of the band / window kernels get on top (locate_fast.hpp, LinkedArgs)
struct EmuLinked {
    const uint32_t *front;                                         // 5' records, 4 dwords per read
    const LinkedPost *post;
};

// K2..K4 of the filtered pipeline for ONE aligner: `bins` = the FILTER_BINS scatter bins holding the
// unresolved reads (K2 + K3 as a stable bucket order), K4a / K4 in waves of 64 gathered reads,
// lock-step like the GPU wave.  la != nullptr: the 3' part of one adapter of a linked set.
template <int MT, bool NOINDEL>
void emu_dp_stage(const atr_aligner *a, const uint32_t *packed, const int32_t *lens, int nchunks, int max_len,
                  uint32_t *out, const std::vector<uint32_t> &win, const std::vector<std::vector<uint32_t>> &bins,
                  const EmuLinked *la) {
    const LocateParams &p = a->p;
    const Uniform u = make_uniform(p, MT);
    int16_t s_thr[ATR_MAX_REF_LEN + 2];
    uint32_t s_init[ATR_MAX_REF_LEN + 1];
    for (int i = 0; i <= MT + 1; ++i) {
        if (i <= u.m + 1) s_thr[i] = p.thr[i];
        if (i <= MT) s_init[i] = init_word(i - u.p0, 0, u.sr, u.sq, u.indel);
    }
    auto finish = [&](long long r) {
        if (!la) return;
        const LinkedPost &q = *la->post;
        linked_finish(out + 4 * r, (int)(la->front[4 * r + 1] >> 16), q.m, q.min_overlap, q.pf_thr, q.accept_full != 0,
                      q.rmp, q.rmp_ld, q.max_rmp);
    };
    std::vector<uint32_t> order;                                   // K2 + K3
    for (auto &b : bins) order.insert(order.end(), b.begin(), b.end());
    const long long total = (long long)order.size();
    long long nband = 0;                                           // band reads come first (bins [0, BAND_BINS))
    for (int b = 0; b < BAND_BINS; ++b) nband += (long long)bins[b].size();
    {                                                              // K4a: banded DP, one wave at a time
        uint32_t codes[FILTER_MAX_M] = {0};
        for (int i = 0; i < u.m && i < FILTER_MAX_M; ++i) codes[i] = (uint32_t)(a->codes[i] & 15u) * 0x11111111u;
        const bool and_mode = a->wildcard_ref || a->wildcard_query, noindel = a->indel_cost > p.k;
        for (long long base = 0; base < nband; base += 64) {
            int smax = 0, smax_l = 0, rows_max = 0, cap_lo = 0x7fffffff;        // the wave-level values of band_kernel
            for (int lane = 0; lane < 64 && base + lane < nband; ++lane) {
                const uint32_t ww = win[order[base + lane]];
                if (window_scan(ww)) {
                    smax_l = std::max(smax_l, last_band_width(ww));
                    rows_max = std::max(rows_max, last_band_rowm(ww) ? u.m : window_rows(ww));
                    cap_lo = std::min(cap_lo, window_rows(ww) - last_band_span(ww));
                } else {
                    smax = std::max(smax, window_hi(ww) - u.m + u.k - window_lo(ww));
                }
            }
            smax = std::min(smax, BAND_W - 1);
            smax_l = std::min(smax_l, BAND_W - 1);
            for (int lane = 0; lane < 64 && base + lane < nband; ++lane) {
                const long long r = order[base + lane];
                const uint32_t *q = packed + (((size_t)(r >> 6) * nchunks) * 64 + (r & 63)) * 4;
                const int n = lens ? lens[r] : max_len;
                uint32_t ns[BAND_STREAM];
                band_stage(q, nchunks, window_lo(win[r]), ns, 1, band_stream_dwords(u.m));
                if (window_scan(win[r])) {
                    if (and_mode) band_locate_last<true>(u, codes, noindel, ns, 1, n, win[r], true, smax_l, rows_max, cap_lo, s_thr, out + 4 * r);
                    else band_locate_last<false>(u, codes, noindel, ns, 1, n, win[r], true, smax_l, rows_max, cap_lo, s_thr, out + 4 * r);
                } else if (and_mode) band_locate<true>(u, codes, noindel, ns, 1, n, win[r], smax, s_thr, out + 4 * r);
                else band_locate<false>(u, codes, noindel, ns, 1, n, win[r], smax, s_thr, out + 4 * r);
                finish(r);
            }
        }
    }
    std::vector<LaneState<MT>> L(64);
    long long rows_bin0 = 0;                                       // binbase[ROWS_BIN0]
    for (int b = 0; b < ROWS_BIN0; ++b) rows_bin0 += (long long)bins[b].size();
    for (long long base = nband; base < total; base += 64) {       // K4, one wave at a time
        int jlo = 0x7fffffff, jhi = 0, rows = 0, s_top = 0;
        long long rr[64];
        bool live[64];
        uint32_t wws[64];
        int s_lane[64];
        for (int lane = 0; lane < 64; ++lane) {
            live[lane] = base + lane < total;
            rr[lane] = live[lane] ? order[base + lane] : 0;
            wws[lane] = live[lane] ? win[rr[lane]] : 0u;
            const int lo = window_lo(wws[lane]), hi = live[lane] ? window_hi(wws[lane]) : 0;
            jlo = std::min(jlo, live[lane] ? lo : 0x7fffffff);
            jhi = std::max(jhi, (live[lane] && hi > lo) ? hi : 0);
            rows = std::max(rows, live[lane] ? window_rows(wws[lane]) : 0);
            s_lane[lane] = (la && live[lane]) ? (int)(la->front[4 * rr[lane] + 1] >> 16) : 0;
            s_top = std::max(s_top, s_lane[lane]);
        }
        const int plimit = u.p0 + rows;
        if (!la && lens != nullptr && base >= rows_bin0 && !u.sr) {        // tail mode (window_kernel)
            int v0 = 0x7fffffff, s_top_t = 0, shift[64];
            for (int lane = 0; lane < 64; ++lane) {
                shift[lane] = live[lane] ? max_len - lens[rr[lane]] : 0;
                if (live[lane]) v0 = std::min(v0, window_lo(wws[lane]) + shift[lane]);
                s_top_t = std::max(s_top_t, shift[lane]);
            }
            if (max_len - v0 <= TAIL_COLUMNS && v0 < max_len) {
                for (int lane = 0; lane < 64; ++lane)
                    lane_init_window<MT, NOINDEL>(L[lane], u, max_len, v0, live[lane] ? max_len : 0,
                                                  live[lane] && window_scan(wws[lane]), s_init, s_thr);
                for (int j = max_len - TAIL_COLUMNS + 1; j <= max_len; ++j) {
                    if (j <= v0) continue;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int own = j - shift[lane];                   // the lane's own column
                        const uint32_t q = (live[lane] && own >= 1) ? read_code(packed, nchunks, rr[lane], own) : 0u;
                        uint32_t nm[(MT + 31) / 32];
                        for (int w2 = 0; w2 < (MT + 31) / 32; ++w2) nm[w2] = p.nmask[q][w2];
                        int pl = std::min(plimit, u.p0 + (j - v0) + u.k);
                        pl = std::min(pl, u.p0 + triangle_rows(rows, max_len, j, u.k));
                        lane_step<MT, NOINDEL, true, true>(L[lane], u, j, nm, s_thr, pl);
                        if (j <= s_top_t && shift[lane] == j) lane_restart_window<MT>(L[lane], u, j);
                    }
                }
                for (int lane = 0; lane < 64; ++lane)
                    if (live[lane]) {
                        uint32_t *rec = out + 4 * rr[lane];
                        lane_result<MT>(L[lane], u, rec);
                        if ((rec[0] >> 16) != 0xFFFFu) rec[1] -= (uint32_t)shift[lane] * 0x00010001u;
                    }
                continue;
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int n = live[lane] ? (lens ? lens[rr[lane]] : max_len) : 0;
            lane_init_window<MT, NOINDEL>(L[lane], u, n, jlo, live[lane] ? window_hi(wws[lane]) : 0,
                                 live[lane] && window_scan(wws[lane]), s_init, s_thr);
        }
        const bool head = !(u.sr && jlo == 0);
        const bool tri = base >= rows_bin0 && head && lens == nullptr;
        if (jhi > jlo)
        for (int j = jlo + 1; j <= jhi; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                // lanes past the end of `order` gather read 0 like the kernel does
                const uint32_t q = (j <= nchunks * 32) ? read_code(packed, nchunks, rr[lane], j) : 0u;
                uint32_t nm[(MT + 31) / 32];
                for (int w2 = 0; w2 < (MT + 31) / 32; ++w2) nm[w2] = p.nmask[q][w2];
                int pl = head ? std::min(plimit, u.p0 + (j - jlo) + u.k) : plimit;
                if (tri) pl = std::min(pl, u.p0 + triangle_rows(rows, max_len, j, u.k));
                lane_step<MT, NOINDEL, true, true>(L[lane], u, j, nm, s_thr, pl);
                if (la && j <= s_top && s_lane[lane] == j) lane_restart_window<MT>(L[lane], u, j);
            }
        for (int lane = 0; lane < 64; ++lane)
            if (live[lane]) { lane_result<MT>(L[lane], u, out + 4 * rr[lane]); finish(rr[lane]); }
    }
}

// The filtered pipeline (locate_fast.hpp): K1 per read, then emu_dp_stage.
template <int MT, bool NOINDEL>
void emu_fast(const atr_aligner *a, const uint32_t *packed, const int32_t *lens, long long nreads, int nchunks,
              int max_len, uint32_t *out) {
    const LocateParams &p = a->p;
    const Uniform u = make_uniform(p, MT);
    std::vector<uint32_t> win((size_t)nreads);
    std::vector<std::vector<uint32_t>> bins(FILTER_BINS);
    const FilterParams fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap);
    for (long long r = 0; r < nreads; ++r) {                       // K1
        const int n = lens ? lens[r] : max_len;
        FilterState F;
        filter_init(F, u, fp.rows);
        for (int j = 1; j <= n; ++j) {
            const uint64_t eq = fp.peq[read_code(packed, nchunks, r, j)];
            if (fp.rows > 32) filter_step<true>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k);
            else filter_step<false>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k);
            if ((j & 31) == 0 || j == n) filter_fold(F, j, fp.rows, (uint32_t)u.k);        // as the kernel: once per 32-column chunk
        }
        uint32_t rec[4];
        const uint32_t *q = packed + (((size_t)(r >> 6) * nchunks) * 64 + (r & 63)) * 4;
        const uint32_t ww = fp.rows > 32 ? filter_decide<true>(F, u, fp, q, nchunks, n, rec) : filter_decide<false>(F, u, fp, q, nchunks, n, rec);
        win[r] = ww;
        if (!window_valid(ww)) memcpy(out + 4 * r, rec, 16);
        else bins[window_bin(ww, u.m, lens == nullptr || ragged_rows_bins(u.sr))].push_back((uint32_t)r);
    }
    emu_dp_stage<MT, NOINDEL>(a, packed, lens, nchunks, max_len, out, win, bins, nullptr);
}

// The two-pass pre-pass on plane64 reads (piece_kernels.hip): pass A per read (piece_scan, the product's own
// source), pass B as a windowed sweep with the product's filter_step / filter_decide_tm, then emu_dp_stage on the
// reads it leaves open (from a tile64 copy of the planes).  The window length varies with the read number the way
// it varies with a wave's other lanes on the GPU: the records must not depend on it.
static std::vector<uint32_t> g_piece_last_windows;   // the window words of the last two-pass call (0: resolved in the pre-pass)
static long long g_piece_stats[6], g_piece_need[12];   // need histogram, 8 columns per bin      // reads, flagged, read-end condition, wide, window columns swept, open after pass B
// nr >= 0: a ragged batch -- pass A sees the read moved to the end of its NW words (n = 32 NW), as the kernel does
template <int NW>
static uint32_t emu_piece_read(const Uniform &u, const FilterParams &fp, const PieceParams &pp, const uint32_t *planes,
                               const uint32_t *nib, long long r, int n, int nr, uint32_t rec[4]) {
    uint32_t pl[NW][4], twp[4];
    for (int w = 0; w < NW; ++w)
        for (int q = 0; q < 4; ++q) pl[w][q] = planes[((((size_t)(r >> 6) * NW) + w) * 64 + (r & 63)) * 4 + q];
    const int back = nr >= 0 ? 32 * NW - nr : 0;
    constexpr int HWN = NW < PIECE_HEAD_WORDS ? NW : PIECE_HEAD_WORDS;
    uint32_t hpl[HWN][4];
    for (int w = 0; w < HWN; ++w)
        for (int q = 0; q < 4; ++q) hpl[w][q] = pl[w][q];
    if (nr >= 0) {
        uint32_t mv[NW][4];
        for (int w = 0; w < NW; ++w)
            for (int q = 0; q < 4; ++q) {
                uint32_t v = 0u;
                for (int b = 0; b < 32; ++b) {
                    const int src = 32 * w + b - back;
                    if (src >= 0 && ((pl[src >> 5][q] >> (src & 31)) & 1u)) v |= 1u << b;
                }
                mv[w][q] = v;
            }
        memcpy(pl, mv, sizeof(pl));
        n = 32 * NW;
    } else nr = n;
    for (int q = 0; q < 4; ++q) {
        const int sh = n & 31;
        twp[q] = sh == 0 ? pl[NW - 1][q] : piece_funnel(pl[NW - 1][q], NW >= 2 ? pl[NW >= 2 ? NW - 2 : 0][q] : 0u, sh);
    }
    const int mf = fp.rows, T = u.m - mf;
    const PieceScan S = piece_scan<NW>(pp, pl, twp, n, mf, T, u.k, hpl);
    const PieceTask pt = piece_task(S, back, nr, u.sr, u.m, u.k, pp.head_cols);
    rec[0] = 0xFFFF0000u; rec[1] = rec[2] = rec[3] = 0u;
    ++g_piece_stats[0];
    if (S.j_exact != 0 && u.m >= u.min_overlap) {          // the adapter verbatim: the reference's early exit
        const int j = S.j_exact - back;
        rec[0] = (uint32_t)u.m << 16; rec[1] = (uint32_t)(j - u.m) | ((uint32_t)j << 16); rec[2] = (uint32_t)u.m;
        return 0u;
    }
    if (!pt.flagged) return 0u;
    if (pp.aonly) {                                        // pass A only (piece_filter.hpp): the window DP on pass A's columns
        const int wlo = pt.full ? 0 : pt.j_e - pt.need, whi = pt.full ? nr : pt.j_e;
        return window_word(wlo, whi, whi == nr, u.m, false);
    }
    const int need = pt.need;
    const int j_e = pt.j_e;                                // from here on: the read's own columns
    n = nr;
    ++g_piece_stats[1];
    if (S.tail) ++g_piece_stats[2];
    ++g_piece_need[std::min(11, (need + 7) / 8)];
    if (need > pp.narrow) ++g_piece_stats[3]; else g_piece_stats[4] += (need + 7) & ~7;
    const uint32_t *q = nib + (((size_t)(r >> 6) * NW) * 64 + (r & 63)) * 4;
    const auto tm = [&](int jp) { return filter_tail_matches(fp, T, q, NW, jp); };
    // the adapter's rows against the read on diagonal d (the substitution certificate of filter_decide_tm)
    const auto dg = [&](int d) -> uint64_t {
        uint64_t mm = 0ull;
        for (int row = 1; row <= u.m && row <= 64; ++row) {
            const int j = d + row;
            const uint32_t code = (j >= 1 && j <= n) ? read_code(nib, NW, r, j) : 0u;
            int c = -1;
            for (int t = 0; t < 4; ++t) if ((fp.rowsel[t] >> (row - 1)) & 1ull) c = t;
            if (c < 0 || code != (1u << c)) mm |= 1ull << (row - 1);
        }
        return mm;
    };
    FilterState F;
    filter_init(F, u, mf);
    if (need <= pp.narrow && !pt.full) {
        const int PW = pp.window;
        const int W = std::min(PW, ((need + 7) & ~7) + 8 * (int)(r % 3));
        const bool zero0 = u.sr && j_e - W <= 0;           // START_WITHIN_SEQ1: the sweep reaches back to column 0 (piece_filter.hpp)
        if (u.sr && !zero0) { Uniform uf = u; uf.sr = false; filter_init(F, uf, mf); }
        for (int rc = PW - W + 1; rc <= PW; ++rc) {
            const int j = j_e - PW + rc;
            const uint64_t eq = (zero0 && j < 1) ? ~0ull : fp.peq[j >= 1 ? read_code(nib, NW, r, j) : 0u];
            if (mf > 32) filter_step<true>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k); else filter_step<false>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k);
            if ((rc & 31) == 0 && rc < PW) filter_fold(F, j, mf, (uint32_t)u.k);
        }
        filter_fold(F, j_e, mf, (uint32_t)u.k);
        const int pa_dlo = j_e - need, pa_dhi = j_e < n ? j_e - u.m : -0x10000;
        if (pp.window > PIECE_WINDOW) return mf > 32 ? filter_decide_tm<true>(F, u, fp, tm, n, rec, 0, j_e == n, FilterNoDiag(), pa_dlo, pa_dhi) : filter_decide_tm<false>(F, u, fp, tm, n, rec, 0, j_e == n, FilterNoDiag(), pa_dlo, pa_dhi);   // (no diagonal view)
        return mf > 32 ? filter_decide_tm<true>(F, u, fp, tm, n, rec, 0, j_e == n, dg, pa_dlo, pa_dhi) : filter_decide_tm<false>(F, u, fp, tm, n, rec, 0, j_e == n, dg, pa_dlo, pa_dhi);
    }
    for (int j = 1; j <= n; ++j) {
        const uint64_t eq = fp.peq[read_code(nib, NW, r, j)];
        if (mf > 32) filter_step<true>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k); else filter_step<false>(F, (uint32_t)eq, (uint32_t)(eq >> 32), (uint32_t)u.k);
        if ((j & 31) == 0 || j == n) filter_fold(F, j, mf, (uint32_t)u.k);
    }
    return mf > 32 ? filter_decide_tm<true>(F, u, fp, tm, n, rec, 0, true, dg) : filter_decide_tm<false>(F, u, fp, tm, n, rec, 0, true, dg);
}

template <int MT, bool NOINDEL>
void emu_piece(const atr_aligner *a, const uint32_t *planes, const int32_t *lens, long long nreads, int nchunks,
               int max_len, uint32_t *out) {
    const LocateParams &p = a->p;
    const Uniform u = make_uniform(p, MT);
    const FilterParams fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap, true);
    PieceParams pp;
    if (!piece_params(a->codes, a->p.m, fp.rows, a->p.k, a->flags, a->wildcard_ref || a->wildcard_query,
                      a->table_kind == ATR_TABLE_CUSTOM, fp.thr_row, lens ? 32 * nchunks : max_len, pp, a->p.thr, a->p.min_overlap)) abort();
    uint32_t spread[4][256];
    piece_spread_tables(spread);
    const long long ntiles = (nreads + 63) / 64;
    std::vector<uint32_t> nib((size_t)ntiles * nchunks * 64 * 4);               // the tile64 twin of the planes
    for (size_t ch = 0; ch < (size_t)ntiles * nchunks * 64; ++ch)
        for (int d = 0; d < 4; ++d)
            nib[ch * 4 + d] = piece_nibbles(spread, planes[ch * 4], planes[ch * 4 + 1], planes[ch * 4 + 2], planes[ch * 4 + 3], d);
    std::vector<uint32_t> win((size_t)nreads);
    std::vector<std::vector<uint32_t>> bins(FILTER_BINS);
    for (long long r = 0; r < nreads; ++r) {
        uint32_t rec[4], ww = 0;
        const int nr = lens ? std::min(std::max(lens[r], 0), max_len) : -1;
        switch (nchunks) {
            case 1: ww = emu_piece_read<1>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 2: ww = emu_piece_read<2>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 3: ww = emu_piece_read<3>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 4: ww = emu_piece_read<4>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 5: ww = emu_piece_read<5>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 6: ww = emu_piece_read<6>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 7: ww = emu_piece_read<7>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 8: ww = emu_piece_read<8>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            case 9: ww = emu_piece_read<9>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
            default: ww = emu_piece_read<10>(u, fp, pp, planes, nib.data(), r, max_len, nr, rec); break;
        }
        win[r] = ww;
        if (!window_valid(ww)) memcpy(out + 4 * r, rec, 16);
        else { bins[window_bin(ww, u.m, true)].push_back((uint32_t)r); ++g_piece_stats[5]; }
    }
    g_piece_last_windows = win;                                  // (diagnostics: tools/c2_unresolved.py)
    emu_dp_stage<MT, NOINDEL>(a, nib.data(), lens, nchunks, max_len, out, win, bins, nullptr);
}

typedef void (*emu_fn)(const atr_aligner *, const uint32_t *, const int32_t *, long long, int, int, uint32_t *);

template <int MT>
emu_fn pick(bool xrep, bool noindel, bool fast) {
    if (fast) return noindel ? &emu_fast<MT, true> : &emu_fast<MT, false>;
    if (xrep) return noindel ? &emu_tiles<MT, true, true> : &emu_tiles<MT, false, true>;
    return noindel ? &emu_tiles<MT, true, false> : &emu_tiles<MT, false, false>;
}

template <int... I>
emu_fn pick_mt(int idx, bool eqmode, bool noindel, bool fast, std::integer_sequence<int, I...>) {
    emu_fn fns[] = {pick<(I + 1) * ROW_GRAN>(eqmode, noindel, fast)...};
    return fns[idx];
}

}  // namespace

template <int... I>
static emu_fn pick_piece(int idx, bool noindel, std::integer_sequence<int, I...>) {
    emu_fn fn = nullptr;
    (void)std::initializer_list<int>{(idx == I ? (fn = noindel ? &emu_piece<ROW_GRAN *(I + 1), true> : &emu_piece<ROW_GRAN *(I + 1), false>, 0) : 0)...};
    return fn;
}

extern "C" {

int emu_aligner_create(const char *ref, int m, double e, int flags, int wr, int wq, int min_overlap,
                       int indel_cost, atr_aligner **out) {
    return aligner_create(ref, m, e, flags, wr, wq, min_overlap, indel_cost, out);
}
void emu_aligner_destroy(atr_aligner *a) { delete a; }
int emu_aligner_set_min_overlap(atr_aligner *a, int v) { return aligner_set_min_overlap(a, v); }
int emu_aligner_set_indel_cost(atr_aligner *a, int v) { return aligner_set_indel_cost(a, v); }
int emu_aligner_query_table(const atr_aligner *a, uint8_t table[256]) {
    if (table) memcpy(table, a->qtable, 256);
    return a->table_kind;
}
size_t emu_packed_bytes(int64_t nreads, int max_len) { return packed_bytes(nreads, max_len); }

int emu_pack_reads(const uint8_t *ascii, int64_t row_stride, const int32_t *lens, const int32_t *starts,
                   int64_t nreads, int max_len, const uint8_t table[256], uint8_t *packed, int32_t *invalid) {
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    uint32_t *dst = (uint32_t *)packed;
    for (long long tile = 0; tile < ntiles; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const long long r = tile * 64 + lane;
            const int start = (r < nreads && starts) ? starts[r] : 0;
            const int n = (r < nreads) ? std::max(0, std::min((lens ? lens[r] : max_len) - start, max_len)) : 0;
            const uint8_t *row = ascii + (r < nreads ? r : 0) * row_stride + start;
            bool zero_seen = false;
            for (int c = 0; c < nchunks; ++c)
                for (int d = 0; d < 4; ++d)
                    dst[(((size_t)tile * nchunks + c) * 64 + lane) * 4 + d] =
                        pack_word(row, c * 32 + d * 8, n, table, zero_seen);
            if (invalid && zero_seen) *invalid += 1;
        }
    return ATR_OK;
}

// atr_pack_planes: plane64 layout (bit planes of the codes, 32 bases per 16-byte chunk)
int emu_pack_planes(const uint8_t *ascii, int64_t row_stride, const int32_t *lens, const int32_t *starts,
                    int64_t nreads, int max_len, const uint8_t table[256], uint8_t *packed, int32_t *invalid) {
    const int nchunks = (max_len + 31) / 32;
    const long long ntiles = (nreads + 63) / 64;
    uint32_t *dst = (uint32_t *)packed;
    uint32_t spread[256];
    for (int c = 0; c < 256; ++c) spread[c] = spread_code((uint32_t)table[c] & 15u);
    for (long long tile = 0; tile < ntiles; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const long long r = tile * 64 + lane;
            const int start = (r < nreads && starts) ? starts[r] : 0;
            const int n = (r < nreads) ? std::max(0, std::min((lens ? lens[r] : max_len) - start, max_len)) : 0;
            const uint8_t *row = ascii + (r < nreads ? r : 0) * row_stride + start;
            bool zero_seen = false;
            for (int c = 0; c < nchunks; ++c)
                pack_planes_chunk(row, c * 32, n, spread, zero_seen, dst + (((size_t)tile * nchunks + c) * 64 + lane) * 4);
            if (invalid && zero_seen) *invalid += 1;
        }
    return ATR_OK;
}

// atr_planes_count_uncoded
int emu_planes_count_uncoded(const uint8_t *planes, const int32_t *lens, const int32_t *other, int64_t nreads, int max_len,
                             int32_t *count) {
    const int nchunks = (max_len + 31) / 32;
    const uint32_t *src = (const uint32_t *)planes;
    for (long long r = 0; r < nreads; ++r) {
        const int len = std::max(0, std::min(std::min(lens ? lens[r] : max_len, other ? other[r] : max_len), max_len));
        bool bad = false;
        for (int c = 0; c < nchunks && 32 * c < len; ++c) {
            const uint32_t *w = src + (((size_t)(r / 64) * nchunks + c) * 64 + (r % 64)) * 4;
            const int left = len - 32 * c;
            const uint32_t want = left >= 32 ? 0xFFFFFFFFu : (1u << left) - 1u;
            bad = bad || (~(w[0] | w[1] | w[2] | w[3]) & want) != 0u;
        }
        if (bad) *count += 1;
    }
    return ATR_OK;
}

}  // extern "C"

typedef void (*emu_dp_fn)(const atr_aligner *, const uint32_t *, const int32_t *, int, int, uint32_t *,
                          const std::vector<uint32_t> &, const std::vector<std::vector<uint32_t>> &, const EmuLinked *);
template <int... I>
static emu_dp_fn pick_dp(int idx, bool noindel, std::integer_sequence<int, I...>) {
    emu_dp_fn yes[] = {&emu_dp_stage<(I + 1) * ROW_GRAN, true>...};
    emu_dp_fn no[] = {&emu_dp_stage<(I + 1) * ROW_GRAN, false>...};
    return noindel ? yes[idx] : no[idx];
}

extern "C" {
int emu_linked_create(const atr_linked_adapter *adapters, int n, atr_linked_set **out) {
    if (!out) return ATR_ERR_INVALID;
    *out = nullptr;
    atr_linked_set *s = new atr_linked_set();
    const int rc = linked_fill(s, adapters, n);
    if (rc != ATR_OK) { delete s; return rc; }
    *out = s;
    return ATR_OK;
}
void emu_linked_destroy(atr_linked_set *s) { delete s; }
int emu_linked_query_table(const atr_linked_set *s) { return s ? s->table_kind : ATR_ERR_INVALID; }
}  // extern "C"

// atr_linked_match_batch (linked_kernels.hip), read by read with the kernels' per-lane functions
template <bool WIDE, bool AND_MODE>
static void emu_linked_l1(const atr_linked_set *s, const uint32_t *pk, const int32_t *lens, long long nreads, int nchunks,
                          int max_len, int8_t *which_out, uint32_t *front, uint32_t *back, std::vector<uint32_t> &win,
                          std::vector<std::vector<std::vector<uint32_t>>> &bins) {
    const LinkedParams &P = s->p;
    for (long long r = 0; r < nreads; ++r) {
        const int n = lens ? lens[r] : max_len;
        const uint32_t *q = pk + (((size_t)(r >> 6) * nchunks) * 64 + (r & 63)) * 4;
        const uint32_t w0[4] = {q[0], q[1], q[2], q[3]};
        int which = -1, count = 0;
        uint32_t frec[4];
        rec_none(frec);
        uint32_t dpmask = 0;
        for (int a = 0; a < P.n; ++a) {
            const FrontParams &fp = P.f[a];
            const bool exact = fp.accept_full != 0 && front_exact(fp.code, fp.code_mask, w0);
            if (exact) {
                ++count;
                if (which < 0) { which = a; front_exact_record(frec, fp.m); }
            } else if (front_pex_candidate<AND_MODE>(fp.pex_code, fp.pex_mask, fp.pex_off, fp.npieces, fp.k, w0)) {
                dpmask |= 1u << a;
            }
        }
        for (int a = 0; a < P.n; ++a) {
            if (!(dpmask & (1u << a))) continue;
            const FrontParams &mp = P.f[a], &gp = P.f[P.group_first[mp.group]];
            const Uniform u = front_uniform(gp.m, gp.k, gp.indel, gp.min_overlap);
            uint32_t ns[BAND_STREAM] = {0}, rec[4];
            front_stage(w0, u.k, ns, 1);
            const uint32_t *rr = mp.rrep;
            band_locate_prefix_rr<AND_MODE>(u, [rr](int i) { return rr[i - 1]; }, gp.noindel != 0, ns, 1, n, gp.thr, rec);
            if (front_accept(rec, u.m, u.min_overlap, mp.pf_thr, mp.accept_full != 0, s->rmp.front[a], s->rmp.front_ld[a], s->rmp.front_max[a])) {
                ++count;
                if (which < 0 || a < which) { which = a; memcpy(frec, rec, 16); frec[3] = 0; }
            }
        }
        memcpy(front + 4 * r, frec, 16);
        which_out[2 * r] = (int8_t)which;
        which_out[2 * r + 1] = (int8_t)count;
        uint32_t ww = 0, brec[4];
        rec_none(brec);
        if (which >= 0) {
            const BackParams &bp = P.b[which];
            const int sft = (int)(frec[1] >> 16);
            Uniform ub = front_uniform(bp.m, bp.k, bp.indel, bp.min_overlap);
            ub.sq = true; ub.er = true;
            FilterState F;
            filter_init(F, ub, bp.rows, WIDE);
            const int z_first = sft >> 3;
            int jlast = 8 * z_first;
            for (int z = z_first; 8 * z < n; ++z) {
                const uint32_t w = read_dword(q, nchunks, z) & start_mask(z, sft);
                for (int b = 0; b < 8; ++b) {
                    const int j = 8 * z + b + 1;
                    if (j > n) break;
                    const uint32_t code = (w >> (4 * b)) & 15u;
                    filter_step<WIDE>(F, bp.peq[code][0], bp.peq[code][1], (uint32_t)ub.k);
                    jlast = j;
                }
                if ((z & 3) == 3 || 8 * (z + 1) >= n) filter_fold(F, jlast, bp.rows, (uint32_t)ub.k);      // as the kernel: once per 32-column chunk
            }
            LaneFilterParams lf;
            lf.rows = bp.rows; lf.and_mode = AND_MODE ? 1 : 0; lf.tail = bp.tail; lf.thr_row = bp.thr_row; lf.cert = bp.cert;
            ww = filter_decide<WIDE>(F, ub, lf, q, nchunks, n, brec, sft);
            if (!window_valid(ww))
                linked_finish(brec, sft, ub.m, ub.min_overlap, bp.pf_thr, bp.accept_full != 0, s->rmp.back[which], s->rmp.back_ld[which],
                              s->rmp.back_max[which]);
        }
        win[r] = ww;
        if (!window_valid(ww)) memcpy(back + 4 * r, brec, 16);
        else bins[which][window_bin(ww, P.b[which].m, lens == nullptr)].push_back((uint32_t)r);
    }
}

extern "C" {
int emu_linked_match_batch(const atr_linked_set *s, const uint8_t *packed, const int32_t *lens, int64_t nreads,
                           int max_len, int8_t *which_out, int16_t *front, int16_t *back) {
    if (!s || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (max_len == 0) return ATR_ERR_UNSUPPORTED;
    const int nchunks = (max_len + 31) / 32;
    const uint32_t *pk = (const uint32_t *)packed;
    std::vector<uint32_t> win((size_t)nreads);
    std::vector<std::vector<std::vector<uint32_t>>> bins(s->p.n, std::vector<std::vector<uint32_t>>(FILTER_BINS));
    const bool and_mode = s->p.and_mode != 0;
    if (s->p.wide) {
        if (and_mode) emu_linked_l1<true, true>(s, pk, lens, nreads, nchunks, max_len, which_out, (uint32_t *)front, (uint32_t *)back, win, bins);
        else emu_linked_l1<true, false>(s, pk, lens, nreads, nchunks, max_len, which_out, (uint32_t *)front, (uint32_t *)back, win, bins);
    } else {
        if (and_mode) emu_linked_l1<false, true>(s, pk, lens, nreads, nchunks, max_len, which_out, (uint32_t *)front, (uint32_t *)back, win, bins);
        else emu_linked_l1<false, false>(s, pk, lens, nreads, nchunks, max_len, which_out, (uint32_t *)front, (uint32_t *)back, win, bins);
    }
    for (int a = 0; a < s->p.n; ++a) {
        const atr_aligner *al = &s->back[a];
        EmuLinked la;
        la.front = (const uint32_t *)front;
        la.post = &s->post[a];
        const int idx = round_up_rows(al->p.m) / ROW_GRAN - 1;
        pick_dp(idx, al->indel_cost > al->p.k, std::make_integer_sequence<int, FILTER_MAX_M / ROW_GRAN>{})(
            al, pk, lens, nchunks, max_len, (uint32_t *)back, win, bins[a], &la);
    }
    return ATR_OK;
}

}  // extern "C"

// locate_wave_kernel (wave_kernel.hip) for one read: the 64 lanes in lock step, the cross-lane move spelled out
template <bool XREP, bool SQ, int R>
static void emu_wave_read(const atr_aligner *a_, const uint32_t *pk, int nchunks, long long r, int n, uint32_t *rec) {
    const LocateParams &p = a_->p;
    const Uniform u = make_uniform(p, round_up_rows(p.m));
    const WaveWindow win = wave_window<XREP>(u, n);
    std::vector<uint32_t> s_code(WAVE_CODE_PAD + (ATR_MAX_READ_LEN + 31) / 32 * 32 + 2 * WAVE_CODE_PAD, 0xEEEEEEEEu);   // pads: junk on purpose
    for (int c = 0; c < (n + 31) / 32; ++c)
        for (int b = 0; b < 32; ++b) {
            const uint32_t w = pk[(((size_t)(r >> 6) * nchunks + c) * 64 + (r & 63)) * 4 + (b >> 3)];
            s_code[WAVE_CODE_PAD + 32 * c + b] = (w >> (4 * (b & 7))) & 15u;
        }
    const WaveGeom g = wave_geom(u.m, R);
    WaveRows<R> W[64];
    uint32_t upa[64], upb[64];
    int a[64];
    Best best[64];
    for (int l = 0; l < 64; ++l) {
        for (int rr = 0; rr < R; ++rr) {
            const int row = wave_slot_row(g, R, l, rr);
            W[l].rowmask[rr] = (row >= 1 && row <= u.m) ? wave_rowmask(p, u.p0, row) : 0u;
            W[l].lstep[rr] = row < 0 ? 0u : row == 0 ? (SQ ? 1u : (uint32_t)u.indel << CSH) : u.delw;
            W[l].col[rr] = row < 0 ? WAVE_HUGE : init_word(row, win.min_n, u.sr, SQ, u.indel);
        }
        a[l] = win.min_n - l - 1;
        wave_best_init(best[l], u, n);
        upa[l] = upb[l] = WAVE_HUGE;
    }
    auto shr1 = [&](uint32_t *keep) {                               // keep[l] = bottom cell of lane l - 1; lane 0 untouched
        uint32_t tmp[64];
        for (int l = 1; l < 64; ++l) tmp[l] = W[l - 1].col[R - 1];
        for (int l = 1; l < 64; ++l) keep[l] = tmp[l];
    };
    shr1(upa);
    const uint32_t *code = s_code.data() + WAVE_CODE_PAD;
    const int steps = win.span > 0 ? win.span + g.lanes - 1 : 0;
    constexpr int TRIP = R == 1 ? 8 : 4;
    auto trip = [&](bool guarded) {
        uint32_t bottom[TRIP][64];
        bool hit[TRIP][64];
        int a0[64];
        for (int l = 0; l < 64; ++l) a0[l] = a[l];
        for (int s = 0; s < TRIP; ++s) {
            uint32_t *up = (s & 1) ? upa : upb, *diag = (s & 1) ? upb : upa;
            shr1(up);
            for (int l = 0; l < 64; ++l) {
                uint32_t nw[R];
                wave_rows_step<XREP, SQ, R, WAVE_ROW0_CAP>(W[l], diag[l], up[l], code[a0[l] + 1 + s], u.insw, nw);
                bottom[s][l] = nw[R - 1];
                if (guarded) {
                    ++a[l];
                    const bool active = (unsigned)(a[l] - win.min_n) < (unsigned)win.span;
                    hit[s][l] = XREP && l == g.lanes - 1 && active && nw[R - 1] < u.klimit;
                    if (active) for (int rr = 0; rr < R; ++rr) W[l].col[rr] = nw[rr];
                } else {
                    for (int rr = 0; rr < R; ++rr) W[l].col[rr] = nw[rr];
                }
            }
        }
        for (int l = 0; l < 64; ++l) {
            if (!guarded) {
                a[l] += TRIP;
                for (int s = 0; s < TRIP; ++s) hit[s][l] = XREP && l == g.lanes - 1 && bottom[s][l] < u.klimit;
            }
            for (int s = 0; s < TRIP; ++s)
                if (hit[s][l]) consider<XREP>(best[l], bottom[s][l], u.m, a0[l] + 2 + s, u.min_overlap, p.thr, u.indel);
        }
    };
    int t = 1;
    for (; t <= steps && t <= g.lanes - 1; t += TRIP) trip(true);
    for (; t + TRIP - 1 <= win.span; t += TRIP) trip(false);
    for (; t <= steps; t += TRIP) trip(true);
    Best fin = best[g.lanes - 1];
    if (win.scan) {
        const int first_row = u.er ? 0 : u.m;
        Best mine[64];
        int top = -1;
        for (int l = 0; l < 64; ++l) {
            mine[l].key = -1; mine[l].word = 0; mine[l].ref_stop = 0; mine[l].query_stop = n; mine[l].matches = 0;
            for (int rr = 0; rr < R; ++rr) {
                const int row = wave_slot_row(g, R, l, rr);
                if (row >= first_row && row <= u.m && W[l].col[rr] < u.klimit)
                    consider<XREP>(mine[l], W[l].col[rr], row, n, u.min_overlap, p.thr, u.indel);
            }
            top = std::max(top, mine[l].key < 0 ? -1 : (mine[l].key << 6) | (63 - l));
        }
        if (top >= 0 && (top >> 6) > fin.key) {
            const int src = 63 - (top & 63);
            fin.key = top >> 6; fin.word = mine[src].word; fin.ref_stop = mine[src].ref_stop; fin.query_stop = n;
            fin.matches = mine[src].matches;
        }
    }
    wave_result(fin, u, n, rec);
}

template <int R>
static void emu_wave_read_r(const atr_aligner *a, const uint32_t *pk, int nchunks, long long r, int n, uint32_t *rec) {
    const bool xrep = (a->flags & ATR_STOP_WITHIN_SEQ2) != 0, sq = (a->flags & ATR_START_WITHIN_SEQ2) != 0;
    if (xrep && sq) emu_wave_read<true, true, R>(a, pk, nchunks, r, n, rec);
    if (xrep && !sq) emu_wave_read<true, false, R>(a, pk, nchunks, r, n, rec);
    if (!xrep && sq) emu_wave_read<false, true, R>(a, pk, nchunks, r, n, rec);
    if (!xrep && !sq) emu_wave_read<false, false, R>(a, pk, nchunks, r, n, rec);
}

extern "C" {

int emu_locate_batch(const atr_aligner *a, const uint8_t *packed, const int32_t *lens, int64_t nreads,
                     int max_len, int16_t *out, int path) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_LONG_READ_LEN) return ATR_ERR_INVALID;
    if (path < ATR_LOCATE_AUTO || path > ATR_LOCATE_WAVE) return ATR_ERR_INVALID;
    if (path == ATR_LOCATE_WAVE && (a->p.m > WAVE_MAX_M || max_len > ATR_MAX_READ_LEN)) return ATR_ERR_UNSUPPORTED;
    if (nreads == 0) return ATR_OK;
    if (max_len > ATR_MAX_READ_LEN) {
        if (a->p.m + a->p.k > LONG_MAX_SPAN) return ATR_ERR_UNSUPPORTED;
        path = ATR_LOCATE_FULL;
    }
    const int filtered = path != ATR_LOCATE_FULL;
    {   // as atr_locate_batch_path
        const bool band = filtered && max_len > 0 && prefix_band_applies(a->flags, a->p.m, a->p.k);
        if (path == ATR_LOCATE_WAVE || (path == ATR_LOCATE_AUTO && !band && wave_applies(a->p.m, nreads))) {
            const int nchunks = (max_len + 31) / 32;
            for (long long r = 0; r < nreads; ++r) {
                const int n = lens ? lens[r] : max_len;
                uint32_t *rec = (uint32_t *)out + 4 * r;
                switch (wave_pair_rows(a->p.m)) {
                    case 1: emu_wave_read_r<1>(a, (const uint32_t *)packed, nchunks, r, n, rec); break;
                    case 2: emu_wave_read_r<2>(a, (const uint32_t *)packed, nchunks, r, n, rec); break;
                    default: emu_wave_read_r<3>(a, (const uint32_t *)packed, nchunks, r, n, rec); break;
                }
            }
            return ATR_OK;
        }
    }
    const bool eqmode = !(a->wildcard_ref || a->wildcard_query);
    const bool noindel = a->indel_cost > a->p.k;
    const int idx = round_up_rows(a->p.m) / ROW_GRAN - 1;
    if (filtered && max_len > 0 && prefix_band_applies(a->flags, a->p.m, a->p.k)) {   // as atr_locate_batch
        const LocateParams &p = a->p;
        const Uniform u = make_uniform(p, round_up_rows(p.m));
        uint32_t codes[FILTER_MAX_M] = {0};
        for (int i = 0; i < u.m && i < FILTER_MAX_M; ++i) codes[i] = (uint32_t)(a->codes[i] & 15u) * 0x11111111u;
        const int nchunks = (max_len + 31) / 32;
        const uint32_t *pk = (const uint32_t *)packed;
        for (long long r = 0; r < nreads; ++r) {
            const uint32_t *q = pk + (((size_t)(r >> 6) * nchunks) * 64 + (r & 63)) * 4;
            uint32_t ns[BAND_STREAM];
            band_stage(q, nchunks, -u.k, ns, 1, band_stream_dwords(u.m));
            const int n = lens ? lens[r] : max_len;
            if (!eqmode) band_locate_prefix<true>(u, codes, noindel, ns, 1, n, p.thr, (uint32_t *)out + 4 * r);
            else band_locate_prefix<false>(u, codes, noindel, ns, 1, n, p.thr, (uint32_t *)out + 4 * r);
        }
        return ATR_OK;
    }
    const bool fast = filtered && a->filterable && max_len > 0;
    emu_fn fn = pick_mt(idx, (a->flags & ATR_STOP_WITHIN_SEQ2) != 0, noindel, fast, std::make_integer_sequence<int, ATR_MAX_REF_LEN / ROW_GRAN>{});
    fn(a, (const uint32_t *)packed, lens, nreads, (max_len + 31) / 32, max_len, (uint32_t *)out);
    return ATR_OK;
}

// atr_locate_planes_applies / atr_locate_planes_batch (the emulation takes every word count up to 10)
int emu_locate_planes_applies(const atr_aligner *a, int max_len, int ragged) {
    if (!a || !a->filterable || max_len < 1 || max_len > ATR_MAX_READ_LEN) return 0;
    if (ragged) max_len = 32 * ((max_len + 31) / 32);
    const FilterParams fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap, true);
    PieceParams pp;
    if (!piece_params(a->codes, a->p.m, fp.rows, a->p.k, a->flags, a->wildcard_ref || a->wildcard_query,
                      a->table_kind == ATR_TABLE_CUSTOM, fp.thr_row, max_len, pp, a->p.thr, a->p.min_overlap)) return 0;
    const int nw = (max_len + 31) / 32;
    return (nw >= 3 && nw <= 10) ? 1 : 0;
}
int emu_locate_planes_all_widths(const atr_aligner *a, int max_len, int ragged) {          // the envelope without the instantiated widths
    if (!a || !a->filterable || max_len < 1 || max_len > 32 * PIECE_MAX_WORDS) return 0;
    if (ragged) max_len = 32 * ((max_len + 31) / 32);
    const FilterParams fp = filter_params(a->peq, a->codes, a->p.m, a->flags, a->wildcard_ref || a->wildcard_query, a->p.thr, a->p.min_overlap, true);
    PieceParams pp;
    return piece_params(a->codes, a->p.m, fp.rows, a->p.k, a->flags, a->wildcard_ref || a->wildcard_query,
                        a->table_kind == ATR_TABLE_CUSTOM, fp.thr_row, max_len, pp, a->p.thr, a->p.min_overlap) ? 1 : 0;
}

long long emu_piece_last_windows(uint32_t *out, long long cap) {
    const long long n = std::min<long long>(cap, (long long)g_piece_last_windows.size());
    for (long long i = 0; i < n; ++i) out[i] = g_piece_last_windows[(size_t)i];
    return (long long)g_piece_last_windows.size();
}

void emu_piece_stats(long long out[18], int reset) {
    for (int i = 0; i < 6; ++i) { out[i] = g_piece_stats[i]; if (reset) g_piece_stats[i] = 0; }
    for (int i = 0; i < 12; ++i) { out[6 + i] = g_piece_need[i]; if (reset) g_piece_need[i] = 0; }
}

int emu_locate_planes_batch(const atr_aligner *a, const uint8_t *planes, const int32_t *lens, int64_t nreads, int max_len,
                            int16_t *out) {
    if (!a || nreads < 0 || max_len < 0 || max_len > ATR_MAX_READ_LEN) return ATR_ERR_INVALID;
    if (nreads == 0) return ATR_OK;
    if (!emu_locate_planes_all_widths(a, max_len, lens != nullptr)) return ATR_ERR_UNSUPPORTED;
    const int idx = round_up_rows(a->p.m) / ROW_GRAN - 1;
    emu_fn fn = pick_piece(idx, a->indel_cost > a->p.k, std::make_integer_sequence<int, FILTER_MAX_M / ROW_GRAN>{});
    if (!fn) return ATR_ERR_UNSUPPORTED;
    fn(a, (const uint32_t *)planes, lens, nreads, (max_len + 31) / 32, max_len, (uint32_t *)out);
    return ATR_OK;
}

}  // extern "C"
