// linked_host.hpp -- host-side state of an atr_linked_set: the parameter block the fused
// linked-adapter kernels read (linked_core.hpp), derived from the set's aligners.  Pure C++ (no
// HIP) so that tests/emu builds the very same block.
#ifndef ATR_LINKED_HOST_HPP
#define ATR_LINKED_HOST_HPP

#include <cstring>
#include <new>

#include "aligner_host.hpp"
#include "linked_core.hpp"

struct atr_linked_set {
    atr::LinkedParams p;
    atr::LinkedRmp rmp;
    atr::LinkedPost post[atr::LINKED_MAX];        // per 3' adapter, for the band / window kernels
    atr_aligner back[atr::LINKED_MAX];            // copies of the 3' aligners
    int table_kind;
    void *d_params;                               // device copy of {p, rmp} (HIP build only)
    void *d_wave;                                 // device copy of the 3' aligners' LocateParams + LinkedPost (linked_wave_kernel)
};

namespace atr {

// largest c with (double)c / (double)L <= e: `errors / size <= max_error_rate`
// (adapters/__init__.py:390) as a table over the alignment length
inline void postfilter_thresholds(double e, int m, int16_t *out, int n_out) {
    for (int L = 0; L < n_out; ++L) {
        int c = -1;
        if (L >= 1 && L <= m) {
            c = (int)std::floor((double)L * e) + 2;
            if (c > 4000) c = 4000;
            while (c >= 0 && !((double)c / (double)L <= e)) --c;
        }
        out[L] = (int16_t)c;
    }
}

// Myers match masks for `rows` rows starting at reference row `row0` (0-based), top-aligned in a
// vector of W bits with always-matching pad rows below (filter_core.hpp, FilterState).
inline uint64_t linked_peq(const atr_aligner *a, int code, int rows, int W) {
    const bool eqmode = !(a->wildcard_ref || a->wildcard_query);
    uint64_t mask = 0;
    for (int i = 0; i < rows; ++i) {
        const int rc = a->codes[i];
        if (eqmode ? (rc == code) : ((rc & code) != 0)) mask |= 1ull << i;
    }
    const int off = W - rows;
    if (off > 0) mask = (mask << off) | ((1ull << off) - 1ull);
    if (W == 32) mask &= 0xFFFFFFFFull;
    return mask;
}

inline bool linked_front_ok(const atr_aligner *f) {
    const int m = f->p.m, k = f->p.k;
    return f->flags == ATR_STOP_WITHIN_SEQ2 && k >= 0 && k < m && m + k <= FRONT_MAX_M && 2 * k + 1 <= BAND_W &&
           k + 1 <= FRONT_MAX_PIECES &&
           f->table_kind != ATR_TABLE_CUSTOM;
}
inline bool linked_back_ok(const atr_aligner *b) {
    return b->filterable && !(b->flags & ATR_START_WITHIN_SEQ1) && b->p.k < b->p.m && b->table_kind != ATR_TABLE_CUSTOM;
}

// adapters[i] = {front, back, front_exact_shortcut, back_exact_shortcut, rmp ...} (atropos_hip.h)
inline int linked_fill(atr_linked_set *s, const atr_linked_adapter *ad, int n) {
    if (!ad || n < 1) return ATR_ERR_INVALID;
    if (n > LINKED_MAX) return ATR_ERR_UNSUPPORTED;
    memset(&s->p, 0, sizeof(s->p));
    memset(&s->rmp, 0, sizeof(s->rmp));
    memset(s->post, 0, sizeof(s->post));
    s->d_params = nullptr;
    for (int i = 0; i < n; ++i) if (!ad[i].front || !ad[i].back) return ATR_ERR_INVALID;
    s->table_kind = ad[0].front->table_kind;
    bool wide = false;
    for (int i = 0; i < n; ++i) {
        const atr_aligner *f = ad[i].front, *b = ad[i].back;
        if (!linked_front_ok(f) || !linked_back_ok(b)) return ATR_ERR_UNSUPPORTED;
        if (f->table_kind != s->table_kind || b->table_kind != s->table_kind) return ATR_ERR_UNSUPPORTED;
        // the literal shortcut on a wildcard comparison cannot be read off the alignment records
        const bool and_f = f->wildcard_ref || f->wildcard_query, and_b = b->wildcard_ref || b->wildcard_query;
        if ((ad[i].front_exact_shortcut && and_f) || (ad[i].back_exact_shortcut && and_b)) return ATR_ERR_UNSUPPORTED;
        if ((ad[i].d_front_rmp && ad[i].front_rmp_ld < 1) || (ad[i].d_back_rmp && ad[i].back_rmp_ld < 1)) return ATR_ERR_INVALID;
        if (b->p.m > FILTER_NARROW_ROWS && !filter_narrow_applies(b->p.m, b->flags)) wide = true;
    }
    LinkedParams &P = s->p;
    P.n = n;
    P.and_mode = s->table_kind != ATR_TABLE_DNA15 ? 1 : 0;
    P.wide = wide ? 1 : 0;
    for (int i = 0; i < n; ++i) {
        const atr_aligner *f = ad[i].front, *b = ad[i].back;
        FrontParams &F = P.f[i];
        const int m = f->p.m;
        const bool eqmode = !(f->wildcard_ref || f->wildcard_query);
        (void)eqmode;
        {   // the k + 1 pieces of the exact-piece filter: nearly equal lengths, the first eight bases of each
            const int np = f->p.k + 1;
            F.npieces = np;
            for (int pc = 0; pc < np; ++pc) {
                const int lo = (int)((long long)m * pc / np), hi = (int)((long long)m * (pc + 1) / np);
                F.pex_off[pc] = lo;
                for (int r = lo; r < hi && r < lo + 8; ++r) {
                    F.pex_code[pc] |= (uint32_t)(f->codes[r] & 15u) << (4 * (r - lo));
                    F.pex_mask[pc] |= 15u << (4 * (r - lo));
                }
            }
        }
        for (int r = 0; r < m; ++r) {
            F.code[r >> 3] |= (uint32_t)(f->codes[r] & 15u) << (4 * (r & 7));
            F.code_mask[r >> 3] |= 15u << (4 * (r & 7));
            F.rrep[r] = (uint32_t)(f->codes[r] & 15u) * 0x11111111u;
        }
        for (int L = 0; L <= m + 1; ++L) F.thr[L] = f->p.thr[L];
        postfilter_thresholds(f->max_error_rate, m, F.pf_thr, FRONT_MAX_M + 2);
        F.m = m; F.k = f->p.k; F.min_overlap = f->p.min_overlap; F.indel = f->p.indel;
        F.noindel = f->indel_cost > f->p.k ? 1 : 0;
        F.accept_full = ad[i].front_exact_shortcut ? 1 : 0;
        // DP group: adapters whose banded DP runs with the same wave-uniform parameters
        int g = -1;
        for (int j = 0; j < i && g < 0; ++j) {
            const FrontParams &G = P.f[j];
            if (G.m == F.m && G.k == F.k && G.min_overlap == F.min_overlap && G.indel == F.indel && G.noindel == F.noindel &&
                memcmp(G.thr, F.thr, sizeof(F.thr)) == 0) g = G.group;
        }
        if (g < 0) { g = P.ngroups++; P.group_first[g] = i; }
        F.group = g;
        P.group_mask[g] |= 1u << i;

        BackParams &B = P.b[i];
        const int mb = b->p.m;
        const bool narrow = !wide && filter_narrow_applies(mb, b->flags);
        const int rows = narrow ? FILTER_NARROW_ROWS : mb, W = wide ? 64 : 32;
        for (int c = 0; c < 16; ++c) {
            const uint64_t mask = linked_peq(b, c, rows, W);
            B.peq[c][0] = (uint32_t)mask; B.peq[c][1] = (uint32_t)(mask >> 32);
        }
        for (int r = 0; r <= FILTER_MAX_M; ++r)
            B.thr_row[r] = (r >= 1 && r <= mb && r >= b->p.min_overlap && ((b->flags & ATR_STOP_WITHIN_SEQ1) || r == mb))
                               ? (int32_t)b->p.thr[r] : -1;
        postfilter_thresholds(b->max_error_rate, mb, B.pf_thr, FILTER_MAX_M + 2);
        B.tail = 0u;
        if (narrow)
            for (int t = 0; t < mb - FILTER_NARROW_ROWS; ++t) B.tail |= (uint32_t)(b->codes[FILTER_NARROW_ROWS + t] & 15u) << (4 * t);
        filter_overlap_certificates(b->codes, mb, rows, B.thr_row, b->wildcard_ref || b->wildcard_query, B.cert);
        B.rows = rows; B.m = mb; B.k = b->p.k; B.min_overlap = b->p.min_overlap; B.indel = b->p.indel;
        B.accept_full = ad[i].back_exact_shortcut ? 1 : 0;

        s->back[i] = *b;
        LinkedPost &Q = s->post[i];
        memcpy(Q.pf_thr, B.pf_thr, sizeof(Q.pf_thr));
        Q.m = mb; Q.min_overlap = b->p.min_overlap; Q.accept_full = B.accept_full;
        Q.rmp = ad[i].d_back_rmp; Q.rmp_ld = ad[i].back_rmp_ld; Q.max_rmp = ad[i].back_max_rmp;
        s->rmp.front[i] = ad[i].d_front_rmp; s->rmp.front_max[i] = ad[i].front_max_rmp; s->rmp.front_ld[i] = ad[i].front_rmp_ld;
        s->rmp.back[i] = ad[i].d_back_rmp; s->rmp.back_max[i] = ad[i].back_max_rmp; s->rmp.back_ld[i] = ad[i].back_rmp_ld;
    }
    P.pex_shared = 1;
    for (int i = 1; i < n; ++i)
        if (P.f[i].npieces != P.f[0].npieces || P.f[i].k != P.f[0].k ||
            memcmp(P.f[i].pex_off, P.f[0].pex_off, sizeof(P.f[0].pex_off)) != 0) P.pex_shared = 0;
    // excl (linked_core.hpp): b cannot match a read that starts with a verbatim when the unit-cost edit distance --
    // a lower bound of the aligner's cost -- between b and every read prefix of m_b - k_b .. m_b + k_b bases exceeds
    // k_b.  Prefixes up to m_a bases are prefixes of a; a longer one is a plus x unknown bases: dist(b, a + x) >=
    // dist(b, a) - |x|.
    if (!P.and_mode) {
        for (int a = 0; a < n; ++a) {
            if (!P.f[a].accept_full) continue;
            const atr_aligner *fa = ad[a].front;
            const int ma = fa->p.m;
            for (int b = 0; b < n; ++b) {
                if (b == a) continue;
                const atr_aligner *fb = ad[b].front;
                const int mb = fb->p.m, kb = fb->p.k;
                int D[FRONT_MAX_M + 1][FRONT_MAX_M + 1];
                for (int i = 0; i <= mb; ++i) D[i][0] = i;
                for (int j = 0; j <= ma; ++j) D[0][j] = j;
                for (int i = 1; i <= mb; ++i)
                    for (int j = 1; j <= ma; ++j)
                        D[i][j] = std::min(std::min(D[i - 1][j] + 1, D[i][j - 1] + 1),
                                           D[i - 1][j - 1] + ((fb->codes[i - 1] & 15) != (fa->codes[j - 1] & 15) ? 1 : 0));
                int low = 1 << 20;
                for (int j = std::max(0, mb - kb); j <= std::min(ma, mb + kb); ++j) low = std::min(low, D[mb][j]);
                if (mb + kb > ma) low = std::min(low, D[mb][ma] - (mb + kb - ma));
                if (low > kb) P.excl[a] |= 1u << b;
            }
        }
    }
    return ATR_OK;
}

}  // namespace atr
#endif
