// aligner_host.hpp -- host-side state of an atr_aligner: translate tables, the
// 4-bit reference codes, and every quantity the reference computes in floating
// point, hoisted out of the kernel (k = int(e*m), floor(L*e) per length).
// Pure C++ (no HIP) so that tests/emu can build the very same parameter block.
#ifndef ATR_ALIGNER_HOST_HPP
#define ATR_ALIGNER_HOST_HPP

#include <cmath>
#include <cstring>
#include <new>
#include <atomic>
#include <string>

#include "atropos_hip.h"
#include "locate_core.hpp"

struct atr_aligner {
    atr::LocateParams p;
    uint64_t peq[16];                 // Myers match masks by query code (filter_core.hpp)
    int filterable;                   // the filtered pipeline applies to this aligner
    double max_error_rate;
    int flags, wildcard_ref, wildcard_query, min_overlap, indel_cost;
    int table_kind;
    uint8_t qtable[256];              // query translate table
    uint8_t codes[ATR_MAX_REF_LEN];   // 4-bit reference codes, row order
    std::string ref;                  // raw reference bytes
    mutable long long planes_seen = 0;   // reads this handle has put through the two-pass pre-pass (jit.hpp's policy)
    unsigned long long uid = 0;          // changes whenever the derived parameters do (aligner_refresh): the key of per-call
                                         // caches of what is computed from them (aligner_filter_params)
};

namespace atr {

struct Tables {
    uint8_t acgt[256], iupac[256], dna15[256];
    Tables() {
        memset(acgt, 0, 256); memset(iupac, 0, 256); memset(dna15, 0, 256);
        const char *sym = "ACGTRYSWKMBDHVN";
        const int code[15] = {1, 2, 4, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};
        for (int i = 0; i < 15; ++i) {
            const unsigned char up = (unsigned char)sym[i], lo = (unsigned char)(sym[i] | 0x20);
            iupac[up] = iupac[lo] = (uint8_t)code[i];        // _align.pyx:46-83 (X stays 0)
            dna15[up] = (uint8_t)code[i];                    // upper case only: equality semantics
            if (i < 4) acgt[up] = acgt[lo] = (uint8_t)code[i];   // _align.pyx:31-44
        }
        acgt['U'] = acgt['u'] = 8;
        iupac['U'] = iupac['u'] = 8;
    }
};

inline const Tables &tables() {
    static const Tables t;
    return t;
}

inline int round_up_rows(int m) { return (m + ROW_GRAN - 1) / ROW_GRAN * ROW_GRAN; }

// Derive everything that depends on (m, e, flags, indel_cost, min_overlap).
inline unsigned long long aligner_next_uid() {
    static std::atomic<unsigned long long> next{1};
    return next.fetch_add(1);
}

inline int aligner_refresh(atr_aligner *a) {
    a->uid = aligner_next_uid();
    LocateParams &p = a->p;
    const int m = p.m;
    const double e = a->max_error_rate;
    const double kd = e * m;
    if (!(kd < 256.0) || !(kd > -1.0e9)) return ATR_ERR_UNSUPPORTED;   // the 8-bit mismatch field must hold k
    p.k = (int)kd;                                        // _align.pyx:312
    if (p.k < 0) p.k = -1;                                // negative error rate: nothing can be accepted
    p.flags = a->flags;
    p.min_overlap = a->min_overlap;
    p.indel = a->indel_cost > p.k ? p.k + 1 : a->indel_cost;
    if (p.indel < 1) p.indel = 1;                         // k < 0 only for a negative error rate
    {   // mismatch masks by register position: row i (1-based) lives at position p0 + i, and
        // the row at position p+1 owns bit p
        const int p0 = round_up_rows(m) - m;
        const bool eqmode = !(a->wildcard_ref || a->wildcard_query);
        memset(p.nmask, 0, sizeof(p.nmask));
        for (int c = 0; c < 16; ++c)
            for (int i = 0; i < m; ++i) {
                const int rc = a->codes[i];
                const bool same = eqmode ? (rc == c) : ((rc & c) != 0);   // _align.pyx:390-393
                const int pos = p0 + i;
                if (!same) p.nmask[c][pos >> 5] |= 1u << (pos & 31);
            }
    }
    {   // match masks of the bit-parallel pre-pass, and whether it applies
        const bool eqmode = !(a->wildcard_ref || a->wildcard_query);
        for (int c = 0; c < 16; ++c) {
            uint64_t mask = 0;
            for (int i = 0; i < m && i < 64; ++i) {
                const int rc = a->codes[i];
                if (eqmode ? (rc == c) : ((rc & c) != 0)) mask |= 1ull << i;
            }
            // top-aligned with always-matching pad rows below (filter_core.hpp, FilterState)
            const int W = m > 32 ? 64 : 32, off = m <= 64 ? W - m : 0;
            mask = off ? ((mask << off) | ((1ull << off) - 1)) : mask;
            a->peq[c] = mask;
        }
        const int need = ATR_START_WITHIN_SEQ2 | ATR_STOP_WITHIN_SEQ2;
        a->filterable = (m <= 64 && (a->flags & need) == need && p.k >= 0) ? 1 : 0;
    }
    for (int L = 0; L <= m + 1; ++L) {
        // integer cost <= L*e  <=>  cost <= floor(L*e)   (_align.pyx:447, :468)
        double t = std::floor((double)L * e);
        if (t > COST_FIELD_MAX) t = COST_FIELD_MAX;
        // a negative product accepts nothing; costs are >= 0, so encode it as "cost <= -1"
        p.thr[L] = t < 0 ? (int16_t)-1 : (int16_t)t;
    }
    // Worst-case cost of a computed cell is row 0 plus m insertions; keep it (and one more
    // indel on top) inside the 12-bit cost field.  Without indels a cell is at most an
    // initial value (<= INIT_COST_CAP) plus one mismatch per column, which always fits.
    const bool sq = (a->flags & ATR_START_WITHIN_SEQ2) != 0;
    if (a->indel_cost <= p.k) {
        const long long bound = (long long)((sq ? 0 : (m + p.k)) + m + 1) * p.indel;
        if (bound > INIT_COST_CAP) return ATR_ERR_UNSUPPORTED;
    }
    return ATR_OK;
}

inline int aligner_create(const char *ref, int m, double max_error_rate, int flags, int wildcard_ref,
                          int wildcard_query, int min_overlap, int indel_cost, atr_aligner **out) {
    if (!out) return ATR_ERR_INVALID;
    *out = nullptr;
    if (!ref || m < 1 || min_overlap < 1 || indel_cost < 1 || flags < 0 || flags > 15) return ATR_ERR_INVALID;
    if (m > ATR_MAX_REF_LEN) return ATR_ERR_UNSUPPORTED;
    for (int i = 0; i < m; ++i) if ((unsigned char)ref[i] >= 128) return ATR_ERR_INVALID;
    const Tables &T = tables();
    atr_aligner *a = new (std::nothrow) atr_aligner();
    if (!a) return ATR_ERR_NOMEM;
    memset(&a->p, 0, sizeof(a->p));
    a->ref.assign(ref, ref + m);
    a->max_error_rate = max_error_rate;
    a->flags = flags;
    a->wildcard_ref = wildcard_ref != 0;
    a->wildcard_query = wildcard_query != 0;
    a->min_overlap = min_overlap;
    a->indel_cost = indel_cost;
    a->p.m = m;
    // Reference side: _align.pyx:245-248.  Query side: :292-297.
    const uint8_t *rt = nullptr;
    if (a->wildcard_ref) rt = T.iupac; else if (a->wildcard_query) rt = T.acgt;
    if (a->wildcard_query) { a->table_kind = ATR_TABLE_IUPAC; memcpy(a->qtable, T.iupac, 256); }
    else if (a->wildcard_ref) { a->table_kind = ATR_TABLE_ACGT; memcpy(a->qtable, T.acgt, 256); }
    if (rt) {
        for (int i = 0; i < m; ++i) a->codes[i] = rt[(unsigned char)ref[i]];
    } else {
        // Byte-equality mode (:298, :390-391).  4 bits hold 15 distinct non-zero symbols:
        // the canonical DNA15 map if the reference fits it, else a per-aligner symbol map;
        // every other query byte maps to 0, which equals no reference symbol.
        bool canon = true;
        for (int i = 0; i < m; ++i) if (!T.dna15[(unsigned char)ref[i]]) { canon = false; break; }
        if (canon) {
            a->table_kind = ATR_TABLE_DNA15;
            memcpy(a->qtable, T.dna15, 256);
        } else {
            a->table_kind = ATR_TABLE_CUSTOM;
            memset(a->qtable, 0, 256);
            int next = 1;
            for (int i = 0; i < m; ++i) {
                uint8_t &slot = a->qtable[(unsigned char)ref[i]];
                if (!slot) {
                    if (next > 15) { delete a; return ATR_ERR_UNSUPPORTED; }
                    slot = (uint8_t)next++;
                }
            }
        }
        for (int i = 0; i < m; ++i) a->codes[i] = a->qtable[(unsigned char)ref[i]];
    }
    const int rc = aligner_refresh(a);
    if (rc != ATR_OK) { delete a; return rc; }
    *out = a;
    return ATR_OK;
}

inline int aligner_set_min_overlap(atr_aligner *a, int min_overlap) {
    if (!a || min_overlap < 1) return ATR_ERR_INVALID;
    a->min_overlap = min_overlap;
    return aligner_refresh(a);
}

inline int aligner_set_indel_cost(atr_aligner *a, int indel_cost) {
    if (!a || indel_cost < 1) return ATR_ERR_INVALID;
    const int old = a->indel_cost;
    a->indel_cost = indel_cost;
    const int rc = aligner_refresh(a);
    if (rc != ATR_OK) { a->indel_cost = old; aligner_refresh(a); }
    return rc;
}

inline size_t packed_bytes(int64_t nreads, int max_len) {
    if (nreads < 0 || max_len < 0) return 0;
    const size_t ntiles = (size_t)((nreads + 63) / 64);
    const size_t nchunks = (size_t)((max_len + 31) / 32);
    return ntiles * nchunks * 64 * 16;
}

}  // namespace atr
#endif
