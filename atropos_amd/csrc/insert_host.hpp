// insert_host.hpp -- host-side state of an atr_insert_aligner: adapter codes for the
// chosen compare mode and the integer threshold tables the reference evaluates in
// double (k = int(frac*L), floor(j*frac)); the RMP tables and round(alen*frac) come
// from the Python host (they need Python's bigint/round semantics).  Pure C++.
#ifndef ATR_INSERT_HOST_HPP
#define ATR_INSERT_HOST_HPP

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "aligner_host.hpp"
#include "insert_core.hpp"

struct atr_insert_aligner {
    atr::InsertParams p;                 // rmp pointers are filled by the owner (device or host copies)
    std::vector<double> rmp_insert, rmp_adapter;   // host copies, [ld][ld]
    void *d_tables;                      // device allocation holding both tables (HIP build only)
    int cased_ok;                        // the adapters allow case-sensitive read codes (atr_insert_match_batch_coded)
};

// the case-sensitive read table (see atr_insert_match_batch_coded): DNA15 without M K R Y S, plus a/t, c/g, n
inline void case_sensitive_table(const uint8_t dna15[256], uint8_t out[256]) {
    memcpy(out, dna15, 256);
    for (const char *ch = "MKRYS"; *ch; ++ch) out[(unsigned char)*ch] = 0;
    out['a'] = 3; out['t'] = 12; out['c'] = 5; out['g'] = 10; out['n'] = 6;
}

namespace atr {

inline int insert_fill(atr_insert_aligner *h, const atr_insert_config *c) {
    if (!c || !c->adapter1 || !c->adapter2 || !c->rmp_insert || !c->rmp_adapter || !c->max_mismatch_by_alen)
        return ATR_ERR_INVALID;
    if (c->alen1 < 1 || c->alen2 < 1 || c->min_insert_overlap < 1) return ATR_ERR_INVALID;
    if (c->alen1 > INS_MAX_ADAPTER || c->alen2 > INS_MAX_ADAPTER) return ATR_ERR_UNSUPPORTED;
    if (c->rmp_ld < INS_MAX_LEN + 1 || c->n_mismatch < INS_MAX_ADAPTER + 1) return ATR_ERR_INVALID;
    const Tables &T = tables();
    InsertParams &p = h->p;
    memset(&p, 0, sizeof(p));
    p.alen1 = c->alen1; p.alen2 = c->alen2;
    p.long_adapters = (c->alen1 > 64 || c->alen2 > 64) ? 1 : 0;
    // case-sensitive read codes: in the literal compare mode an adapter letter on one of the lower-case codes (M K R Y S)
    // or a lower-case adapter letter could not be told from a soft-masked read base
    h->cased_ok = 1;
    if (!c->adapter_wildcards && !c->read_wildcards)
        for (int side = 0; side < 2; ++side) {
            const char *ad = side ? c->adapter2 : c->adapter1;
            for (int i = 0; i < (side ? c->alen2 : c->alen1); ++i)
                if (strchr("MKRYS", ad[i]) || (ad[i] >= 'a' && ad[i] <= 'z')) h->cased_ok = 0;
        }
    // compare_prefixes(read_overhang, adapter, wildcard_ref=adapter_wildcards,
    // wildcard_query=read_wildcards) (align/__init__.py:285-288, _align.pyx:521-530):
    // the READ is the "ref" side: IUPAC table if adapter_wildcards, else ACGT table if
    // read_wildcards; the ADAPTER is the "query" side: IUPAC if read_wildcards, else ACGT
    // if adapter_wildcards.  Reads are packed with upper-case IUPAC codes already.
    const uint8_t *at;
    if (c->adapter_wildcards) { p.cmp_mode = INS_CMP_AND; at = c->read_wildcards ? T.iupac : T.acgt; }
    else if (c->read_wildcards) { p.cmp_mode = INS_CMP_AND_READ_ACGT; at = T.iupac; }
    else { p.cmp_mode = INS_CMP_EQ; at = T.dna15; }
    for (int i = 0; i < c->alen1; ++i)
        for (int pl = 0; pl < 4; ++pl)
            p.a1[pl][i / 32] |= (uint32_t)((at[(unsigned char)c->adapter1[i]] >> pl) & 1) << (i % 32);
    for (int i = 0; i < c->alen2; ++i)
        for (int pl = 0; pl < 4; ++pl)
            p.a2[pl][i / 32] |= (uint32_t)((at[(unsigned char)c->adapter2[i]] >> pl) & 1) << (i % 32);
    for (int a = 0; a <= INS_MAX_ADAPTER; ++a) {
        int v = c->max_mismatch_by_alen[a];
        p.mm_by_alen[a] = (int16_t)(v < -1 ? -1 : (v > 30000 ? 30000 : v));
    }
    const double frac = c->max_insert_mismatch_frac;
    for (int j = 0; j <= INS_MAX_LEN; ++j) {
        double t = std::floor((double)j * frac);               // cost <= length*e  (_align.pyx:728)
        p.thr_ins[j] = (int16_t)(t < 0 ? -1 : (t > 30000 ? 30000 : t));
        double kd = frac * j;                                  // k = <int>(max_error_rate * m)  (:634)
        p.k_by_len[j] = (int16_t)(kd < 0 ? -1 : (kd > 30000 ? 30000 : (int)kd));
    }
    p.min_insert_overlap = c->min_insert_overlap;
    for (int j = 0; j <= INS_MAX_LEN; ++j) p.thr_hit[j] = j >= c->min_insert_overlap ? (int32_t)p.thr_ins[j] : -1;
    p.min_adapter_overlap = c->min_adapter_overlap;
    p.adapter_check_cutoff = c->adapter_check_cutoff;
    p.insert_max_rmp = c->insert_max_rmp;
    p.adapter_max_rmp = c->adapter_max_rmp;
    p.rmp_ld = INS_MAX_LEN + 1;
    // repack the host tables to the device leading dimension
    const int ld = INS_MAX_LEN + 1;
    h->rmp_insert.assign((size_t)ld * ld, 0.0);
    h->rmp_adapter.assign((size_t)ld * ld, 0.0);
    for (int s = 0; s < ld; ++s)
        for (int k = 0; k <= s; ++k) {
            h->rmp_insert[(size_t)s * ld + k] = c->rmp_insert[(size_t)s * c->rmp_ld + k];
            h->rmp_adapter[(size_t)s * ld + k] = c->rmp_adapter[(size_t)s * c->rmp_ld + k];
        }
    // hits shorter than this cannot pass the insert RMP filter even when perfect
    // (rmp(matches, size) decreases with matches): the unordered sweep only counts them
    p.min_hit_j = INS_MAX_LEN + 1;
    for (int j = 1; j <= INS_MAX_LEN; ++j)
        if (h->rmp_insert[(size_t)j * ld + j] <= p.insert_max_rmp) { p.min_hit_j = j; break; }
    return ATR_OK;
}

}  // namespace atr
#endif
