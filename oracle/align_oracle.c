/*
 * oracle/align_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's alignment hot
 * path, used only as the parity checker (tests/, __graft_entry__.smoke()) and
 * as the "port" CPU baseline timed by bench.py.  Nothing under atropos_amd/
 * may link, import or call this file; the product path is HIP-only.
 *
 * Parity status: PINNED.  Every entry point is checked tuple-for-tuple against
 * the reference's own implementation (imported in the build container from
 * /root/reference, Cython-compiled in a scratch dir) by
 * tests/golden/make_golden.py, and against the committed golden vectors in
 * tests/golden/ by tests/test_oracle_golden.py.
 *
 * What is restated (reference file:line):
 *   orc_acgt_table / orc_iupac_table  atropos/align/_align.pyx:31-86
 *   orc_locate                        atropos/align/_align.pyx:266-491
 *   orc_compare_prefixes              atropos/align/_align.pyx:501-544
 *   orc_compare_suffixes              atropos/align/__init__.py:28-44
 *   orc_multi_locate                  atropos/align/_align.pyx:593-783
 *   orc_reverse_complement            atropos/util/__init__.py:67-88,479-482
 *   orc_match_insert                  atropos/align/__init__.py:250-377
 *   orc_match_to                      atropos/adapters/__init__.py:338-400 (adapters with indels, no RMP filter)
 *   orc_linked                        atropos/adapters/__init__.py:671-690 under
 *                                     atropos/commands/trim/modifiers.py:107-122 (which adapter, both matches)
 *   orc_correct_errors                atropos/commands/trim/modifiers.py:219-350 (ErrorCorrectorMixin.correct_errors)
 *   orc_insert_correct_many           atropos/commands/trim/modifiers.py:397-404, :448-449 (the correction step of
 *                                     InsertAdapterCutter.__call__ right after an insert match with errors)
 *   orc_quality_trim_index            atropos/commands/trim/_qualtrim.pyx:7-50  (quality_trim_index)
 *   orc_nextseq_trim_index            atropos/commands/trim/_qualtrim.pyx:53-84 (nextseq_trim_index)
 *   orc_n_end_trim                    atropos/commands/trim/modifiers.py:766-784 (NEndTrimmer: ^N+ / N+$, upper-case N only)
 *   orc_locate_many / orc_linked_many / orc_match_insert_many (threaded drivers) -- harness only, no reference twin
 *
 * The DP keeps the reference's exact evaluation order: one column of
 * (cost, matches, origin) cells, Ukkonen's `last` cut-off, tie order
 * mismatch <= insertion <= deletion, candidate rule "more matches, then fewer
 * errors, then first seen".  Floating point appears in exactly the places the
 * reference has it: k = (int)(e*m) and cost <= length*e, both in double.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

enum { F_START_REF = 1, F_START_QRY = 2, F_STOP_REF = 4, F_STOP_QRY = 8 };

#define ORC_OVERHANG 100000  /* _align.pyx:546 */

/* ---- translate tables (_align.pyx:31-86) ------------------------------- */

static unsigned char g_acgt[256], g_iupac[256];
static int g_tables_ready;

static void put2(unsigned char *t, char c, int v) {
    t[(unsigned char)c] = (unsigned char)v;
    t[(unsigned char)(c | 0x20)] = (unsigned char)v;   /* lower case twin */
}

static void tables_init(void) {
    if (g_tables_ready) return;
    memset(g_acgt, 0, 256);
    memset(g_iupac, 0, 256);
    const int A = 1, C = 2, G = 4, T = 8;
    put2(g_acgt, 'A', A); put2(g_acgt, 'C', C); put2(g_acgt, 'G', G);
    put2(g_acgt, 'T', T); put2(g_acgt, 'U', T);
    put2(g_iupac, 'X', 0);
    put2(g_iupac, 'A', A); put2(g_iupac, 'C', C); put2(g_iupac, 'G', G);
    put2(g_iupac, 'T', T); put2(g_iupac, 'U', T);
    put2(g_iupac, 'R', A | G); put2(g_iupac, 'Y', C | T);
    put2(g_iupac, 'S', G | C); put2(g_iupac, 'W', A | T);
    put2(g_iupac, 'K', G | T); put2(g_iupac, 'M', A | C);
    put2(g_iupac, 'B', C | G | T); put2(g_iupac, 'D', A | G | T);
    put2(g_iupac, 'H', A | C | T); put2(g_iupac, 'V', A | C | G);
    put2(g_iupac, 'N', A | C | G | T);
    g_tables_ready = 1;
}

void orc_acgt_table(unsigned char out[256]) { tables_init(); memcpy(out, g_acgt, 256); }
void orc_iupac_table(unsigned char out[256]) { tables_init(); memcpy(out, g_iupac, 256); }

/* ---- one DP cell -------------------------------------------------------- */

typedef struct { int cost, matches, origin; } cell_t;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Fill the initial column (column min_n) for the four start-flag cases.
 * unit = per-gap cost used for the leading overhang: indel cost for Aligner
 * (_align.pyx:333-352), ORC_OVERHANG for MultiAligner (_align.pyx:646-665). */
static void init_column(cell_t *col, int m, int min_n, int flags, int unit) {
    const int sr = (flags & F_START_REF) != 0, sq = (flags & F_START_QRY) != 0;
    for (int i = 0; i <= m; ++i) {
        col[i].matches = 0;
        if (!sr && !sq)      { col[i].cost = imax(i, min_n) * unit; col[i].origin = 0; }
        else if (sr && !sq)  { col[i].cost = min_n * unit;          col[i].origin = imin(0, min_n - i); }
        else if (!sr && sq)  { col[i].cost = i * unit;              col[i].origin = imax(0, min_n - i); }
        else                 { col[i].cost = imin(i, min_n) * unit; col[i].origin = min_n - i; }
    }
}

typedef struct { int origin, cost, matches, ref_stop, query_stop; } hit_t;

static void hit_to_tuple(const hit_t *h, int out[6]) {
    int s1 = 0, s2 = 0;
    if (h->origin >= 0) s2 = h->origin; else s1 = -h->origin;
    out[0] = s1; out[1] = h->ref_stop; out[2] = s2; out[3] = h->query_stop;
    out[4] = h->matches; out[5] = h->cost;
}

/* ---- Aligner.locate (_align.pyx:266-491) -------------------------------- */

/* s1/s2 are already translated when a wildcard flag is set; `ascii` selects
 * byte equality versus 4-bit AND (_align.pyx:298, :390-393). */
static int locate_core(const unsigned char *s1, int m, const unsigned char *s2, int n,
                       double e, int flags, int ascii, int min_overlap, int indel,
                       cell_t *col, int out[6]) {
    const int sq = (flags & F_START_QRY) != 0, sr = (flags & F_START_REF) != 0;
    const int eq_ = (flags & F_STOP_QRY) != 0, er = (flags & F_STOP_REF) != 0;
    const int k = (int)(e * m);                       /* :312 */
    int max_n = n, min_n = 0;
    if (!sq) max_n = imin(n, m + k);                  /* :317-319 */
    if (!eq_) min_n = imax(0, n - m - k);             /* :320-321 */

    init_column(col, m, min_n, flags, indel);

    hit_t best = { 0, m + n, 0, m, n };               /* :358-363 */
    int last = sr ? m : imin(m, k + 1);               /* :366-368 */

    for (int j = min_n + 1; j <= max_n; ++j) {
        cell_t diag = col[0];
        if (sq) col[0].origin = j; else col[0].cost = j * indel;   /* :385-388 */
        const unsigned char qc = s2[j - 1];
        for (int i = 1; i <= last; ++i) {
            const int same = ascii ? (s1[i - 1] == qc) : ((s1[i - 1] & qc) != 0);
            cell_t nw;
            if (same) {
                nw.cost = diag.cost; nw.origin = diag.origin; nw.matches = diag.matches + 1;
            } else {
                const int c_sub = diag.cost + 1;
                const int c_del = col[i].cost + indel;
                const int c_ins = col[i - 1].cost + indel;
                if (c_sub <= c_del && c_sub <= c_ins) {            /* :405 */
                    nw.cost = c_sub; nw.origin = diag.origin; nw.matches = diag.matches;
                } else if (c_ins <= c_del) {                       /* :410 */
                    nw.cost = c_ins; nw.origin = col[i - 1].origin; nw.matches = col[i - 1].matches;
                } else {
                    nw.cost = c_del; nw.origin = col[i].origin; nw.matches = col[i].matches;
                }
            }
            diag = col[i];
            col[i] = nw;
        }
        while (last >= 0 && col[last].cost > k) --last;           /* :433-434 */
        if (last < m) {
            ++last;
        } else if (eq_) {
            const int length = m + imin(col[m].origin, 0);
            const int cost = col[m].cost, matches = col[m].matches;
            if (length >= min_overlap && cost <= length * e &&
                (matches > best.matches || (matches == best.matches && cost < best.cost))) {
                best.matches = matches; best.cost = cost; best.origin = col[m].origin;
                best.ref_stop = m; best.query_stop = j;
                if (cost == 0 && matches == m) break;             /* :456-458 */
            }
        }
    }

    if (max_n == n) {                                              /* :461-474 */
        for (int i = er ? 0 : m; i <= m; ++i) {
            const int length = i + imin(col[i].origin, 0);
            const int cost = col[i].cost, matches = col[i].matches;
            if (length >= min_overlap && cost <= length * e &&
                (matches > best.matches || (matches == best.matches && cost < best.cost))) {
                best.matches = matches; best.cost = cost; best.origin = col[i].origin;
                best.ref_stop = i; best.query_stop = n;
            }
        }
    }
    if (best.cost == m + n) return 0;                             /* :476-480 */
    hit_to_tuple(&best, out);
    /* `assert best.ref_stop - start1 > 0` (:490, "Do not return empty alignments").  It cannot fire: a candidate needs
     * length = ref_stop - start1 >= min_overlap (:446, :468) and the min_overlap setter refuses values below 1
     * (:218-221) -- kept as a return code of its own so that the restatement does not drop a line of the reference. */
    if (out[1] - out[0] <= 0) return -2;
    return 1;
}

static void translate(unsigned char *dst, const unsigned char *src, int len, const unsigned char *tab) {
    for (int i = 0; i < len; ++i) dst[i] = tab[src[i]];
}

/* Public: one (reference, query) pair, strings are raw ASCII bytes.
 * Returns 1 and fills out[6], or 0 for "None".  -1 on allocation failure. */
int orc_locate(const char *ref, int m, const char *query, int n, double e, int flags,
               int wc_ref, int wc_query, int min_overlap, int indel_cost, int out[6]) {
    tables_init();
    unsigned char *buf = (unsigned char *)malloc((size_t)m + n + 2);
    cell_t *col = (cell_t *)malloc(sizeof(cell_t) * ((size_t)m + 1));
    if (!buf || !col) { free(buf); free(col); return -1; }
    unsigned char *r = buf, *q = buf + m + 1;
    /* reference side: _align.pyx:245-248; query side: :292-297 */
    if (wc_ref) translate(r, (const unsigned char *)ref, m, g_iupac);
    else if (wc_query) translate(r, (const unsigned char *)ref, m, g_acgt);
    else memcpy(r, ref, m);
    if (wc_query) translate(q, (const unsigned char *)query, n, g_iupac);
    else if (wc_ref) translate(q, (const unsigned char *)query, n, g_acgt);
    else memcpy(q, query, n);
    int rc = locate_core(r, m, q, n, e, flags, !(wc_ref || wc_query), min_overlap, indel_cost, col, out);
    free(buf); free(col);
    return rc;
}

/* ---- compare_prefixes / compare_suffixes -------------------------------- */

void orc_compare_prefixes(const char *ref, int m, const char *query, int n,
                          int wc_ref, int wc_query, int out[6]) {
    tables_init();
    const int len = imin(m, n);
    int matches = 0;
    if (!wc_ref && !wc_query) {
        for (int i = 0; i < len; ++i) matches += (ref[i] == query[i]);
    } else {
        const unsigned char *tr = wc_ref ? g_iupac : g_acgt;     /* :521-524 */
        const unsigned char *tq = wc_query ? g_iupac : g_acgt;   /* :527-530 */
        for (int i = 0; i < len; ++i)
            matches += ((tr[(unsigned char)ref[i]] & tq[(unsigned char)query[i]]) != 0);
    }
    out[0] = 0; out[1] = len; out[2] = 0; out[3] = len; out[4] = matches; out[5] = len - matches;
}

void orc_compare_suffixes(const char *ref, int m, const char *query, int n,
                          int wc_ref, int wc_query, int out[6]) {
    /* align/__init__.py:28-44: reverse both, prefix-compare, re-base. */
    const int len = imin(m, n);
    int t[6];
    orc_compare_prefixes(ref + (m - len), len, query + (n - len), len, wc_ref, wc_query, t);
    /* Hamming count over the aligned tails is order independent. */
    out[0] = m - len; out[1] = m; out[2] = n - len; out[3] = n; out[4] = t[4]; out[5] = t[5];
}

/* ---- MultiAligner.locate (_align.pyx:593-783) --------------------------- */

/* Returns the number of tuples written to out (6 ints each), 0 == None.
 * out must hold (max_matches + m + 2) tuples: the last-column scan appends
 * without checking max_matches (:750-763). */
int orc_multi_locate(const char *ref, int m, const char *query, int n, double e, int flags,
                     int min_overlap, int max_matches, int *out) {
    const int sq = (flags & F_START_QRY) != 0, sr = (flags & F_START_REF) != 0;
    const int eq_ = (flags & F_STOP_QRY) != 0, er = (flags & F_STOP_REF) != 0;
    const int max_cost = m + n;
    const int k = (int)(e * m);
    int max_n = n, min_n = 0;
    if (!sq) max_n = imin(n, m + k);
    if (!eq_) min_n = imax(0, n - m - k);
    cell_t *col = (cell_t *)malloc(sizeof(cell_t) * ((size_t)m + 1));
    hit_t *hits = (hit_t *)malloc(sizeof(hit_t) * ((size_t)max_matches + m + 2));
    if (!col || !hits) { free(col); free(hits); return -1; }
    init_column(col, m, min_n, flags, ORC_OVERHANG);
    int last = sr ? m : imin(m, k + 1);
    int nh = 0, exact = -1, broke = 0;

    for (int j = min_n + 1; j <= max_n; ++j) {
        cell_t diag = col[0];
        if (sq) col[0].origin = j; else col[0].cost = j * ORC_OVERHANG;
        const char qc = query[j - 1];
        for (int i = 1; i <= last; ++i) {
            cell_t nw = diag;                        /* diagonal only: :695-704 */
            if (ref[i - 1] == qc) nw.matches += 1; else nw.cost += 1;
            diag = col[i];
            col[i] = nw;
        }
        while (last >= 0 && col[last].cost > k) --last;
        if (last < m) { ++last; continue; }
        if (!eq_) continue;
        const int cost = col[m].cost;
        if (cost > max_cost) continue;               /* :724-725 */
        const int length = m + imin(col[m].origin, 0);
        if (length >= min_overlap && cost <= length * e) {
            hit_t *h = &hits[nh];
            h->ref_stop = m; h->query_stop = j; h->cost = cost;
            h->origin = col[m].origin; h->matches = col[m].matches;
            if (cost == 0 && h->matches == m) { exact = nh++; broke = 1; break; }
            if (++nh >= max_matches) { broke = 1; break; }
        }
    }
    if (!broke && max_n == n) {                      /* for...else: :746-763 */
        for (int i = er ? 0 : m; i <= m; ++i) {
            const int cost = col[i].cost;
            if (cost > max_cost) continue;
            const int length = i + imin(col[i].origin, 0);
            if (length >= min_overlap && cost <= length * e) {
                hit_t *h = &hits[nh++];
                h->ref_stop = i; h->query_stop = n; h->cost = cost;
                h->origin = col[i].origin; h->matches = col[i].matches;
            }
        }
    }
    int nout = 0;
    if (nh > 0) {
        if (exact >= 0) { hit_to_tuple(&hits[exact], out); nout = 1; }
        else { for (int t = 0; t < nh; ++t) hit_to_tuple(&hits[t], out + 6 * t); nout = nh; }
    }
    free(col); free(hits);
    return nout;
}

/* ---- reverse_complement (util/__init__.py:67-88, 479-482) --------------- */

/* Returns 0, or -1 when a base has no complement (the reference raises
 * KeyError).  dst may not alias src. */
int orc_reverse_complement(const char *src, int n, char *dst) {
    static char comp[256];
    static int ready;
    if (!ready) {
        memset(comp, 0, 256);
        const char *a = "ACRSWKBDN", *b = "TGYSWMVHN";
        for (int i = 0; a[i]; ++i) {
            comp[(unsigned char)a[i]] = b[i]; comp[(unsigned char)b[i]] = a[i];
            comp[(unsigned char)(a[i] | 0x20)] = (char)(b[i] | 0x20);
            comp[(unsigned char)(b[i] | 0x20)] = (char)(a[i] | 0x20);
        }
        ready = 1;
    }
    for (int i = 0; i < n; ++i) {
        const char c = comp[(unsigned char)src[n - 1 - i]];
        if (!c) return -1;
        dst[i] = c;
    }
    return 0;
}

/* ---- InsertAligner.match_insert (align/__init__.py:250-377) ------------- */

typedef struct {
    const char *adapter1; int alen1;
    const char *adapter2; int alen2;
    double insert_max_rmp, adapter_max_rmp;
    int min_insert_overlap; double max_insert_mismatch_frac;
    int min_adapter_overlap; double max_adapter_mismatch_frac;
    int adapter_check_cutoff;
    int adapter_wildcards, read_wildcards;
    /* Random-match-probability tables built by the host with the reference's
     * own expression order (util/__init__.py:117-155), row-major
     * [size][matches], leading dimension rmp_ld:
     *   rmp_insert  -> match_probability(matches, size, **base_probs)  (:359)
     *   rmp_adapter -> match_probability(matches, size)               (:303-304) */
    const double *rmp_insert; const double *rmp_adapter; int rmp_ld;
    /* round(alen * max_adapter_mismatch_frac) for alen = 0..rmp_ld-1 (:290),
     * Python's round-half-even evaluated by the host. */
    const int *max_mismatch_by_alen;
} orc_insert_params;

typedef struct { hit_t h; int offset, size; double prob; int tuple[6]; } cand_t;

/* out_insert[6]; out_m1[6], out_m2[6] are Match(astart, astop, rstart, rstop,
 * matches, errors); has[0]/has[1] say whether each Match is present.
 * Returns 1 match, 0 None, -1 error (unknown base for reverse complement),
 * -2 allocation failure. */
int orc_match_insert(const orc_insert_params *p, const char *seq1, int len1,
                     const char *seq2, int len2,
                     int out_insert[6], int out_m1[6], int out_m2[6], int has[2]) {
    const int L = imin(len1, len2);                               /* :259-265 */
    has[0] = has[1] = 0;
    char *rc = (char *)malloc((size_t)L + 1);
    int *tuples = (int *)malloc(sizeof(int) * 6 * (size_t)(100 + L + 2));
    cand_t *cands = (cand_t *)malloc(sizeof(cand_t) * (size_t)(100 + L + 2));
    if (!rc || !tuples || !cands) { free(rc); free(tuples); free(cands); return -2; }
    if (orc_reverse_complement(seq2, L, rc) != 0) { free(rc); free(tuples); free(cands); return -1; }

    const int nt = orc_multi_locate(rc, L, seq1, L, p->max_insert_mismatch_frac,
                                    F_START_REF | F_STOP_QRY, p->min_insert_overlap, 100, tuples);
    int nc = 0;
    for (int t = 0; t < nt; ++t) {                                /* :356-361 */
        const int *im = tuples + 6 * t;
        const int offset = imin(im[0], L - im[3]);
        const int size = L - offset;
        const double prob = p->rmp_insert[(size_t)size * p->rmp_ld + im[4]];
        if (prob <= p->insert_max_rmp) {
            cand_t *c = &cands[nc++];
            memcpy(c->tuple, im, sizeof(int) * 6);
            c->offset = offset; c->size = size; c->prob = prob;
        }
    }
    /* stable sort by probability (:371); insertion sort keeps equal keys in order */
    if (nc > 1) {
        for (int a = 1; a < nc; ++a) {
            cand_t key = cands[a];
            int b = a - 1;
            while (b >= 0 && cands[b].prob > key.prob) { cands[b + 1] = cands[b]; --b; }
            cands[b + 1] = key;
        }
    }
    int result = 0;
    for (int t = 0; t < nc && !result; ++t) {
        const cand_t *c = &cands[t];
        if (c->offset < p->min_adapter_overlap) {                 /* :270-276 */
            memcpy(out_insert, c->tuple, sizeof(int) * 6);
            result = 1;
            break;
        }
        int a1[6], a2[6];
        /* NB argument order: the read overhang is the *ref*, the adapter the
         * *query*, while the wildcard switches keep their names (:285-288). */
        orc_compare_prefixes(seq1 + c->size, L - c->size, p->adapter1, p->alen1,
                             p->adapter_wildcards, p->read_wildcards, a1);
        orc_compare_prefixes(seq2 + c->size, L - c->size, p->adapter2, p->alen2,
                             p->adapter_wildcards, p->read_wildcards, a2);
        const int al1 = imin(c->offset, p->alen1), al2 = imin(c->offset, p->alen2);
        const int mm1 = p->max_mismatch_by_alen[al1], mm2 = p->max_mismatch_by_alen[al2];
        if (a1[5] > mm1 && a2[5] > mm2) continue;                  /* :297-300 */
        if (imin(al1, al2) > p->adapter_check_cutoff) {           /* :302-306 */
            const double p1 = p->rmp_adapter[(size_t)al1 * p->rmp_ld + a1[4]];
            const double p2 = p->rmp_adapter[(size_t)al2 * p->rmp_ld + a2[4]];
            if (p1 * p2 > p->adapter_max_rmp) continue;
        }
        const int mism = imin(a1[5], a2[5]);                      /* :308 */
        memcpy(out_insert, c->tuple, sizeof(int) * 6);
        {   /* _create_match(a1_length, seq_len1) (:310-319) */
            const int alen = imin(al1, len1 - c->size), mmv = imin(alen, mism);
            out_m1[0] = 0; out_m1[1] = alen; out_m1[2] = c->size; out_m1[3] = len1;
            out_m1[4] = alen - mmv; out_m1[5] = mmv; has[0] = 1;
        }
        {
            const int alen = imin(al2, len2 - c->size), mmv = imin(alen, mism);
            out_m2[0] = 0; out_m2[1] = alen; out_m2[2] = c->size; out_m2[3] = len2;
            out_m2[4] = alen - mmv; out_m2[5] = mmv; has[1] = 1;
        }
        result = 1;
    }
    free(rc); free(tuples); free(cands);
    return result;
}

/* ---- threaded batch driver (bench.py cpu_baseline "port") --------------- */

typedef struct {
    const char *ref; int m; double e; int flags, wc_ref, wc_query, min_overlap, indel;
    const char *reads; const int *lens; int64_t stride; int64_t lo, hi; int *out;
} job_t;

static void *job_main(void *arg) {
    job_t *jb = (job_t *)arg;
    /* like the reference, one aligner instance per worker: the translated
     * adapter and the scratch column are set up once (_align.pyx:238-249) */
    const int m = jb->m, ascii = !(jb->wc_ref || jb->wc_query);
    int maxn = 0;
    for (int64_t r = jb->lo; r < jb->hi; ++r) if (jb->lens[r] > maxn) maxn = jb->lens[r];
    unsigned char *rbuf = (unsigned char *)malloc((size_t)m + 1);
    unsigned char *qbuf = (unsigned char *)malloc((size_t)maxn + 1);
    cell_t *col = (cell_t *)malloc(sizeof(cell_t) * ((size_t)m + 1));
    if (!rbuf || !qbuf || !col) { free(rbuf); free(qbuf); free(col); return (void *)1; }
    if (jb->wc_ref) translate(rbuf, (const unsigned char *)jb->ref, m, g_iupac);
    else if (jb->wc_query) translate(rbuf, (const unsigned char *)jb->ref, m, g_acgt);
    else memcpy(rbuf, jb->ref, m);
    for (int64_t r = jb->lo; r < jb->hi; ++r) {
        int *o = jb->out + 6 * r;
        const unsigned char *q = (const unsigned char *)jb->reads + r * jb->stride;
        const int n = jb->lens[r];
        if (jb->wc_query) { translate(qbuf, q, n, g_iupac); q = qbuf; }
        else if (jb->wc_ref) { translate(qbuf, q, n, g_acgt); q = qbuf; }
        int rc = locate_core(rbuf, m, q, n, jb->e, jb->flags, ascii, jb->min_overlap, jb->indel, col, o);
        if (rc != 1) { o[0] = o[2] = o[3] = o[4] = o[5] = 0; o[1] = -1; }   /* ref_stop = -1 <=> None */
    }
    free(rbuf); free(qbuf); free(col);
    return 0;
}

/* reads: nreads rows of `stride` ASCII bytes; out: nreads x 6 ints, ref_stop
 * (out[1]) == -1 marks None.  nthreads >= 1 contiguous shards. */
int orc_locate_many(const char *ref, int m, double e, int flags, int wc_ref, int wc_query,
                    int min_overlap, int indel_cost, const char *reads, const int *lens,
                    int64_t stride, int64_t nreads, int *out, int nthreads) {
    tables_init();
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * nthreads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    const int64_t per = (nreads + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        job_t *jb = &jobs[t];
        jb->ref = ref; jb->m = m; jb->e = e; jb->flags = flags; jb->wc_ref = wc_ref;
        jb->wc_query = wc_query; jb->min_overlap = min_overlap; jb->indel = indel_cost;
        jb->reads = reads; jb->lens = lens; jb->stride = stride; jb->out = out;
        jb->lo = per * t < nreads ? per * t : nreads;
        jb->hi = per * (t + 1) < nreads ? per * (t + 1) : nreads;
        if (nthreads == 1) job_main(jb);
        else pthread_create(&th[t], 0, job_main, jb);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
    free(th); free(jobs);
    return 0;
}


/* ---- Adapter.match_to (adapters/__init__.py:338-400), adapters with indels, max_rmp None ---- */

/* seq: the adapter (upper case, as Adapter.__init__ stores it); flags: its `where`; min_overlap:
 * already min(given, m) (:285); indel_cost: 100000 when the adapter was built with indels=False
 * (:316-322; anchored no-indel adapters use compare_prefixes/suffixes instead and are NOT covered
 * here); adapter_wildcards: after the ACGT-only override (:268-270).  The read is upper-cased
 * first (:349).  Returns 1 and the Match fields (astart, astop, rstart, rstop, matches, errors), or
 * 0 (None), or -2 (allocation failure). */
int orc_match_to(const char *seq, int m, int flags, const char *read, int n, double e, int min_overlap,
                 int indel_cost, int adapter_wildcards, int read_wildcards, int out[6]) {
    char *up = (char *)malloc((size_t)n + 1);
    if (!up) return -2;
    for (int i = 0; i < n; ++i) up[i] = (read[i] >= 'a' && read[i] <= 'z') ? (char)(read[i] - 32) : read[i];
    if (!adapter_wildcards) {                                         /* :351-367 */
        int pos = -1;
        if (flags == F_STOP_QRY) {                                    /* PREFIX: startswith */
            if (n >= m && memcmp(up, seq, (size_t)m) == 0) pos = 0;
        } else if (flags == F_START_QRY) {                            /* SUFFIX: endswith */
            if (n >= m && memcmp(up + n - m, seq, (size_t)m) == 0) pos = n - m;
        } else {
            for (int i = 0; i + m <= n && pos < 0; ++i) if (memcmp(up + i, seq, (size_t)m) == 0) pos = i;
        }
        if (pos >= 0) {
            out[0] = 0; out[1] = m; out[2] = pos; out[3] = pos + m; out[4] = m; out[5] = 0;
            free(up);
            return 1;
        }
    }
    const int rc = orc_locate(seq, m, up, n, e, flags, adapter_wildcards, read_wildcards, min_overlap, indel_cost, out);
    free(up);
    if (rc != 1) return rc < 0 ? -2 : 0;
    const int size = out[1] - out[0];                                 /* :386-398 */
    if (size >= min_overlap && (double)out[5] / (double)size <= e) return 1;
    return 0;
}

/* LinkedAdapter.match_to (:671-690) for every adapter of a set, as AdapterCutter._best_match walks
 * them (modifiers.py:107-122): *which = the first adapter whose anchored 5' part matches (-1:
 * none), *count = how many 5' parts match (> 1: the reference raises AttributeError), front /
 * back = its two matches, back relative to read[front.rstop:]; back[1] == -1: None. */
typedef struct {
    int nad;
    const char *const *fronts; const int *flens;
    const char *const *backs; const int *blens;
    double e; int min_overlap, indel_cost, read_wildcards;
    const int *front_wildcards, *back_wildcards;                    /* adapter_wildcards per part */
} orc_linked_params;

int orc_linked(const orc_linked_params *p, const char *read, int n, int *which, int *count, int front[6], int back[6]) {
    *which = -1; *count = 0;
    front[0] = front[2] = front[3] = front[4] = front[5] = 0; front[1] = -1;
    back[0] = back[2] = back[3] = back[4] = back[5] = 0; back[1] = -1;
    for (int a = 0; a < p->nad; ++a) {
        int f[6];
        const int mo = imin(p->min_overlap, p->flens[a]);
        const int rc = orc_match_to(p->fronts[a], p->flens[a], F_STOP_QRY, read, n, p->e, mo, p->indel_cost,
                                    p->front_wildcards[a], p->read_wildcards, f);
        if (rc < 0) return rc;
        if (rc == 0) continue;                                        /* :672-673 */
        ++*count;
        if (*which >= 0) continue;
        *which = a;
        memcpy(front, f, sizeof(int) * 6);
        const int rest = f[3];                                        /* read[front_match.rstop:]  (:683) */
        int b[6];
        const int rb = orc_match_to(p->backs[a], p->blens[a], F_START_QRY | F_STOP_REF | F_STOP_QRY, read + rest, n - rest,
                                    p->e, imin(p->min_overlap, p->blens[a]), p->indel_cost, p->back_wildcards[a],
                                    p->read_wildcards, b);
        if (rb < 0) return rb;
        if (rb == 1) memcpy(back, b, sizeof(int) * 6);
    }
    return 0;
}

typedef struct {
    const orc_linked_params *p; const char *reads; const int *lens; int64_t stride, lo, hi;
    signed char *which; int *front, *back;
} ljob_t;

static void *ljob_main(void *arg) {
    ljob_t *jb = (ljob_t *)arg;
    for (int64_t r = jb->lo; r < jb->hi; ++r) {
        int w, c;
        if (orc_linked(jb->p, jb->reads + r * jb->stride, jb->lens[r], &w, &c, jb->front + 6 * r, jb->back + 6 * r) != 0)
            return (void *)1;
        jb->which[2 * r] = (signed char)w;
        jb->which[2 * r + 1] = (signed char)c;
    }
    return 0;
}

int orc_linked_many(const orc_linked_params *p, const char *reads, const int *lens, int64_t stride, int64_t nreads,
                    signed char *which, int *front, int *back, int nthreads) {
    tables_init();
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    ljob_t *jobs = (ljob_t *)malloc(sizeof(ljob_t) * nthreads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    const int64_t per = (nreads + nthreads - 1) / nthreads;
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        ljob_t *jb = &jobs[t];
        jb->p = p; jb->reads = reads; jb->lens = lens; jb->stride = stride; jb->which = which; jb->front = front; jb->back = back;
        jb->lo = per * t < nreads ? per * t : nreads;
        jb->hi = per * (t + 1) < nreads ? per * (t + 1) : nreads;
        if (nthreads == 1) bad |= ljob_main(jb) != 0;
        else pthread_create(&th[t], 0, ljob_main, jb);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) { void *rv = 0; pthread_join(th[t], &rv); bad |= rv != 0; }
    free(th); free(jobs);
    return bad ? -1 : 0;
}

/* ---- threaded driver of orc_match_insert: out = npairs x 18 ints (insert tuple, Match 1, Match 2;
 * [1] == -1 of a block: None) */
typedef struct {
    const orc_insert_params *p; const char *r1, *r2; const int *l1, *l2; int64_t stride, lo, hi; int *out;
} ijob_t;

static void *ijob_main(void *arg) {
    ijob_t *jb = (ijob_t *)arg;
    for (int64_t r = jb->lo; r < jb->hi; ++r) {
        int *o = jb->out + 18 * r, has[2];
        memset(o, 0, sizeof(int) * 18);
        const int rc = orc_match_insert(jb->p, jb->r1 + r * jb->stride, jb->l1[r], jb->r2 + r * jb->stride, jb->l2[r],
                                        o, o + 6, o + 12, has);
        if (rc < 0) return (void *)1;
        if (rc == 0) o[1] = -1;
        if (rc == 0 || !has[0]) { memset(o + 6, 0, sizeof(int) * 6); o[7] = -1; }
        if (rc == 0 || !has[1]) { memset(o + 12, 0, sizeof(int) * 6); o[13] = -1; }
    }
    return 0;
}

int orc_match_insert_many(const orc_insert_params *p, const char *reads1, const int *lens1, const char *reads2,
                          const int *lens2, int64_t stride, int64_t npairs, int *out, int nthreads) {
    tables_init();
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    ijob_t *jobs = (ijob_t *)malloc(sizeof(ijob_t) * nthreads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    const int64_t per = (npairs + nthreads - 1) / nthreads;
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        ijob_t *jb = &jobs[t];
        jb->p = p; jb->r1 = reads1; jb->r2 = reads2; jb->l1 = lens1; jb->l2 = lens2; jb->stride = stride; jb->out = out;
        jb->lo = per * t < npairs ? per * t : npairs;
        jb->hi = per * (t + 1) < npairs ? per * (t + 1) : npairs;
        if (nthreads == 1) bad |= ijob_main(jb) != 0;
        else pthread_create(&th[t], 0, ijob_main, jb);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) { void *rv = 0; pthread_join(th[t], &rv); bad |= rv != 0; }
    free(th); free(jobs);
    return bad ? -1 : 0;
}

/* ---- ErrorCorrectorMixin.correct_errors (commands/trim/modifiers.py:219-350) ----------------
 * In place on seq1/qual1 (len1 bytes) and seq2/qual2 (len2 bytes); qual1/qual2 may both be NULL
 * (`has_quals` false, :239).  im = insert_match[0..3].  action: 0 'N', 1 'conservative', 2 'liberal'.
 * changed[2] = (r1_changed, r2_changed); newlen[2] = len(read.sequence) afterwards (update_read
 * replaces a changed read 1 by the TRUNCATED list when truncate_seqs cut it, :329-334: `partial`
 * compares against the un-updated len1).  Python's list semantics are kept: a negative index wraps
 * once, anything else out of range is an IndexError, slices clamp.
 * Returns 0, or -1 KeyError (a base without complement, :273, :286, :293), -2 IndexError (:272-273),
 * -3 ValueError (:245-248 quality-based action without qualities; util mean() of an empty slice,
 * :303-304); on an error nothing is written (the reference only assigns read.sequence at the end). */
static int py_index(int i, int n) {            /* list[i]: resolved position or -1 (IndexError) */
    if (i < 0) i += n;
    return (i < 0 || i >= n) ? -1 : i;
}

static void py_slice(int a, int b, int n, int *lo, int *hi) {   /* list[a:b] on a list of n items */
    if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
    *lo = a; *hi = b > a ? b : a;
}

static unsigned char g_comp[256];
static int g_comp_ready;
static void comp_init(void) {                  /* BASE_COMPLEMENTS, util/__init__.py:67-88 */
    if (g_comp_ready) return;
    memset(g_comp, 0, 256);
    const char *a = "ACRSWKBDN", *b = "TGYSWMVHN";
    for (int i = 0; a[i]; ++i) {
        g_comp[(unsigned char)a[i]] = (unsigned char)b[i]; g_comp[(unsigned char)b[i]] = (unsigned char)a[i];
        g_comp[(unsigned char)(a[i] | 0x20)] = (unsigned char)(b[i] | 0x20);
        g_comp[(unsigned char)(b[i] | 0x20)] = (unsigned char)(a[i] | 0x20);
    }
    g_comp_ready = 1;
}

int orc_correct_errors(char *seq1, char *qual1, int len1, char *seq2, char *qual2, int len2, const int im[4],
                       int action, int min_qual_difference, int truncate_seqs, int changed[2], int newlen[2]) {
    comp_init();
    changed[0] = changed[1] = 0;
    newlen[0] = len1; newlen[1] = len2;
    const int has_quals = qual1 && qual2 && len1 > 0 && len2 > 0;                 /* :239 (empty str is falsy) */
    if (!has_quals && action != 0) return -3;                                      /* :244-248 */
    int n1 = len1, n2 = len2, l2 = len2;       /* list lengths; l2 = the reference's `len2` variable */
    if (truncate_seqs) {                                                           /* :250-259 */
        if (len1 > len2) n1 = len2;
        else if (len2 > len1) { n2 = len1; l2 = len1; }
    }
    unsigned char *s1 = (unsigned char *)malloc((size_t)(2 * n1 + 2 * n2 + 4));
    int *defer = (int *)malloc(sizeof(int) * 2 * (size_t)(n1 + 1));
    if (!s1 || !defer) { free(s1); free(defer); return -4; }
    unsigned char *q1 = s1 + n1, *s2 = q1 + n1, *q2 = s2 + n2;
    memcpy(s1, seq1, n1); memcpy(s2, seq2, n2);
    if (has_quals) { memcpy(q1, qual1, n1); memcpy(q2, qual2, n2); }
    const int r1_start = im[2], r1_end = im[3];                                    /* :261-266 */
    const int r2_start = l2 - im[1], r2_end = l2 - im[0];
    int c1 = 0, c2 = 0, ndefer = 0, rc = 0;
    const int cnt1 = r1_end > r1_start ? r1_end - r1_start : 0, cnt2 = r2_end > r2_start ? r2_end - r2_start : 0;
    const int cnt = cnt1 < cnt2 ? cnt1 : cnt2;                                     /* zip of the two ranges, :269-270 */
    for (int t = 0; t < cnt && !rc; ++t) {
        const int i = py_index(r1_start + t, n1);
        if (i < 0) { rc = -2; break; }
        const int j = py_index(r2_end - 1 - t, n2);
        if (j < 0) { rc = -2; break; }
        const unsigned char base1 = s1[i], base2 = g_comp[s2[j]];
        if (!base2) { rc = -1; break; }
        if (base1 == base2) continue;
        if (action == 0) {                                                         /* :275-279 */
            s1[i] = 'N'; s2[j] = 'N'; ++c1; ++c2;
        } else if (base1 == 'N') {                                                 /* :280-284 */
            s1[i] = base2; if (has_quals) q1[i] = q2[j]; ++c1;
        } else if (base2 == 'N') {                                                 /* :285-289 */
            if (!g_comp[base1]) { rc = -1; break; }
            s2[j] = g_comp[base1]; if (has_quals) q2[j] = q1[i]; ++c2;
        } else if (has_quals) {                                                    /* :290-301 */
            const int diff = (int)q1[i] - (int)q2[j];
            if (diff >= min_qual_difference) {
                if (!g_comp[base1]) { rc = -1; break; }
                s2[j] = g_comp[base1]; q2[j] = q1[i]; ++c2;
            } else if (diff <= -min_qual_difference) {
                s1[i] = base2; q1[i] = q2[j]; ++c1;
            } else if (action == 2) {
                defer[2 * ndefer] = i; defer[2 * ndefer + 1] = j; ++ndefer;
                /* base1 / base2 of the tuple are the values read here; positions are distinct unless an
                 * index wrapped, and then the reference uses the remembered bases too: keep them */
                defer[2 * ndefer - 2] |= (int)base1 << 16; defer[2 * ndefer - 1] |= (int)base2 << 16;
            }
        }
    }
    if (!rc && ndefer) {                                                           /* :303-322 */
        int lo, hi;
        long sum1 = 0, sum2 = 0;
        py_slice(r1_start, r1_end, n1, &lo, &hi);
        const int k1 = hi - lo;
        for (int x = lo; x < hi; ++x) sum1 += q1[x];
        py_slice(r2_start, r2_end, n2, &lo, &hi);
        const int k2 = hi - lo;
        for (int x = lo; x < hi; ++x) sum2 += q2[x];
        if (k1 == 0 || k2 == 0) rc = -3;
        else {
            const double d = (double)sum1 / (double)k1 - (double)sum2 / (double)k2;
            if (d > 1) {
                for (int t = 0; t < ndefer && !rc; ++t) {
                    const int i = defer[2 * t] & 0xFFFF, j = defer[2 * t + 1] & 0xFFFF;
                    const unsigned char base1 = (unsigned char)(defer[2 * t] >> 16);
                    if (!g_comp[base1]) { rc = -1; break; }
                    s2[j] = g_comp[base1]; q2[j] = q1[i]; ++c2;
                }
            } else if (d < -1) {
                for (int t = 0; t < ndefer; ++t) {
                    const int i = defer[2 * t] & 0xFFFF, j = defer[2 * t + 1] & 0xFFFF;
                    s1[i] = (unsigned char)(defer[2 * t + 1] >> 16); q1[i] = q2[j]; ++c1;
                }
            }
        }
    }
    if (!rc) {                                                                     /* :324-350 */
        if (c1) { memcpy(seq1, s1, n1); if (has_quals) memcpy(qual1, q1, n1); newlen[0] = n1; }
        if (c2) { memcpy(seq2, s2, n2); if (has_quals) memcpy(qual2, q2, n2); }
        changed[0] = c1; changed[1] = c2;
    }
    free(s1); free(defer);
    return rc;
}

/* ---- threaded driver: InsertAdapterCutter's correction step over a batch (modifiers.py:397-404, :448-449):
 * pairs whose insert match (records: npairs x 18 ints as written by orc_match_insert_many) exists and has
 * errors > 0 get correct_errors(read1, read2, insert_match, truncate_seqs=True), in place on the four
 * equal-stride matrices.  changed / newlen: npairs x 2 ints (changed[0] = -1 / -2 / -3: the exception). */
typedef struct {
    const int *rec; char *s1, *q1, *s2, *q2; const int *l1, *l2; int64_t stride, lo, hi;
    int action, mqd; int *changed, *newlen;
} cjob_t;

static void *cjob_main(void *arg) {
    cjob_t *jb = (cjob_t *)arg;
    for (int64_t r = jb->lo; r < jb->hi; ++r) {
        const int *im = jb->rec + 18 * r;
        int *ch = jb->changed + 2 * r, *nl = jb->newlen + 2 * r;
        ch[0] = ch[1] = 0; nl[0] = jb->l1[r]; nl[1] = jb->l2[r];
        if (im[1] < 0 || im[5] <= 0) continue;
        const int rc = orc_correct_errors(jb->s1 + r * jb->stride, jb->q1 ? jb->q1 + r * jb->stride : 0, jb->l1[r],
                                          jb->s2 + r * jb->stride, jb->q2 ? jb->q2 + r * jb->stride : 0, jb->l2[r],
                                          im, jb->action, jb->mqd, 1, ch, nl);
        if (rc == -4) return (void *)1;
        if (rc < 0) { ch[0] = rc; ch[1] = 0; }
    }
    return 0;
}

int orc_insert_correct_many(const int *records, char *seq1, char *qual1, const int *lens1, char *seq2, char *qual2,
                            const int *lens2, int64_t stride, int64_t npairs, int action, int min_qual_difference,
                            int *changed, int *newlen, int nthreads) {
    comp_init();
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    cjob_t *jobs = (cjob_t *)malloc(sizeof(cjob_t) * nthreads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    const int64_t per = (npairs + nthreads - 1) / nthreads;
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        cjob_t *jb = &jobs[t];
        jb->rec = records; jb->s1 = seq1; jb->q1 = qual1; jb->s2 = seq2; jb->q2 = qual2; jb->l1 = lens1; jb->l2 = lens2;
        jb->stride = stride; jb->action = action; jb->mqd = min_qual_difference; jb->changed = changed; jb->newlen = newlen;
        jb->lo = per * t < npairs ? per * t : npairs;
        jb->hi = per * (t + 1) < npairs ? per * (t + 1) : npairs;
        if (nthreads == 1) bad |= cjob_main(jb) != 0;
        else pthread_create(&th[t], 0, cjob_main, jb);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) { void *rv = 0; pthread_join(th[t], &rv); bad |= rv != 0; }
    free(th); free(jobs);
    return bad ? -1 : 0;
}

/* ---- quality trimming (atropos/commands/trim/_qualtrim.pyx) --------------------------------------------------
 * quality_trim_index (:7-50): BWA's running-sum rule from both ends.  5' end: s += cutoff_front - q[i] from i = 0,
 * stop at the first negative sum, start = one past the position of the LAST strict maximum of the sums seen; 3' end
 * the same from the back with cutoff_back; an empty or inverted segment is (0, 0).  All in C int like the reference's
 * cdef ints; qualities are characters, q = ord(c) - base. */
void orc_quality_trim_index(const char *qual, int n, int cutoff_front, int cutoff_back, int base, int *start_out,
                            int *stop_out) {
    int s = 0, max_qual = 0, start = 0, stop = n;
    for (int i = 0; i < n; ++i) {
        s += cutoff_front - ((int)(unsigned char)qual[i] - base);
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; start = i + 1; }
    }
    max_qual = 0; s = 0;
    for (int i = n - 1; i >= 0; --i) {
        s += cutoff_back - ((int)(unsigned char)qual[i] - base);
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; stop = i; }
    }
    if (start >= stop) { start = 0; stop = 0; }
    *start_out = start; *stop_out = stop;
}

/* nextseq_trim_index (:53-84): the 3' rule with every 'G' (upper case only: bases[i] == 'G') counted at quality
 * cutoff - 1, i.e. as one unit towards trimming. */
int orc_nextseq_trim_index(const char *bases, const char *qual, int n, int cutoff, int base) {
    int s = 0, max_qual = 0, max_i = n;
    for (int i = n - 1; i >= 0; --i) {
        int q = (int)(unsigned char)qual[i] - base;
        if (bases[i] == 'G') q = cutoff - 1;
        s += cutoff - q;
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; max_i = i; }
    }
    return max_i;
}

/* NEndTrimmer.__call__ (modifiers.py:776-784): re '^N+' and 'N+$' on the sequence as it stands (no case folding).
 * A read of N's only: start_cut = n, end_cut = 0 -- subseq(read, n, 0) is the empty read. */
void orc_n_end_trim(const char *bases, int n, int *start_out, int *stop_out) {
    int a = 0, b = n;
    while (a < n && bases[a] == 'N') ++a;
    while (b > 0 && bases[b - 1] == 'N') --b;
    *start_out = a; *stop_out = b;
}
