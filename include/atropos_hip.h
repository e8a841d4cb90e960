/*
 * atropos_hip.h -- C ABI of the MI355X (gfx950) adapter-alignment library.
 *
 * This is the drop-in boundary for the one hot path of jdidion/atropos:
 * the Cython module atropos/align/_align.pyx (Aligner.locate, MultiAligner.locate,
 * compare_prefixes) and atropos/align/__init__.py:InsertAligner.match_insert.
 * Each entry point cites the reference interface it replaces.  The reference
 * calls these once per read from Python; this library takes whole batches that
 * are already resident in GPU memory and writes result records back to GPU
 * memory.  No exceptions, no C++ types, no torch types: plain pointers, sizes
 * and integer status codes.  All `d_*` pointers are device pointers on the
 * current HIP device; `stream` is a hipStream_t passed as void* (NULL = the
 * null stream).  Calls are asynchronous with respect to the host.
 *
 * Data layouts
 * ------------
 * ASCII reads   : row-major bytes, read r at d_ascii + r*row_stride, length
 *                 d_lens[r] (or max_len for every read when d_lens is NULL).
 * Packed reads  : 4 bits per base, "tile64" layout.  Reads are grouped in tiles
 *                 of 64 (one wavefront, one read per lane); a read occupies
 *                 nchunks = ceil(max_len/32) chunks of 16 bytes (32 bases);
 *                 chunk c of lane l of tile t lives at byte offset
 *                 ((t*nchunks + c)*64 + l)*16, so one wavefront load of chunk c
 *                 is a single contiguous 1 KiB transaction.  Base j of a read is
 *                 in chunk j/32, little-endian dword (j%32)/8, bits 4*(j%8)..+3.
 *                 Bases past the read's length are 0.  Buffer size:
 *                 atr_packed_bytes().
 * Result record : 8 x int16 = 16 bytes per read (atr_result), refstop == -1
 *                 means "no match" (the reference returns None).
 */
#ifndef ATROPOS_HIP_H
#define ATROPOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (negative = error) */
#define ATR_OK                 0
#define ATR_ERR_INVALID       -1   /* bad argument (the reference raises ValueError) */
#define ATR_ERR_UNSUPPORTED   -2   /* outside the device kernels' envelope, see atr_aligner_create */
#define ATR_ERR_HIP           -3   /* HIP runtime error; atr_last_error() has the text */
#define ATR_ERR_NOMEM         -4   /* the reference raises MemoryError */
#define ATR_ERR_NODEVICE      -5   /* no HIP device visible */

/* alignment flags, atropos/align/__init__.py:17-26 */
#define ATR_START_WITHIN_SEQ1  1
#define ATR_START_WITHIN_SEQ2  2
#define ATR_STOP_WITHIN_SEQ1   4
#define ATR_STOP_WITHIN_SEQ2   8
#define ATR_SEMIGLOBAL        15

/* envelope of the device kernels */
#define ATR_MAX_REF_LEN      128   /* adapter / reference length m of an aligner handle (longer references, up to
                                    * ATR_PAIRS_MAX_LEN: atr_locate_pairs_batch with the same reference on every pair) */
#define ATR_MAX_READ_LEN     736   /* read length n (23 chunks) of the batch pipelines, the per-read calls and the text stages */
#define ATR_MAX_LONG_READ_LEN 32736 /* atr_pack_reads + atr_locate_batch take reads up to here (1023 chunks; the record's
                                    * int16 fields end at 32767): batches with max_len > ATR_MAX_READ_LEN go through the
                                    * full column sweep with a rolling origin base (locate_kernel.hpp) whatever the path */

/* which 256-entry translate table the packed reads must have been built with */
#define ATR_TABLE_DNA15   0   /* equality compare: 15 upper-case IUPAC letters -> their bit codes, rest 0 */
#define ATR_TABLE_ACGT    1   /* _align.pyx:31-44  (_acgt_table)  */
#define ATR_TABLE_IUPAC   2   /* _align.pyx:46-83  (_iupac_table) */
#define ATR_TABLE_CUSTOM  3   /* equality compare over a per-aligner symbol map (arbitrary ASCII reference) */

typedef struct {
    int16_t refstart, refstop, querystart, querystop, matches, errors, aux0, aux1;
} atr_result;

typedef struct atr_aligner atr_aligner;           /* replaces the cdef class Aligner, _align.pyx:121-494 */
typedef struct atr_insert_aligner atr_insert_aligner;   /* replaces InsertAligner, align/__init__.py:178-377 */

/* ---- library / device ---------------------------------------------------- */

int atr_version(void);
int atr_device_count(void);                       /* >= 0, or ATR_ERR_HIP */
const char *atr_last_error(void);                 /* thread-local text of the last ATR_ERR_HIP */

/* ---- translate tables and the 4-bit packer ------------------------------- */

/* Fill `table` with one of the fixed tables above (kind != ATR_TABLE_CUSTOM). */
int atr_translate_table(int kind, uint8_t table[256]);

/* Bytes of a tile64 packed buffer for nreads reads of at most max_len bases. */
size_t atr_packed_bytes(int64_t nreads, int max_len);

/* ASCII -> 4-bit tile64.  `table` is a HOST pointer to the 256-entry translate
 * table (the reference does bytes.translate(table) per read: _align.pyx:243-248,
 * :292-297).  d_lens may be NULL (all reads max_len long).  d_starts may be NULL; otherwise
 * read r is packed from its base d_starts[r] on, i.e. the slice read[start:] of length
 * d_lens[r] - d_starts[r] (what LinkedAdapter.match_to hands to the 3' adapter after
 * removing the 5' match, adapters/__init__.py:683-690).  d_invalid may be NULL;
 * otherwise *d_invalid (device int32, zeroed by the caller) is incremented once per
 * read that contains a byte the table maps to 0 -- the insert aligner needs every base
 * to have a complement (reverse_complement raises KeyError, util/__init__.py:479-482). */
int atr_pack_reads(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                   const int32_t *d_starts, int64_t nreads, int max_len, const uint8_t table[256],
                   uint8_t *d_packed, int32_t *d_invalid, void *stream);

/* Same arguments, same tiles and chunk addresses, but each 16-byte chunk holds the FOUR BIT
 * PLANES of its 32 bases (word p, bit b = bit p of the code of base 32c + b): the layout the
 * insert aligner consumes (atr_insert_match_batch), which compares 32 bases per boolean op. */
int atr_pack_planes(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                   const int32_t *d_starts, int64_t nreads, int max_len, const uint8_t table[256],
                   uint8_t *d_packed, int32_t *d_invalid, void *stream);

/* Counts into *d_count (device int32, zeroed by the caller) the reads of a plane64 buffer whose
 * first min(d_lens[r], d_other_lens[r]) bases hold an uncoded base (a byte the pack table maps
 * to 0).  InsertAligner.match_insert reverse-complements only seq2[:min(len(seq1), len(seq2))]
 * (align/__init__.py:251-259), so that prefix -- not the whole read, which is what d_invalid of
 * the pack calls covers -- is what must have complements.  Either length array may be NULL
 * (max_len). */
int atr_planes_count_uncoded(const uint8_t *d_planes, const int32_t *d_lens, const int32_t *d_other_lens,
                             int64_t nreads, int max_len, int32_t *d_count, void *stream);

/* ---- Aligner (atropos/align/_align.pyx:121-494) -------------------------- */

/* Aligner.__cinit__(reference, max_error_rate, flags, wildcard_ref,
 * wildcard_query, min_overlap, indel_cost)  (_align.pyx:197-208).
 * ATR_ERR_INVALID  : m < 1, min_overlap < 1 or indel_cost < 1 (ValueError, :219-220, :229-230),
 *                    non-ASCII reference byte (UnicodeEncodeError, :243).
 * ATR_ERR_UNSUPPORTED: m > ATR_MAX_REF_LEN, int(e*m) >= 1000, or an indel cost
 *                    large enough to overflow the 12-bit cost field while still
 *                    <= int(e*m) (never the case for the trim command's settings). */
int atr_aligner_create(const char *ref, int m, double max_error_rate, int flags,
                       int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                       atr_aligner **out);
void atr_aligner_destroy(atr_aligner *a);
/* property setters, _align.pyx:214-232 */
int atr_aligner_set_min_overlap(atr_aligner *a, int min_overlap);
int atr_aligner_set_indel_cost(atr_aligner *a, int indel_cost);
/* Which table queries must be packed with (ATR_TABLE_*), and the table itself
 * (meaningful for ATR_TABLE_CUSTOM). */
int atr_aligner_query_table(const atr_aligner *a, uint8_t table[256]);

/* Aligner.locate(query) for a batch (_align.pyx:266-491): d_out[r] receives
 * (refstart, refstop, querystart, querystop, matches, errors) or refstop = -1.
 * d_lens may be NULL.  max_len is the layout parameter the buffer was packed
 * with.  d_work: optional device scratch of atr_locate_work_bytes(nreads) bytes; when
 * given (and the aligner qualifies: START/STOP_WITHIN_SEQ2 set, m <= 64) the filtered
 * pipeline runs -- a bit-parallel pre-pass resolves most reads and the full DP only
 * sweeps a short column window of the rest; results are identical either way.
 * All work is enqueued on `stream` (a hipStream_t; NULL = the default stream) and is complete
 * when the stream reaches the end of the call: the filtered pipeline forks one internal
 * per-thread side stream after its scatter pass and joins it again before it returns (events,
 * no host synchronisation; the pattern is legal inside a stream capture).  d_work must stay
 * valid until then and must not be shared by calls that may overlap. */
size_t atr_locate_work_bytes(int64_t nreads);
/* Diagnostics (bench.py's hard-batch figures; no reference twin): how many reads of the LAST filtered call that used
 * d_work (atr_locate_batch with d_work, atr_locate_planes_batch: n_adapters = 1; atr_linked_match_batch on a long
 * batch: the set's adapter count) the pre-pass left to the exact DP kernels.  Waits for `stream`. */
int atr_locate_work_unresolved(const void *d_work, int64_t nreads, int n_adapters, void *stream, int64_t *out);
int atr_locate_batch(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens,
                     int64_t nreads, int max_len, atr_result *d_out, void *d_work, void *stream);

/* Aligner.locate(query) for a LONG batch of reads that are still ASCII rows (_align.pyx:266-297 takes the string; every
 * real caller starts from text): ONE pre-pass kernel stages a tile's rows through LDS, packs each read into bit planes in
 * registers (the aligner's query table), stores them to d_planes -- atr_packed_bytes(nreads, max_len) bytes, afterwards a
 * complete plane64 batch, e.g. for the next adapter -- and runs the two-pass pre-pass on the registers; then the DP
 * kernels of atr_locate_planes_batch.  Same records as atr_pack_planes + atr_locate_planes_batch, which is also what runs
 * when there is no fused kernel (it exists as a run-time compiled build for this aligner, read length and row stride:
 * no hiprtc, ATR_JIT=0, reads of more than 256 bases, rows wider than 256 bytes).  Envelope: atr_locate_planes_applies.
 * d_work: atr_locate_work_bytes(nreads).  Short batches: atr_locate_ascii_batch. */
int atr_locate_ascii_planes_batch(const atr_aligner *a, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                                  int64_t nreads, int max_len, uint8_t *d_planes, atr_result *d_out, void *d_work, void *stream);

/* Aligner.locate(query) for a batch of reads in the plane64 layout (atr_pack_planes with the aligner's query
 * table; _align.pyx:266-491, same records as atr_locate_batch).  d_lens as in atr_locate_batch (NULL: every read
 * has max_len bases; else the positions of a read past its length must hold code 0, as atr_pack_planes leaves
 * them).  The two-pass pre-pass: exact
 * pieces of the adapter located 32 read positions per boolean op, then Myers' bit-vector sweep over a 64-column
 * window of the flagged reads only, then the exact DP of atr_locate_batch's filtered pipeline on what is left.
 * atr_locate_planes_applies: 1 when this aligner and read length are inside its envelope (3' adapters -- no
 * START_WITHIN_SEQ1 --, up to 40 bases, int(e * m) <= 4, A / C / G / T in the adapter's first 32 bases, reads of
 * 65 .. 320 bases; ragged != 0: a batch with d_lens, max_len the longest read), else 0:
 * pack tile64 and call atr_locate_batch.
 * d_work: atr_locate_work_bytes(nreads) bytes.  ATR_ERR_UNSUPPORTED outside the envelope. */
int atr_locate_planes_applies(const atr_aligner *a, int max_len, int ragged);
/* Build (or load from the code-object cache) the pre-pass kernel SPECIALISED for this aligner and read length: the
 * library compiles its own kernel source once more with the adapter's piece codes, masks and thresholds as
 * constants (hiprtc, found at run time), which removes the per-term scalar dispatch of the generic kernel.  An
 * aligner object of the reference lives for a whole run (adapters/__init__.py:311-322), so the compile (seconds,
 * once per adapter and machine: objects are kept under $ATR_KCACHE_DIR / ~/.cache/atropos_amd) is paid once.
 * atr_locate_planes_batch does this by itself once a handle has seen $ATR_JIT_MIN_READS (2 M) reads; $ATR_JIT=0 turns
 * it off, =1 on for every call.  Same records either way.  ATR_OK: a specialised kernel is ready on the current device;
 * ATR_ERR_UNSUPPORTED: none (outside the envelope, switched off, no hiprtc) -- the generic kernel serves the calls. */
int atr_aligner_prepare(const atr_aligner *a, int max_len, int ragged);
int atr_locate_planes_batch(const atr_aligner *a, const uint8_t *d_planes, const int32_t *d_lens, int64_t nreads,
                            int max_len, atr_result *d_out, void *d_work, void *stream);

/* The same with the kernels named (identical records on every path; for cross-checks and tuning).
 * Short batches -- the 1000 reads the reference's trim command hands over per call
 * (commands/base.py:179), the per-read API -- are bound by the latency of one wavefront, not by
 * throughput: ATR_LOCATE_WAVE gives every read a WAVEFRONT (one DP row per lane, anti-diagonal
 * sweep: n + m steps instead of n * m cells per lane; up to three rows per lane for long references, no scratch). */
#define ATR_LOCATE_AUTO      0   /* what atr_locate_batch does: WAVE for batches of at most 32768 reads where it
                                    applies, else the anchored-prefix band / FILTERED with d_work, else FULL */
#define ATR_LOCATE_FULL      1   /* full-matrix sweep, one read per lane */
#define ATR_LOCATE_FILTERED  2   /* bit-vector pre-pass + window / band DP (needs d_work); FULL where it does not apply */
#define ATR_LOCATE_WAVE      3   /* one read per wavefront */
int atr_locate_batch_path(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens,
                          int64_t nreads, int max_len, atr_result *d_out, void *d_work, int path, void *stream);

/* Short batches straight from ASCII (no atr_pack_reads in between): row r of d_ascii (row_stride bytes apart; base
 * address and stride multiples of four) holds read r, translated inside the wavefront-per-read kernel with the aligner's
 * own query table (_align.pyx:243-248, :292-297).  At most 32768 reads per call (ATR_ERR_UNSUPPORTED beyond: pack the
 * batch and call atr_locate_batch). */
int atr_locate_ascii_batch(const atr_aligner *a, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                           int64_t nreads, int max_len, atr_result *d_out, void *stream);

/* Aligner.locate(query) for ONE read in HOST memory, synchronously: what the module swap of INTEGRATION.md section 1
 * calls per read.  The read goes through a page-locked staging buffer of the calling thread that the kernel reads
 * directly, the record comes back the same way; one launch, one stream synchronisation, no allocation.
 * query: n ASCII bytes (not translated, not terminated); *out: the record (refstop = -1 for None). */
int atr_locate_one(const atr_aligner *a, const char *query, int n, atr_result *out, void *stream);

/* The same for the other two functions the module swap calls once per read pair with the insert aligner
 * (align/__init__.py:250-377): MultiAligner(max_error_rate, flags, min_overlap).locate(ref, query, max_matches)
 * (_align.pyx:593-783; out: up to cap records, *count: the number of hits the reference returns, 0 = None) and
 * compare_prefixes / compare_suffixes(ref, query, wildcard_ref, wildcard_query) (_align.pyx:501-544,
 * align/__init__.py:28-44).  Host strings in, host records out, one launch and one synchronisation each. */
int atr_multi_locate_one(const char *ref, int m, const char *query, int n, double max_error_rate, int flags, int min_overlap,
                         int max_matches, atr_result *out, int cap, int32_t *count, void *stream);
int atr_compare_one(const char *ref, int m, const char *query, int n, int wildcard_ref, int wildcard_query, int suffix,
                    atr_result *out, void *stream);

/* ---- linked adapters: LinkedAdapter.match_to under AdapterCutter._best_match ------------
 * (atropos/adapters/__init__.py:648-690, atropos/commands/trim/modifiers.py:107-122) --------- */

#define ATR_LINKED_MAX_ADAPTERS 4

/* One `-a ^FRONT...BACK` adapter: Adapter(FRONT, PREFIX) and Adapter(BACK, BACK) with what
 * Adapter.match_to wraps around each alignment (adapters/__init__.py:338-400).  The aligners
 * carry sequence, max_error_rate, wildcard flags, min_overlap (Adapter sets aligner.min_overlap
 * = min(min_overlap, len), :285, :313) and indel cost (:316-322).
 *   *_exact_shortcut : the literal startswith / find shortcut is active (`not adapter_wildcards`,
 *                      :351-367): a full-length zero-error occurrence bypasses the acceptance test;
 *   d_*_rmp          : DEVICE table rmp[size * *_rmp_ld + matches] = match_probability(matches, size)
 *                      with its bound *_max_rmp (:393-397), or NULL (max_rmp is None). */
typedef struct {
    const atr_aligner *front;      /* flags == ATR_STOP_WITHIN_SEQ2 (anchored 5') */
    const atr_aligner *back;       /* flags == START_WITHIN_SEQ2 | STOP_WITHIN_SEQ1 | STOP_WITHIN_SEQ2 (regular 3') */
    int front_exact_shortcut, back_exact_shortcut;
    const double *d_front_rmp, *d_back_rmp;
    int front_rmp_ld, back_rmp_ld;
    double front_max_rmp, back_max_rmp;
} atr_linked_adapter;

typedef struct atr_linked_set atr_linked_set;

/* Copies what it needs from the aligners (they may be destroyed afterwards; later setter calls on
 * them are not seen) and uploads one parameter block to the current device.
 * ATR_ERR_UNSUPPORTED: more than ATR_LINKED_MAX_ADAPTERS adapters; a 5' part with m + int(e*m) > 32
 * or int(e*m) > 7; a 3' part outside the filtered pipeline (m > 64) or with START_WITHIN_SEQ1;
 * int(e*m) >= m; aligners that need different translate tables; the literal shortcut together with
 * a wildcard comparison (read wildcards on an ACGT-only adapter).  Callers then match the parts
 * one by one (atr_locate_batch + atr_adapter_postfilter). */
int atr_linked_create(const atr_linked_adapter *adapters, int n_adapters, atr_linked_set **out);
void atr_linked_destroy(atr_linked_set *s);
/* ATR_TABLE_* the reads must be packed with */
int atr_linked_query_table(const atr_linked_set *s);

/* For every read of a tile64 batch (whole reads, packed once): which linked adapter matches, its
 * 5' match and its 3' match on read[front.rstop:].
 *   d_which[2r]     = index of the first adapter whose 5' part matches, or -1 (LinkedAdapter.match_to
 *                     returns None for every adapter);
 *   d_which[2r + 1] = how many 5' parts match (with more than one the reference's _best_match raises
 *                     AttributeError: LinkedMatch has no `matches`, modifiers.py:120);
 *   d_front[r]      = front_match of that adapter as (astart, astop, rstart, rstop, matches, errors);
 *   d_back[r]       = back_match, coordinates relative to read[front.rstop:] as in the reference
 *                     (refstop == -1: None -- also when no 5' part matched).
 * d_work: scratch of atr_linked_work_bytes() bytes; stream semantics as atr_locate_batch. */
size_t atr_linked_work_bytes(const atr_linked_set *s, int64_t nreads);
int atr_linked_match_batch(const atr_linked_set *s, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads,
                           int max_len, int8_t *d_which, atr_result *d_front, atr_result *d_back, void *d_work,
                           void *stream);

/* The same matches with the anchored 5' parts decided AT PACK TIME (round 6; linked_group.hip).  An anchored 5' part
 * only looks at the read's first m + int(e*m) <= 32 bases, and LinkedAdapter.match_to (adapters/__init__.py:671-690)
 * matches the 3' part of the adapter whose 5' part matched on read[front.rstop:]: atr_linked_group_pack reads the ASCII
 * rows once, decides every read's 5' part (d_which, d_front: as atr_linked_match_batch) and packs read[front.rstop:] as
 * bit planes (plane64) into a sub-batch PER ADAPTER -- group g's reads in batch order on consecutive slots:
 *   d_grouped   atr_linked_group_bytes(nreads, max_len) bytes; d_glens[slot] = len - rstop (0 on the padding slots that
 *               fill a group's last tile); d_perm[slot] = the read (-1: padding); d_slot_of[r] = its slot (-1: no 5'
 *               match, nothing packed); all three int32, nreads + 64 * ATR_LINKED_MAX_ADAPTERS entries for glens / perm;
 *   info (HOST, 8 x int64, written before the call returns -- the call waits for `stream`): [g] = reads of group g,
 *               [4 + g] = first tile of its sub-batch.
 * atr_linked_group_match then runs each group through atr_locate_planes_batch's pipeline with ITS 3' aligner (the pre-pass
 * compiled at run time for that aligner; the groups side by side on internal streams, joined before the call's end on
 * `stream`): d_slab[slot] = the raw Aligner.locate record of the slot's read[rstop:]; with d_back != NULL also
 * d_back[r] = back_match in batch order after Adapter.match_to's acceptance test (:386-398), exactly
 * atr_linked_match_batch's d_back.  atr_linked_group_applies: 1 when every 3' aligner of the set is inside the two-pass
 * pre-pass's envelope for ragged reads of at most max_len bases (else use atr_linked_match_batch).
 * table: the 256-byte translate table of atr_linked_query_table's kind.  d_work: atr_linked_group_work_bytes() bytes,
 * the SAME buffer for the pack and the match call of a batch. */
int atr_linked_group_applies(const atr_linked_set *s, int max_len);
size_t atr_linked_group_bytes(int64_t nreads, int max_len);
size_t atr_linked_group_work_bytes(const atr_linked_set *s, int64_t nreads);
int atr_linked_group_pack(const atr_linked_set *s, const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                          int64_t nreads, int max_len, const uint8_t table[256], uint8_t *d_grouped, int32_t *d_glens,
                          int32_t *d_perm, int32_t *d_slot_of, int8_t *d_which, atr_result *d_front, int64_t info[8],
                          void *d_work, void *stream);
int atr_linked_group_match(const atr_linked_set *s, const uint8_t *d_grouped, const int32_t *d_glens, const int64_t info[8],
                           int max_len, const int32_t *d_slot_of, const int8_t *d_which, int64_t nreads, atr_result *d_slab,
                           atr_result *d_back, void *d_work, void *stream);

/* ---- InsertAligner (atropos/align/__init__.py:178-377) -------------------- */

#define ATR_INSERT_MAX_ADAPTER 128   /* adapter length handled by the insert kernel */
#define ATR_INSERT_MAX_READ    320   /* read length handled by the insert kernel */

/* InsertAligner.__init__ arguments (align/__init__.py:206-233) plus the host-built
 * tables.  All pointers are HOST pointers; the library copies what it needs.
 *   rmp_insert[size*rmp_ld + matches]  = match_probability(matches, size, **base_probs)  (:359)
 *   rmp_adapter[size*rmp_ld + matches] = match_probability(matches, size)                (:303-304)
 *   max_mismatch_by_alen[a]            = round(a * max_adapter_mismatch_frac)            (:290)
 * computed by the caller with Python's float/bigint/round semantics (rmp_ld >= ATR_INSERT_MAX_READ + 1,
 * n_mismatch >= ATR_INSERT_MAX_ADAPTER + 1). */
typedef struct {
    const char *adapter1; int alen1;
    const char *adapter2; int alen2;
    double insert_max_rmp, adapter_max_rmp;
    int min_insert_overlap; double max_insert_mismatch_frac;
    int min_adapter_overlap; double max_adapter_mismatch_frac;
    int adapter_check_cutoff;
    int adapter_wildcards, read_wildcards;
    const double *rmp_insert; const double *rmp_adapter; int rmp_ld;
    const int32_t *max_mismatch_by_alen; int n_mismatch;
} atr_insert_config;

/* Allocates two small device tables on the current device (freed by _destroy).
 * ATR_ERR_UNSUPPORTED: an adapter longer than ATR_INSERT_MAX_ADAPTER. */
int atr_insert_aligner_create(const atr_insert_config *cfg, atr_insert_aligner **out);
void atr_insert_aligner_destroy(atr_insert_aligner *a);

/* InsertAligner.match_insert(seq1, seq2) for a batch (align/__init__.py:250-377).
 * Both read sets are PLANE64 buffers (atr_pack_planes) packed with ATR_TABLE_DNA15 and the
 * same max_len (<= ATR_INSERT_MAX_READ).  d_out receives THREE records per pair:
 *   d_out[3p+0] = the insert match tuple (refstop == -1: match_insert returns None),
 *   d_out[3p+1] = Match 1 as (astart, astop, rstart, rstop, matches, errors), astop == -1: None,
 *   d_out[3p+2] = Match 2, likewise. */
int atr_insert_match_batch(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                           const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                           atr_result *d_out, void *stream);

/* The same for SOFT-MASKED reads.  match_insert compares the two reads character by character (lower case is not
 * upper case, _align.pyx:690; reverse_complement keeps the case, util/__init__.py:67-88) but folds the case of
 * the overhangs where compare_prefixes translates them (a wildcard flag set, _align.pyx:521-530).  With
 * ATR_READ_CODES_CASED both reads are packed with atr_case_sensitive_table(): upper-case A C G T N W B D H V keep
 * their DNA15 codes, lower-case a/t, c/g, n take the codes 3/12, 5/10, 6 (closed under the device's complement);
 * M K R Y S have no code in that table.  ATR_ERR_UNSUPPORTED: without wildcard flags an adapter that holds one of
 * M K R Y S or a lower-case letter cannot be told from a soft-masked read base. */
#define ATR_READ_CODES_DNA15 0
#define ATR_READ_CODES_CASED 1
int atr_case_sensitive_table(uint8_t table[256]);
int atr_insert_match_batch_coded(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                                 const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                                 int read_codes, atr_result *d_out, void *stream);

/* InsertAligner.match_insert(seq1, seq2) for ONE pair of reads in host memory, synchronously (the per-pair call of
 * InsertAdapterCutter.__call__, commands/trim/modifiers.py:381-390): n1 / n2 upper-case IUPAC letters each (the caller
 * checks; soft-masked reads take atr_insert_match_batch_coded).  out: the three records of atr_insert_match_batch. */
int atr_insert_match_one(const atr_insert_aligner *a, const char *seq1, int n1, const char *seq2, int n2, atr_result *out,
                         void *stream);

/* ---- MultiAligner.locate, compare_prefixes / compare_suffixes (general) --- */

/* Aligner.enable_debug() / .dpmatrix (_align.pyx:88-119, :259-264, :354-357, :428-431): the DP matrix of
 * ONE locate() call as the reference stores it while debugging -- the costs of exactly the cells its loop
 * computes (Ukkonen's cut-off included), computed with the aligner's true indel cost.  d_packed: one read
 * packed for `a` (tile64, the read is lane 0 of tile 0), n its length.  d_matrix: caller scratch of
 * atr_locate_debug_bytes(a, n) bytes; its first (m + 1) x (n + 1) int32 (row-major, row = reference
 * position) receive the costs, INT32_MIN = never computed (printed as blanks).  d_out: the result record
 * of the same call.  A debugging aid: one lane, not a throughput path. */
size_t atr_locate_debug_bytes(const atr_aligner *a, int n);
int atr_locate_debug(const atr_aligner *a, const uint8_t *d_packed, int n, void *d_matrix, atr_result *d_out,
                     void *stream);

/* MultiAligner(max_error_rate, flags, min_overlap).locate(reference, query, max_matches)
 * (_align.pyx:548-787) for npairs independent pairs of raw ASCII strings (row-major,
 * refs at d_refs + p*ref_stride with length d_ref_lens[p], likewise the queries).
 * d_out holds out_stride records per pair; d_counts[p] receives the number of hits the
 * reference returns (0 == None; hits beyond out_stride are counted but not stored).
 * d_work: caller scratch of atr_multi_locate_work_bytes(npairs, max_ref_len) bytes. */
size_t atr_multi_locate_work_bytes(int64_t npairs, int max_ref_len);
int atr_multi_locate_batch(const uint8_t *d_refs, int64_t ref_stride, const int32_t *d_ref_lens,
                           const uint8_t *d_queries, int64_t query_stride, const int32_t *d_query_lens,
                           int64_t npairs, double max_error_rate, int flags, int min_overlap, int max_matches,
                           int max_ref_len, void *d_work, atr_result *d_out, int32_t *d_counts, int out_stride,
                           void *stream);

/* compare_prefixes(ref, query, wildcard_ref, wildcard_query) (_align.pyx:501-544), or
 * compare_suffixes (align/__init__.py:28-44) when suffix != 0, of ONE reference (host
 * pointer, m <= 1024 raw ASCII bytes) against n queries (device, raw ASCII).  d_lens may
 * be NULL (all queries max_len long). */
int atr_compare_batch(const char *ref, int m, const uint8_t *d_queries, int64_t query_stride,
                      const int32_t *d_lens, int64_t n, int max_len, int wildcard_ref, int wildcard_query,
                      int suffix, atr_result *d_out, void *stream);

/* The same comparison for reads that are already packed for an aligner (tile64, the table of
 * atr_aligner_query_table): `a`'s reference against the first (suffix: last) min(m, len) bases of
 * every read, matching by the aligner's own rule (byte equality without wildcard flags, 4-bit AND
 * with them -- _align.pyx:521-539 picks the same tables as Aligner.__cinit__, :243-248).  This is
 * the branch Adapter.match_to takes for anchored adapters without indels
 * (adapters/__init__.py:370-380); it lets the device pipelines, whose reads only exist packed
 * (FASTQ chunks, the remainders a 5' linked part leaves), run it without an ASCII copy. */
int atr_compare_packed(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens, int64_t nreads,
                       int max_len, int suffix, atr_result *d_out, void *stream);

/* ---- Adapter.match_to post-filter (adapters/__init__.py:386-398) ---------- */

/* Applies, in place on n result records, the acceptance test Adapter.match_to runs on an
 * alignment: size >= min_overlap, errors/size <= max_error_rate (double division) and --
 * when d_rmp is not NULL -- d_rmp[size*rmp_ld + matches] <= max_rmp (d_rmp: DEVICE table of
 * match_probability(matches, size)).  accept_full != 0: a full-length zero-error occurrence
 * (the exact-match shortcut, :351-367) is kept regardless.  Rejected records get refstop -1. */
int atr_adapter_postfilter(atr_result *d_records, int64_t n, int adapter_len, int min_overlap,
                           double max_error_rate, const double *d_rmp, int rmp_ld, double max_rmp,
                           int accept_full, void *stream);

/* ---- ErrorCorrectorMixin.correct_errors (commands/trim/modifiers.py:219-350) ---- */

#define ATR_CORRECT_N             0   /* mismatch_action 'N' */
#define ATR_CORRECT_CONSERVATIVE  1
#define ATR_CORRECT_LIBERAL       2

/* Correct mismatches in the overlap of n read pairs IN PLACE.  d_seq1/d_qual1/d_seq2/
 * d_qual2: raw ASCII, row stride `stride` (qualities may both be NULL); d_insert: 4 x int16
 * per pair = insert_match[0..3]; d_mask (may be NULL) selects the pairs to touch;
 * comp: HOST 256-entry complement table (BASE_COMPLEMENTS; 0 = no complement).
 * d_changed receives 2 x int32 per pair (bases changed in read1, read2; changed[0] == -1:
 * -1 KeyError, -2 IndexError, -3 ValueError: the exception the reference raises), d_newlen the
 * 2 x int32 sequence lengths afterwards (the reference truncates a corrected, longer
 * read 1 to read 2's length when truncate_seqs is set). */
int atr_correct_errors_batch(uint8_t *d_seq1, uint8_t *d_qual1, const int32_t *d_lens1,
                             uint8_t *d_seq2, uint8_t *d_qual2, const int32_t *d_lens2, int64_t stride,
                             const int16_t *d_insert, const uint8_t *d_mask, int64_t n, int max_len,
                             int action, int min_qual_difference, int truncate_seqs, const uint8_t comp[256],
                             int32_t *d_changed, int32_t *d_newlen, void *stream);

/* The correction step of InsertAdapterCutter.__call__ right after the insert match
 * (modifiers.py:397-404): d_insert_records are the 3 records per pair of atr_insert_match_batch;
 * a pair is corrected (correct_errors(read1, read2, insert_match, truncate_seqs=True)) when its
 * insert match exists and has errors > 0.  d_planes1 / d_planes2 (may both be NULL): the plane64
 * buffers the records were computed from (atr_pack_planes, ATR_TABLE_DNA15, layout width
 * planes_max_len) -- with them the kernel finds the disagreeing positions 32 at a time and touches
 * the ASCII matrices only there.  Other arguments as atr_correct_errors_batch. */
int atr_insert_correct_batch(const atr_result *d_insert_records, const uint8_t *d_planes1, const uint8_t *d_planes2,
                             int planes_max_len, uint8_t *d_seq1, uint8_t *d_qual1,
                             const int32_t *d_lens1, uint8_t *d_seq2, uint8_t *d_qual2, const int32_t *d_lens2,
                             int64_t stride, int64_t n, int max_len, int action, int min_qual_difference,
                             const uint8_t comp[256], int32_t *d_changed, int32_t *d_newlen, void *stream);

/* InsertAdapterCutter.__call__'s first two steps as ONE call (modifiers.py:385-404): the insert match of
 * atr_insert_match_batch (records to d_out, 3 per pair) and, for every pair whose insert match has errors > 0,
 * correct_errors(read1, read2, insert_match, truncate_seqs=True) in place on the ASCII matrices -- what
 * atr_insert_match_batch followed by atr_insert_correct_batch(d_out, d_packed1, d_packed2, max_len, ...) computes,
 * bit for bit, but with the reads' planes streamed once: the disagreeing positions are the mask words the match
 * counted.  d_packed1 / d_packed2: plane64 batches (ATR_TABLE_DNA15) of layout width max_len; d_lens1 / d_lens2
 * (may be NULL: every read has max_len bases) are the lengths of BOTH the packed reads and the matrix rows.
 * Batches the fused kernel does not take (more than 256 or fewer than 97 bases of layout width, adapters of
 * more than 64 bases, a row pitch `stride` of 64 MB or more) run the two kernels one after the other inside this call. */
int atr_insert_match_correct_batch(const atr_insert_aligner *a, const uint8_t *d_packed1, const int32_t *d_lens1,
                                   const uint8_t *d_packed2, const int32_t *d_lens2, int64_t npairs, int max_len,
                                   atr_result *d_out, uint8_t *d_seq1, uint8_t *d_qual1, uint8_t *d_seq2,
                                   uint8_t *d_qual2, int64_t stride, int action, int min_qual_difference,
                                   const uint8_t comp[256], int32_t *d_changed, int32_t *d_newlen, void *stream);

/* ---- Aligner.locate with a per-pair reference (MergeOverlapping) ----------- */

/* `Aligner(ref_p, max_error_rate, flags, wildcard_ref, wildcard_query, min_overlap,
 * indel_cost).locate(query_p)` (_align.pyx:197-491) for npairs independent pairs: the
 * aligner MergeOverlapping constructs per read pair with reference = reverse complement of
 * read 2 and query = read 1 (commands/trim/modifiers.py:889-894).  Both sides are 4-bit
 * tile64 batches (atr_pack_reads / atr_pack_records) packed with the tables Aligner would
 * translate with: ATR_TABLE_DNA15 on both sides for the literal compare (every base must be
 * one of the 15 upper-case IUPAC letters), ATR_TABLE_ACGT / ATR_TABLE_IUPAC per wildcard flag
 * otherwise (_align.pyx:243-248, :292-297).  revcomp_ref != 0: the reference of a pair is the
 * reverse complement of the packed sequence (util/__init__.py:479-482).  d_*_lens may be NULL.
 * ATR_ERR_UNSUPPORTED: a side longer than ATR_PAIRS_MAX_LEN (longer than 255 when the flags lack
 * ATR_STOP_WITHIN_SEQ2: only then must the DP cell count matches rather than mismatches) or
 * int(e * m) >= 256. */
#define ATR_PAIRS_MAX_LEN 320
int atr_locate_pairs_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len, int revcomp_ref,
                           const uint8_t *d_query_packed, const int32_t *d_query_lens, int query_max_len,
                           int64_t npairs, double max_error_rate, int flags, int wildcard_ref, int wildcard_query,
                           int min_overlap, int indel_cost, atr_result *d_out, void *stream);

/* The same for pairs with a side of more than ATR_PAIRS_MAX_LEN bases (up to ATR_MAX_LONG_READ_LEN; the reference has
 * no limit, _align.pyx:266-291): 64-bit cells, the DP column in d_work (atr_locate_pairs_long_work_bytes(npairs,
 * ref_max_len) bytes), one pair per lane over the whole matrix -- a fallback, not a throughput path.  Any lengths
 * are accepted.  ATR_ERR_UNSUPPORTED: max_error_rate * ref_max_len > 32 000. */
size_t atr_locate_pairs_long_work_bytes(int64_t npairs, int ref_max_len);
int atr_locate_pairs_long_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len, int revcomp_ref,
                                const uint8_t *d_query_packed, const int32_t *d_query_lens, int query_max_len,
                                int64_t npairs, double max_error_rate, int flags, int wildcard_ref, int wildcard_query,
                                int min_overlap, int indel_cost, atr_result *d_out, void *d_work, void *stream);

/* The same with a per-pair lower bound on the alignment's matches (d_need, device, may be NULL = 1 everywhere):
 * MergeOverlapping only looks at an alignment when `matches >= min_overlap` (modifiers.py:896-897, min_overlap
 * from :870-874), so a pair whose reference alignment has fewer matches than d_need[p] may be reported as
 * refstop -1 instead -- which lets the library stop after its cost pass for pairs that cannot overlap that far.
 * Pairs whose alignment has at least d_need[p] matches get the reference's record, bit for bit.
 * With STOP_WITHIN_SEQ2, indel cost 1 and no wildcard flag (both flag sets of MergeOverlapping) the batch runs
 * through: exact costs of all candidate cells by Myers' bit-vector recurrence, the candidates that can still be
 * the reference's choice, and the packed-word DP on the band of diagonals that holds all their optimal paths;
 * other settings -- and pairs outside that pipeline's envelope -- sweep the whole matrix. */
int atr_locate_pairs_need_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len, int revcomp_ref,
                                const uint8_t *d_query_packed, const int32_t *d_query_lens, int query_max_len,
                                int64_t npairs, double max_error_rate, int flags, int wildcard_ref, int wildcard_query,
                                int min_overlap, int indel_cost, const int32_t *d_need, atr_result *d_out, void *stream);

/* The same with the kernel family named (identical records up to the `need` rule; cross-checks, tuning).  Short
 * batches -- the 1000 pairs MergeOverlapping sees per call of the reference's trim command -- take ATR_PAIRS_WAVE:
 * one pair per wavefront, 64 lanes x up to 5 rows each (references of up to 319 bases), anti-diagonal sweep. */
#define ATR_PAIRS_AUTO  0   /* WAVE for at most 32768 pairs where it applies, else FAST where it pays, else FULL */
#define ATR_PAIRS_FULL  1   /* full-matrix sweep, one pair per lane (atr_locate_pairs_full_batch) */
#define ATR_PAIRS_FAST  2   /* costs by bit-vector, threats, banded payload; FULL for what is outside its envelope */
#define ATR_PAIRS_WAVE  3   /* one pair per wavefront; ATR_ERR_UNSUPPORTED for references of more than 319 bases */
int atr_locate_pairs_path_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len,
                                int revcomp_ref, const uint8_t *d_query_packed, const int32_t *d_query_lens,
                                int query_max_len, int64_t npairs, double max_error_rate, int flags,
                                int wildcard_ref, int wildcard_query, int min_overlap, int indel_cost,
                                const int32_t *d_need, int path, atr_result *d_out, void *stream);

/* The same for ONE pair in host memory, synchronously -- what MergeOverlapping does per read pair through the module swap
 * (a new Aligner per pair, commands/trim/modifiers.py:889-894) when the reads are longer than an aligner handle's 128
 * bases.  ref_codes / query_codes: the two strings translated to 4-bit codes, one per byte, with the tables named
 * above (what atr_pack_reads would pack); references of up to 319 codes. */
int atr_locate_pair_one(const uint8_t *ref_codes, int m, int revcomp_ref, const uint8_t *query_codes, int n,
                        double max_error_rate, int flags, int wildcard_ref, int wildcard_query, int min_overlap,
                        int indel_cost, atr_result *out, void *stream);
/* atr_locate_pairs_batch by the full-matrix sweep alone (same records; the checker of the pipeline above). */
int atr_locate_pairs_full_batch(const uint8_t *d_ref_packed, const int32_t *d_ref_lens, int ref_max_len, int revcomp_ref,
                                const uint8_t *d_query_packed, const int32_t *d_query_lens, int query_max_len,
                                int64_t npairs, double max_error_rate, int flags, int wildcard_ref, int wildcard_query,
                                int min_overlap, int indel_cost, atr_result *d_out, void *stream);

/* ---- device-resident FASTQ batch (io/_seqio.pyx:163-245, io/seqio.py:686-700) ---- */

/* One chunk of FASTQ text (whole records, < 4 GiB, 16-byte aligned, ending in a line end) is
 * uploaded as raw bytes; the records are found, validated, trimmed and formatted on the
 * device.  Every trimming modifier of the single-end path only clips read ends, so its
 * state is the kept interval [d_begin[r], d_end[r]) of each record's sequence line. */
typedef struct {
    uint32_t name_off, name_len;   /* description after '@', line end stripped (FastqReader, _seqio.pyx:208-216) */
    uint32_t seq_off, seq_len;     /* :217-218 */
    uint32_t qual_off, qual_len;   /* :239-243 */
    uint32_t flags;                /* bit 0: the '+' line repeats the description (name2 = name, :228-236);
                                    * bit 1 (set by a caller that re-pointed name_off at a rewritten name -- LengthTagModifier,
                                    * SuffixRemover, PrefixSuffixAdder, commands/trim/modifiers.py:652-695): the '+' line's text
                                    * is the flags >> 8 bytes at `reserved`, as FastqFormat prints the name2 the file had */
    uint32_t reserved;             /* offset of that text (flags bit 1), else unused */
} atr_fastq_record;

#define ATR_FASTQ_ERR_AT      1    /* FormatError "... expected to start with '@'" (:209-211) */
#define ATR_FASTQ_ERR_PLUS    2    /* "... expected to start with '+'" (:224-227) */
#define ATR_FASTQ_ERR_NAME2   3    /* "Sequence descriptions in the FASTQ file don't match" (:229-235) */
#define ATR_FASTQ_ERR_LENGTH  4    /* "length of quality sequence ... and length of read ... do not match" (:33-43) */

/* Count the lines of the chunk: *d_nlines (device int64).  Line ends follow Python's universal
 * newlines, which is how the reference reads its input (text mode): "\n", "\r\n" and a lone
 * "\r".  d_work: caller scratch of atr_fastq_work_bytes(nbytes) bytes, to be handed unchanged
 * to atr_fastq_index.  The buffer must be readable up to the next multiple of 16 plus one byte. */
size_t atr_fastq_work_bytes(int64_t nbytes);
int atr_fastq_count_lines(const uint8_t *d_bytes, int64_t nbytes, void *d_work, int64_t *d_nlines, void *stream);

/* Positions of the line ends (d_line_ends[nlines], uint32: the last byte of each terminator)
 * and the descriptors of the nlines / 4 records.  *d_error (device int64) receives min over
 * invalid records of (record * 8 + ATR_FASTQ_ERR_*), or INT64_MAX. */
int atr_fastq_index(const uint8_t *d_bytes, int64_t nbytes, const void *d_work, uint32_t *d_line_ends, int64_t nlines,
                    atr_fastq_record *d_records, int64_t *d_error, void *stream);

/* 4-bit tile64 pack (as atr_pack_reads) of sequence[begin:end] of every record; d_begin /
 * d_end may be NULL (the whole sequence line).  d_lens receives the packed lengths.
 * planes != 0: plane64 layout (atr_pack_planes) instead of tile64. */
int atr_pack_records(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                     const int32_t *d_end, int64_t nreads, int max_len, const uint8_t table[256], int planes,
                     uint8_t *d_packed, int32_t *d_lens, int32_t *d_invalid, void *stream);

/* Interval updates, each the device twin of one reference modifier applied to
 * read[begin:end] (records with begin == end are left alone, as `if len(read) == 0`):
 *   atr_clip_batch          UnconditionalCutter.clip(front, back <= 0)   (modifiers.py:565-585, :69-82)
 *   atr_quality_trim_batch  QualityTrimmer / quality_trim_index          (:748-764, _qualtrim.pyx:7-48)
 *                           nextseq != 0: NextseqQualityTrimmer with cutoff_back  (:732-746, _qualtrim.pyx:51-84)
 *   atr_nend_trim_batch     NEndTrimmer                                   (:766-784)
 *                           (d_unmasked_begin/end, may be NULL: AdapterCutter action 'mask' -- the
 *                           bases of [begin, end) outside [unmasked_begin, unmasked_end) read as 'N',
 *                           modifiers.py:155-172; the same pair is honoured by the filter and the writer)
 *   atr_match_trim_batch    Adapter.trimmed(match) for the best match of the round: front != 0
 *                           removes read[:querystop] (adapters/__init__.py:453-457), else keeps
 *                           read[:querystart] (:459-471); d_front[r] selects per read (may be NULL
 *                           = guess rstart == 0, align/__init__.py:108-114 -- 'anywhere' adapters);
 *                           only records with refstop >= 0 and d_active[r] != 0 are touched,
 *                           d_active[r] is cleared for reads without a match (AdapterCutter stops
 *                           at the first failed round, modifiers.py:133-139) and d_matched[r] set. */
int atr_clip_batch(const atr_fastq_record *d_records, int32_t *d_begin, int32_t *d_end, int64_t n, int front,
                   int back, void *stream);
int atr_quality_trim_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, int32_t *d_begin,
                           int32_t *d_end, int64_t n, int cutoff_front, int cutoff_back, int base, int nextseq,
                           void *stream);
int atr_nend_trim_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, int32_t *d_begin,
                        int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end, int64_t n,
                        void *stream);
int atr_match_trim_batch(const atr_result *d_matches, const uint8_t *d_front, int default_front, int32_t *d_begin,
                         int32_t *d_end, uint8_t *d_active, uint8_t *d_matched, int64_t n, void *stream);

/* Filters (commands/trim/filters.py:109-184) in the order trim/__init__.py:566-601 installs
 * them; d_dest[r] = the first that fires.  min_len <= 0 / max_len < 0 / max_n < 0: off. */
#define ATR_DEST_KEEP        0
#define ATR_DEST_TOO_SHORT   1
#define ATR_DEST_TOO_LONG    2
#define ATR_DEST_TOO_MANY_N  3
#define ATR_DEST_TRIMMED     4
#define ATR_DEST_UNTRIMMED   5
int atr_read_filter_batch(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                          const int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end,
                          const uint8_t *d_matched, int64_t n, int min_len, int max_len, double max_n,
                          int discard_trimmed, int discard_untrimmed, uint8_t *d_dest, uint8_t *d_fail_mask,
                          void *stream);

/* Paired-end filtering (PairedWrapper, commands/trim/filters.py:66-90): d_fail_mask1/2 are the
 * per-read masks atr_read_filter_batch writes (bit d: the filter with destination d fires;
 * d_dest may be NULL there); a filter fires for the pair when it fires for at least
 * min_affected (1: --pair-filter=any, 2: both) of its reads; d_dest[p] = the first such. */
int atr_pair_filter_batch(const uint8_t *d_fail_mask1, const uint8_t *d_fail_mask2, int64_t n, int min_affected,
                          uint8_t *d_dest, void *stream);

/* InsertAdapterCutter.__call__ after the alignments (commands/trim/modifiers.py:391-496):
 * d_insert = the 3 records per pair of atr_insert_match_batch; d_fallback1/2 = the
 * Adapter.match_to records of read 1 / read 2 (used for pairs without an insert match); the
 * kept intervals of both reads are updated in place, d_matched1/2 receive `read.match is not
 * None`.  symmetric: mirror a lone adapter match onto the other read (:419-446); trim_action:
 * 0 = --no-trim (match only), 1 = trim, 2 = mask (the trimmed end goes to d_unmasked_end*,
 * begin/end stay).  correct_action: -1 = no error correction, else ATR_CORRECT_* -- the
 * overlap of the pairs the reference would correct (:397-446) is corrected IN PLACE in the two
 * FASTQ chunks (bases and qualities, correct_errors(..., truncate_seqs=True), :219-350) before
 * trimming; d_corrected (may be NULL) receives the 2 x int32 changed-base counts per pair and
 * *d_error min over failing pairs of (pair * 8 + code), code 1 KeyError, 2 IndexError,
 * 3 ValueError (the exception the reference raises), or INT64_MAX. */
int atr_insert_plan_batch(const atr_result *d_insert, const atr_result *d_fallback1, const atr_result *d_fallback2,
                          uint8_t *d_bytes1, const atr_fastq_record *d_records1, uint8_t *d_bytes2,
                          const atr_fastq_record *d_records2, int32_t *d_begin1, int32_t *d_end1, int32_t *d_begin2,
                          int32_t *d_end2, int32_t *d_unmasked_end1, int32_t *d_unmasked_end2, int64_t n,
                          int min_insert_len, int symmetric, int trim_action, int correct_action, int min_qual_difference,
                          const uint8_t comp[256], uint8_t *d_matched1, uint8_t *d_matched2, int32_t *d_corrected,
                          int64_t *d_error, void *stream);

/* FastqFormat.format_entry (io/seqio.py:690-699) of every record with d_dest[r] == dest, in
 * input order, into d_out.  First call with d_out == NULL: d_offsets[n + 1] (device int64)
 * receives the exclusive prefix sums of the formatted sizes (d_offsets[n] = total bytes);
 * second call with d_out of that many bytes writes the text.  d_work: scratch of
 * atr_fastq_emit_work_bytes(n) bytes.  record_bytes_hint: average bytes per input record (0:
 * unknown) -- only picks how many records a wave stages at once. */
size_t atr_fastq_emit_work_bytes(int64_t n);
int atr_fastq_emit(const uint8_t *d_bytes, const atr_fastq_record *d_records, const int32_t *d_begin,
                   const int32_t *d_end, const int32_t *d_unmasked_begin, const int32_t *d_unmasked_end,
                   const uint8_t *d_dest, int dest, int64_t n, int record_bytes_hint, int64_t *d_offsets, void *d_work,
                   uint8_t *d_out, void *stream);

/* MergeOverlapping.__call__ after the alignment (commands/trim/modifiers.py:864-931), two calls:
 *
 * atr_merge_plan_batch: d_align = one record per pair of Aligner(reverse_complement(read2), error_rate,
 * flags).locate(read1) on the kept intervals (atr_locate_pairs_batch with revcomp_ref; refstop -1 for pairs
 * without an alignment and for pairs that were too short to be aligned, :876-877), d_need[p] = the pair's
 * minimum overlap (:870-874).  d_kind[p] receives 0 (the pair stays a pair) or which of the four shapes
 * of :904-921 the merged read has; d_offsets[n + 1] the exclusive prefix sums of the merged records'
 * formatted sizes (FastqFormat.format_entry of read 1 with the merged sequence; d_offsets[n] = total).
 * *d_error: min over pairs with an "Invalid alignment" (:922-926) of pair * 8 + 4, or INT64_MAX.
 * d_work: scratch of atr_merge_work_bytes(n) bytes.
 *
 * atr_merge_emit_batch: writes the text of the merged records into d_out (d_offsets[n] bytes).
 * correct_action: -1, or ATR_CORRECT_* = MergeOverlapping's mismatch_action: pairs whose alignment has
 * errors and for which d_insert_matched[p] == 0 (may be NULL: none matched; callers also set a non-zero
 * byte for pairs with read.corrected > 0, which correct_errors leaves alone, :232-233) are corrected IN PLACE in
 * the two chunks first (correct_errors(read1, read2, alignment), :900-902) -- except that the mate's
 * bases are the uncorrected ones, as in the reference, which reverse-complements read 2 before (:887).
 * d_corrected (may be NULL): 2 x int32 changed-base counts per pair; *d_error (as initialised by the
 * plan call) additionally receives pair * 8 + {1 KeyError, 2 IndexError, 3 ValueError} of a failing
 * correction. */
size_t atr_merge_work_bytes(int64_t n);
int atr_merge_plan_batch(const atr_result *d_align, const int32_t *d_need, const atr_fastq_record *d_records1,
                         const int32_t *d_begin1, const int32_t *d_end1, const int32_t *d_begin2, const int32_t *d_end2,
                         int64_t n, uint8_t *d_kind, int64_t *d_offsets, void *d_work, int64_t *d_error, void *stream);
int atr_merge_emit_batch(const atr_result *d_align, const uint8_t *d_kind, const uint8_t *d_insert_matched,
                         uint8_t *d_bytes1, const atr_fastq_record *d_records1, uint8_t *d_bytes2,
                         const atr_fastq_record *d_records2, const int32_t *d_begin1, const int32_t *d_end1,
                         const int32_t *d_begin2, const int32_t *d_end2, int64_t n, int correct_action,
                         int min_qual_difference, const uint8_t comp[256], const int64_t *d_offsets, int32_t *d_corrected,
                         int64_t *d_error, uint8_t *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
