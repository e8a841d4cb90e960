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
#define ATR_MAX_REF_LEN      128   /* adapter / reference length m */
#define ATR_MAX_READ_LEN     736   /* read length n (23 chunks) */

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
 * :292-297).  d_lens may be NULL (all reads max_len long). */
int atr_pack_reads(const uint8_t *d_ascii, int64_t row_stride, const int32_t *d_lens,
                   int64_t nreads, int max_len, const uint8_t table[256],
                   uint8_t *d_packed, void *stream);

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
 * with. */
int atr_locate_batch(const atr_aligner *a, const uint8_t *d_packed, const int32_t *d_lens,
                     int64_t nreads, int max_len, atr_result *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
