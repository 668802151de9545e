/*
 * ra_gpu_wal.h -- batched WAL entry checksums, record framing and recovery validation for
 * libra_gpu_batch.
 *
 * The reference checksums every WAL record with Adler-32 over the entry as it is framed on disk,
 *
 *     Entry    = [<<Idx:64/unsigned, Term:64/unsigned>> | EntryData],
 *     Checksum = erlang:adler32(Entry)                       (src/ra_log_wal.erl:528-534)
 *     Record   = [HeaderData, <<Checksum:32/integer, EntryDataLen:32/unsigned>> | Entry]
 *
 * when writing (compute_checksums = true) and again for every record it reads back
 * (validate_checksum, src/ra_log_wal.erl:861, 873, 1028).  erlang:adler32/1 is zlib's Adler-32
 * (RFC 1950 section 8.2).  This entry point computes the checksums of a whole batch of entries
 * whose payload bytes are resident in device memory; framing, file I/O, fsync and the
 * durability decision stay on the host, exactly as in ra_log_wal.
 *
 * Plain C ABI: device pointers and sizes only.
 */
#ifndef RA_GPU_WAL_H
#define RA_GPU_WAL_H
#include "ra_gpu_batch.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One WAL entry of the batch, 32 bytes.  The payload is data[data_offset .. data_offset+data_len)
 * of the batch's data buffer (any alignment, any length including 0). */
typedef struct rgb_wal_entry {
  uint64_t index;        /* Idx  -- framed big endian, as <<Idx:64/unsigned>>  */
  uint64_t term;         /* Term -- framed big endian, as <<Term:64/unsigned>> */
  uint64_t data_offset;
  uint32_t data_len;     /* EntryDataLen */
  uint32_t _pad;
} rgb_wal_entry;

/* d_checksums[i] = adler32(<<index:64, term:64, payload/binary>>) for the n entries of d_entries.
 * d_data must be readable for 16-byte aligned accesses around every payload (allocate the buffer
 * 16 bytes longer than its content; hipMalloc aligns its start).  Enqueued on `stream` (NULL = the
 * context's stream); no synchronisation.  Replaces erlang:adler32/1 at src/ra_log_wal.erl:532
 * (write path) and :861/:873/:1028 (recovery: compare with the stored Checksum on the host). */
int rgb_wal_adler32_device(rgb_ctx *ctx, const void *d_entries, uint32_t n, const void *d_data,
                           uint64_t data_bytes, void *d_checksums, void *stream);

/* Host-buffer form (what the NIF binds): entries and the packed payload bytes come from host
 * memory, the n checksums go back to host memory.  Stages through device buffers the context
 * keeps (grown on demand), synchronises before returning: PCIe-inclusive. */
int rgb_wal_adler32(rgb_ctx *ctx, const rgb_wal_entry *entries, uint32_t n, const void *data,
                    uint64_t data_bytes, uint32_t *checksums);

/* ---- record framing (write path) ---------------------------------------------------------
 *
 * The reference writes, per entry (src/ra_log_wal.erl:513-537),
 *
 *     Record = [HeaderData, <<Checksum:32/integer, EntryDataLen:32/unsigned>>,
 *               <<Idx:64/unsigned, Term:64/unsigned>> | EntryData]
 *
 * where HeaderData is what serialize_header/3 (:482-499) returns: 3 bytes <<Trunc:1, 1:1, IdRef:22>>
 * for a writer already named in this file, or <<Trunc:1, 0:1, IdRef:22, IdDataLen:16, UId/binary>>
 * on its first appearance.  The writer-name cache is a map of binaries and stays on the host; the
 * host hands the HeaderData bytes over verbatim (inside the batch's data buffer) and the device
 * produces the batch's contiguous on-disk bytes -- checksum, lengths, big-endian cursors and the
 * payload copy -- in one pass over the payload.  Writing the bytes to the file, fsync and the
 * `written` notifications stay in ra_log_wal. */
typedef struct rgb_wal_record {
  uint64_t index;        /* Idx  */
  uint64_t term;         /* Term */
  uint64_t data_offset;  /* EntryData = data[data_offset .. +data_len) */
  uint32_t data_len;     /* EntryDataLen */
  uint32_t hdr_len;      /* byte_size(HeaderData): 3, or 5 + IdDataLen */
  uint64_t hdr_offset;   /* HeaderData = data[hdr_offset .. +hdr_len) */
  uint64_t out_offset;   /* where the record starts in the output (rgb_wal_layout fills it) */
} rgb_wal_record;

#define RGB_WAL_NO_CHECKSUMS 1u   /* compute_checksums = false: Checksum = 0 (src/ra_log_wal.erl:531-534) */

/* Host helper: out_offset of every record when they are written back to back from `base`
 * (DataSize = HeaderLen + 24 + EntryDataLen each); returns the offset behind the last one. */
uint64_t rgb_wal_layout(rgb_wal_record *records, uint32_t n, uint64_t base);

/* d_out[out_offset ..) = the framed record, for the n records of d_records.  d_checksums (may be
 * NULL) receives the n checksums as well.  d_data and d_out may have any alignment (payloads are read and written
 * in 16-byte pieces at their own byte alignment), d_out must hold the highest out_offset + record size; records may
 * not overlap.  Only the payload's own bytes are read, only the record's own bytes are written.  `flags`:
 * RGB_WAL_NO_CHECKSUMS.  Enqueued on `stream`, no synchronisation. */
int rgb_wal_frame_device(rgb_ctx *ctx, const void *d_records, uint32_t n, const void *d_data,
                         uint64_t data_bytes, void *d_out, uint64_t out_bytes, void *d_checksums,
                         uint32_t flags, void *stream);

/* Host-buffer form: records (out_offset filled, e.g. by rgb_wal_layout from 0) and data in host
 * memory, the framed bytes [0, out_bytes) back into `out`.  Synchronises; PCIe-inclusive. */
int rgb_wal_frame(rgb_ctx *ctx, const rgb_wal_record *records, uint32_t n, const void *data,
                  uint64_t data_bytes, void *out, uint64_t out_bytes, uint32_t flags);

/* ---- recovery (read path) ---------------------------------------------------------------
 *
 * recover_records/5 (src/ra_log_wal.erl:884-984) walks a WAL file record by record; every record of
 * a known writer is validated with validate_checksum/4 (:1022-1033) before it is recovered.  The
 * walk is a dependent chain over a few header bytes per record and is host code here
 * (rgb_wal_scan); the checksums of all records it finds are one device batch (rgb_wal_validate). */
typedef struct rgb_wal_scanned {
  uint64_t index;
  uint64_t term;
  uint64_t data_offset;  /* EntryData inside the file bytes */
  uint32_t data_len;
  uint32_t checksum;     /* the stored Checksum */
  uint64_t uid_offset;   /* UId bytes inside the file (first appearance only) */
  uint32_t id_ref;       /* IdRef:22 */
  uint16_t uid_len;      /* IdDataLen (first appearance only) */
  uint8_t  trunc;        /* Trunc:1 */
  uint8_t  flags;        /* RGB_WAL_REC_* */
  uint64_t next_offset;  /* file offset behind this record (`Rest`) */
} rgb_wal_scanned;

#define RGB_WAL_REC_FIRST    1u  /* long header: first appearance of the writer in this file */
#define RGB_WAL_REC_VALIDATE 2u  /* to be checksum-validated; the caller clears it for writers that are
                                    not registered (ra_directory:is_registered_uid, :902).  The scan
                                    cannot ask the directory, so it treats every long header as
                                    introducing its IdRef: for a writer the caller finds unregistered
                                    the reference leaves the IdRef out of its cache (:902-904, :929-931) and
                                    skips that writer's later short-header records (:968-971) — the
                                    caller must therefore clear VALIDATE on EVERY later record of that
                                    id_ref in this file, not only on the RGB_WAL_REC_FIRST one */
#define RGB_WAL_REC_UNKNOWN  4u  /* short header whose IdRef was never introduced: skipped (:968-971) */

#define RGB_WAL_END_ZEROS 0u     /* all-zero record: end of a pre-allocated file (:877-883) */
#define RGB_WAL_END_DATA  1u     /* not enough bytes left for a whole record: end of file (:973-984) */
#define RGB_WAL_END_CAP   2u     /* `cap` records written, more may follow from *consumed */

/* Parse `bytes` (a whole WAL file including its 5-byte "RAWA", version 1 header, :826-835; an
 * unknown header gives RGB_E_INVAL) into at most `cap` records.  out = NULL counts the records of the
 * file into *n_out without storing them (size the array, then scan again).  Pure host code. */
int rgb_wal_scan(const void *bytes, uint64_t n_bytes, rgb_wal_scanned *out, uint32_t cap,
                 uint32_t *n_out, uint64_t *consumed, uint32_t *end);

#define RGB_WAL_CLEAN        0u  /* every validated record matched */
#define RGB_WAL_DROPPED_LAST 1u  /* record *n_ok failed its checksum and is the last one: dropped,
                                    recovery resumes (is_last_record/3, :994-1004) */
#define RGB_WAL_CORRUPT      2u  /* record *n_ok failed and data follows: the reference throws
                                    wal_checksum_validation_failure (:1006-1008) */

/* Validate the scanned records of a file in order (stored Checksum 0 = "checksum not used", always
 * ok, :1022-1024).  *n_ok = number of leading records that are good, *status = RGB_WAL_*. */
int rgb_wal_validate(rgb_ctx *ctx, const void *bytes, uint64_t n_bytes, const rgb_wal_scanned *recs,
                     uint32_t n, uint32_t *n_ok, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif
