/*
 * ra_gpu_wal.h -- batched WAL entry checksums for libra_gpu_batch.
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

#ifdef __cplusplus
}
#endif
#endif
