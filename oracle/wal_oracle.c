/*
 * wal_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the WAL entry checksum of rabbitmq/ra (src/ra_log_wal.erl:528-534 write
 * path, :861/:873/:1028 validation):  erlang:adler32([<<Idx:64, Term:64>> | EntryData]).
 * erlang:adler32/1 is zlib's Adler-32; zlib is not part of /root/reference, so the algorithm is
 * restated from its published definition (RFC 1950 section 8.2 / 9, the byte-at-a-time loop) and
 * pinned in tests/ by RFC 1950's own structure (adler32("") = 1), the well-known vector
 * adler32("Wikipedia") = 0x11E60398 and Python's zlib.adler32 on random inputs.
 */
#include <stddef.h>
#include <stdint.h>

#define ADLER_BASE 65521u   /* largest prime smaller than 65536 (RFC 1950) */

/* RFC 1950 section 9: update a running Adler-32 with buf[0..len) */
uint32_t ora_adler32_update(uint32_t adler, const uint8_t *buf, size_t len) {
  uint32_t s1 = adler & 0xFFFFu, s2 = (adler >> 16) & 0xFFFFu;
  for (size_t i = 0; i < len; i++) {
    s1 = (s1 + buf[i]) % ADLER_BASE;
    s2 = (s2 + s1) % ADLER_BASE;
  }
  return (s2 << 16) | s1;
}

/* Checksum = erlang:adler32([<<Idx:64/unsigned, Term:64/unsigned>> | EntryData]) */
uint32_t ora_wal_entry_checksum(uint64_t index, uint64_t term, const uint8_t *data, uint32_t len) {
  uint8_t prefix[16];
  for (int k = 0; k < 8; k++) {
    prefix[k] = (uint8_t)(index >> (8 * (7 - k)));       /* big endian */
    prefix[8 + k] = (uint8_t)(term >> (8 * (7 - k)));
  }
  uint32_t a = ora_adler32_update(1u, prefix, 16);
  return ora_adler32_update(a, data, len);
}

/* batch form used by the tests and tools: entries = (index, term, data_offset, data_len, pad) */
void ora_wal_checksums(const uint64_t *idx_term_off, const uint32_t *lens, uint32_t n,
                       const uint8_t *data, uint32_t *out) {
  for (uint32_t i = 0; i < n; i++)
    out[i] = ora_wal_entry_checksum(idx_term_off[3 * i], idx_term_off[3 * i + 1],
                                    data + idx_term_off[3 * i + 2], lens[i]);
}

/* Record = [HeaderData, <<Checksum:32/integer, EntryDataLen:32/unsigned>>,
 *           <<Idx:64/unsigned, Term:64/unsigned>> | EntryData]      (src/ra_log_wal.erl:513-537)
 * written at out + out_offset; returns the record's size (HeaderLen + 24 + EntryDataLen, :526).
 * compute_checksums = false stores Checksum 0 (:531-534). */
uint64_t ora_wal_frame_record(uint64_t index, uint64_t term, const uint8_t *hdr, uint32_t hdr_len,
                              const uint8_t *data, uint32_t len, int compute_checksums, uint8_t *out) {
  uint32_t cs = compute_checksums ? ora_wal_entry_checksum(index, term, data, len) : 0u;
  uint8_t *p = out;
  for (uint32_t k = 0; k < hdr_len; k++) *p++ = hdr[k];
  for (int k = 3; k >= 0; k--) *p++ = (uint8_t)(cs >> (8 * k));
  for (int k = 3; k >= 0; k--) *p++ = (uint8_t)(len >> (8 * k));
  for (int k = 7; k >= 0; k--) *p++ = (uint8_t)(index >> (8 * k));
  for (int k = 7; k >= 0; k--) *p++ = (uint8_t)(term >> (8 * k));
  for (uint32_t k = 0; k < len; k++) *p++ = data[k];
  return (uint64_t)(p - out);
}
