/*
 * rgb_wal_host.cpp -- the host-only half of include/ra_gpu_wal.h: record layout and the WAL file walk
 * of recover_records/5.  No HIP here, so the file also builds on its own under the sanitizers
 * (tests/test_wal_framing.py::test_scan_under_sanitizers).
 */
#include <stdint.h>
#include <vector>
#include "../../include/ra_gpu_wal.h"

extern "C" uint64_t rgb_wal_layout(rgb_wal_record *records, uint32_t n, uint64_t base) {
  for (uint32_t i = 0; i < n; ++i) {
    records[i].out_offset = base;
    base += (uint64_t)records[i].hdr_len + 24u + records[i].data_len;   /* DataSize, src/ra_log_wal.erl:526 */
  }
  return base;
}

/* ---- recovery: the record walk of recover_records/5 (src/ra_log_wal.erl:877-984), host code ---- */
namespace {
inline uint64_t be(const unsigned char *p, int nbytes) {
  uint64_t v = 0;
  for (int k = 0; k < nbytes; ++k) v = (v << 8) | p[k];
  return v;
}
}  // namespace

extern "C" int rgb_wal_scan(const void *bytes, uint64_t n_bytes, rgb_wal_scanned *out, uint32_t cap,
                            uint32_t *n_out, uint64_t *consumed, uint32_t *end) {
  if (!bytes || !n_out || !consumed || !end) return RGB_E_INVAL;
  const bool count_only = out == nullptr;              /* out = NULL: count the records, store nothing */
  const unsigned char *b = (const unsigned char *)bytes;
  /* <<"RAWA", 1:8/unsigned>> (:34-36, :826-835) */
  if (n_bytes < 5 || b[0] != 'R' || b[1] != 'A' || b[2] != 'W' || b[3] != 'A' || b[4] != 1) return RGB_E_INVAL;
  std::vector<bool> named(1u << 22, false);             /* IdRefs introduced by a long header so far */
  uint64_t pos = 5;
  uint32_t n = 0;
  *end = RGB_WAL_END_DATA;
  for (;;) {
    const uint64_t left = n_bytes - pos;
    if (left < 3) break;
    const uint32_t h = (uint32_t)be(b + pos, 3);
    const uint32_t trunc = h >> 23, form = (h >> 22) & 1u, id_ref = h & 0x3FFFFFu;
    uint64_t fixed = pos + 3;                           /* -> Checksum */
    uint32_t uid_len = 0;
    uint64_t uid_off = 0;
    if (form == 0) {
      if (left < 5) break;
      uid_len = (uint32_t)be(b + pos + 3, 2);
      uid_off = pos + 5;
      fixed = uid_off + uid_len;
    }
    if (fixed + 8 > n_bytes) break;
    const uint32_t checksum = (uint32_t)be(b + fixed, 4), data_len = (uint32_t)be(b + fixed + 4, 4);
    /* first clause: an all-zero record ends a pre-allocated file (:877-883) */
    if (h == 0 && checksum == 0 && data_len == 0) { *end = RGB_WAL_END_ZEROS; break; }
    if (fixed + 24 > n_bytes || fixed + 24 + (uint64_t)data_len > n_bytes) break;
    if (count_only) {
      if (form == 0) named[id_ref] = true;
      n += 1;
      pos = fixed + 24 + data_len;
      continue;
    }
    if (n == cap) { *end = RGB_WAL_END_CAP; break; }
    rgb_wal_scanned &r = out[n++];
    r.index = be(b + fixed + 8, 8);
    r.term = be(b + fixed + 16, 8);
    r.data_offset = fixed + 24;
    r.data_len = data_len;
    r.checksum = checksum;
    r.uid_offset = uid_off;
    r.id_ref = id_ref;
    r.uid_len = (uint16_t)uid_len;
    r.trunc = (uint8_t)trunc;
    r.next_offset = fixed + 24 + data_len;
    if (form == 0) {
      named[id_ref] = true;
      r.flags = RGB_WAL_REC_FIRST | RGB_WAL_REC_VALIDATE;
    } else {
      r.flags = named[id_ref] ? RGB_WAL_REC_VALIDATE : RGB_WAL_REC_UNKNOWN;
    }
    pos = r.next_offset;
  }
  *n_out = n;
  *consumed = pos;
  return RGB_OK;
}

