/*
 * rgb_wal.hip -- Adler-32 of a batch of WAL entries (include/ra_gpu_wal.h; reference
 * src/ra_log_wal.erl:528-534, 861, 873, 1028).  One wavefront (or a quarter of one) per entry; bytes
 * stream through 16-byte lane loads, sums through v_dot4_u32_u8.  HBM-bound: every payload byte is read once.
 *
 * Adler-32 (RFC 1950 8.2) of bytes d_0..d_{n-1}:  A = 1 + sum d_i,  B = n + sum (n - i) d_i,
 * both mod 65521, checksum = B << 16 | A.  The weighted sum is additive over any partition of the
 * bytes, so each lane handles whole 16-byte aligned chunks with local sums
 *     a = sum d_j,  b = sum (16 - j) d_j        (j = 0..15 inside the chunk)
 * and a chunk that starts s bytes into the stream contributes  (W - s - 16) * a + b  where W is
 * the weight of the stream's first byte.  Bytes outside the payload are masked to zero, which
 * makes the entry's alignment irrelevant.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ra_gpu_wal.h"

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;
#define ADLER_MOD 65521u
#ifndef WAL_WAVES_PER_BLOCK
#define WAL_WAVES_PER_BLOCK 4
#endif
#define WAL_UNROLL 4            /* 16-byte loads in flight per lane: 4 KiB per wavefront iteration */
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 dot4(u32 a, u32 b, u32 c) { return __builtin_amdgcn_udot4(a, b, c, false); }

/* local sums of one 16-byte chunk (little-endian dwords, byte 0 = lowest address) */
__device__ __forceinline__ void chunk_sums(const uint4 v, u32 &a, u32 &b) {
  a = dot4(v.x, 0x01010101u, 0); a = dot4(v.y, 0x01010101u, a);
  a = dot4(v.z, 0x01010101u, a); a = dot4(v.w, 0x01010101u, a);
  b = dot4(v.x, 0x0D0E0F10u, 0); b = dot4(v.y, 0x090A0B0Cu, b);
  b = dot4(v.z, 0x05060708u, b); b = dot4(v.w, 0x01020304u, b);
}

/* 0xFF for every byte of the dword at chunk bytes [first, first+4) that lies inside [lo, hi) */
__device__ __forceinline__ u32 byte_mask(u32 first, u32 lo, u32 hi) {
  u32 m = 0;
#pragma unroll
  for (u32 b = 0; b < 4; ++b) if (first + b >= lo && first + b < hi) m |= 0xFFu << (8 * b);
  return m;
}
/* the same for a whole 16-byte chunk, from two 64-bit shifts per half instead of sixteen byte tests */
__device__ __forceinline__ u64 ones64(u32 nbytes) { return nbytes >= 8u ? ~0ull : ((1ull << (8u * nbytes)) - 1ull); }
__device__ __forceinline__ uint4 keep_bytes(const uint4 w, u32 lo, u32 hi) {
  const u32 lo_a = lo < 8u ? lo : 8u, hi_a = hi < 8u ? hi : 8u;
  const u32 lo_b = lo > 8u ? lo - 8u : 0u, hi_b = hi > 8u ? hi - 8u : 0u;
  const u64 ma = ones64(hi_a) & ~ones64(lo_a), mb = ones64(hi_b) & ~ones64(lo_b);
  return make_uint4(w.x & (u32)ma, w.y & (u32)(ma >> 32), w.z & (u32)mb, w.w & (u32)(mb >> 32));
}

/* sum over the GROUP lanes that share a record (xor butterflies stay inside aligned groups) */
template <int GROUP>
__device__ __forceinline__ u32 group_sum(u32 v) {
#pragma unroll
  for (int off = GROUP / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

/* GROUP lanes per entry: 64 (one wavefront per entry) for KiB-sized payloads, 16 (four entries per
 * wavefront) for small ones -- the host picks by the batch's mean payload size. */
template <int GROUP>
__global__ __launch_bounds__(WAL_WAVES_PER_BLOCK * 64) void rgb_wal_adler32_kernel(
    const rgb_wal_entry *__restrict__ entries, u32 n, const unsigned char *__restrict__ data,
    u32 *__restrict__ out) {
  constexpr u32 PER_BLOCK = WAL_WAVES_PER_BLOCK * 64 / GROUP;
  const u32 lane = threadIdx.x & (GROUP - 1);
  const u32 e = blockIdx.x * PER_BLOCK + threadIdx.x / GROUP;
  const bool live = e < n;
  rgb_wal_entry en;
  en.index = en.term = en.data_offset = 0; en.data_len = 0; en._pad = 0;
  if (live) en = entries[e];
  const u64 off = en.data_offset;
  const u32 len = en.data_len;
  const u32 lead = (u32)(off & 15ull);                 /* masked bytes in front of the payload */
  const v4u *base = reinterpret_cast<const v4u *>(data + (off - lead));
  const u32 span = lead + len;                         /* aligned stream: [0, span) */
  const u32 n_chunks = live ? (span + 15u) >> 4 : 0u;
  const u32 span_q = span % ADLER_MOD;
  /* weight of stream byte j is (len + lead) - j: the last payload byte weighs 1 */
  u32 a_acc = 0, b_acc = 0;                            /* per lane, folded before they can wrap */
  for (u32 c0 = 0; c0 < n_chunks; c0 += GROUP * WAL_UNROLL) {
    uint4 v[WAL_UNROLL];
#pragma unroll
    for (int k = 0; k < WAL_UNROLL; ++k) {
      const u32 c = c0 + (u32)k * GROUP + lane;
      v[k] = make_uint4(0, 0, 0, 0);
      if (c < n_chunks) { const v4u t = __builtin_nontemporal_load(base + c); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
    }
#pragma unroll
    for (int k = 0; k < WAL_UNROLL; ++k) {
      const u32 c = c0 + (u32)k * GROUP + lane;
      if (c >= n_chunks) continue;
      const u32 s = c << 4;
      uint4 w = v[k];
      /* mask the bytes before the payload (first chunk) and after it (last chunk) */
      if (s < lead || s + 16u > span) {
        const u32 lo = s < lead ? lead - s : 0u, hi = span - s < 16u ? span - s : 16u;
        w.x &= byte_mask(0, lo, hi); w.y &= byte_mask(4, lo, hi);
        w.z &= byte_mask(8, lo, hi); w.w &= byte_mask(12, lo, hi);
      }
      u32 a, b;
      chunk_sums(w, a, b);
      /* (W - s - 16) may be negative on the last chunk: work modulo 65521 */
      const u32 wq = (span_q + 2u * ADLER_MOD - (s % ADLER_MOD) - 16u) % ADLER_MOD;
      a_acc += a;                                      /* <= 4080 per chunk */
      b_acc += (wq * a + b) % ADLER_MOD;
      if (b_acc >= 0x7FFF0000u) b_acc %= ADLER_MOD;
      if (a_acc >= 0x7FFF0000u) a_acc %= ADLER_MOD;
    }
  }
  const u32 a_sum = group_sum<GROUP>(a_acc % ADLER_MOD);   /* GROUP * 65520 fits */
  const u32 b_sum = group_sum<GROUP>(b_acc % ADLER_MOD);
  if (live && lane == 0) {
    /* the 16 framed bytes <<Idx:64, Term:64>> in front are one more chunk whose byte k weighs
     * n - k = len + (16 - k): big-endian words, so the dot4 weight vectors run the other way */
    const u32 ih = (u32)(en.index >> 32), il = (u32)en.index, th = (u32)(en.term >> 32), tl = (u32)en.term;
    u32 pa = dot4(ih, 0x01010101u, 0); pa = dot4(il, 0x01010101u, pa);
    pa = dot4(th, 0x01010101u, pa); pa = dot4(tl, 0x01010101u, pa);
    u32 pb = dot4(ih, 0x100F0E0Du, 0); pb = dot4(il, 0x0C0B0A09u, pb);
    pb = dot4(th, 0x08070605u, pb); pb = dot4(tl, 0x04030201u, pb);
    const u32 len_q = len % ADLER_MOD;
    const u32 A = (1u + pa + a_sum) % ADLER_MOD;
    const u32 B = ((16u + len_q) + (len_q * pa + pb) % ADLER_MOD + b_sum) % ADLER_MOD;
    out[e] = (B << 16) | A;
  }
}


/* ---- record framing: checksum + header + payload copy in one pass (src/ra_log_wal.erl:513-537) ----
 *
 * A record is  HeaderData ++ <<Checksum:32, EntryDataLen:32, Idx:64, Term:64>> ++ Payload  at out + out_offset.
 * The payload is READ in 16-byte chunks aligned to the SOURCE (non-temporal, one aligned request per lane; the
 * checksum is taken on these chunks exactly as the checksum kernel does) and WRITTEN in 16-byte chunks aligned to
 * the DESTINATION: destination chunk c is the byte-wise funnel of source chunks c - qd - 1 and c - qd (the shift
 * is constant over the record), and the older of the two arrives from the neighbouring lane through DPP
 * (row_ror / wave_ror), so no byte is loaded twice and no load is misaligned -- a 16-byte load that straddles two
 * aligned granules cost the first version of this kernel 30 % of its rate (588 us vs 401 us per GiB of 4 KiB
 * payloads, same kernel, source and destination in phase).  Only the destination chunks that hold the payload's
 * first and last byte are partial (the rest of those 16 bytes is the prefix, or the neighbouring record, written by
 * another lane group in no particular order): they go as at most four aligned power-of-two stores each.  The 24
 * fixed bytes are two unaligned vector stores by the group's first lane once the checksum is known; HeaderData is
 * copied byte per lane (3 bytes for a known writer). */

/* bytes [lo, hi) of a 16-byte chunk held in registers to its 16-byte aligned place `base`: aligned power-of-two
 * stores (1, 2, 4, 8 bytes going up to the first 8-byte boundary that fits, then 8, 4, 2, 1 coming down) */
struct halves16 {
  u64 lo, hi;
  /* the 8 bytes from chunk byte q on (q and the store's size never straddle the halves).  By value: a lambda that
   * captured the halves by reference made the compiler select between their ADDRESSES -- a scratch array, a pointer
   * table in LDS and a flat load in front of every store */
  __device__ __forceinline__ u64 operator()(u32 q) const { const u64 x = (q & 8u) ? hi : lo; return x >> (8u * (q & 7u)); }
};
__device__ __forceinline__ halves16 halves_of(const uint4 w) {
  return halves16{(u64)w.x | ((u64)w.y << 32), (u64)w.z | ((u64)w.w << 32)};
}
/* bytes [lo, 16): the chunk that holds the payload's first byte.  From the top down -- 8, 4, 2, 1 bytes by the bits of
 * 16 - lo -- as a shift chain over one 64-bit word: no extraction at a variable offset (which the compiler once
 * turned into a scratch store of the chunk and loads at computed offsets) */
__device__ __forceinline__ void store_head16(unsigned char *base, const uint4 w, u32 lo) {
  const halves16 h = halves_of(w);
  const u32 n = 16u - lo;                               /* 1..15 bytes */
  u64 cur = h.hi;
  u32 e = 16u;                                          /* the bytes still to store end here */
  if (n & 8u) { *reinterpret_cast<u64 *>(base + 8) = h.hi; cur = h.lo; e = 8u; }
  if (n & 4u) { *reinterpret_cast<u32 *>(base + e - 4u) = (u32)(cur >> 32); cur <<= 32; e -= 4u; }
  if (n & 2u) { *reinterpret_cast<unsigned short *>(base + e - 2u) = (unsigned short)(cur >> 48); cur <<= 16; e -= 2u; }
  if (n & 1u) { base[e - 1u] = (unsigned char)(cur >> 56); }
}
/* bytes [0, hi): the chunk that holds the payload's last byte.  From the bottom up, the same chain mirrored */
__device__ __forceinline__ void store_tail16(unsigned char *base, const uint4 w, u32 hi) {
  const halves16 h = halves_of(w);
  u64 cur = h.lo;
  u32 p = 0;
  if (hi & 8u) { *reinterpret_cast<u64 *>(base) = h.lo; cur = h.hi; p = 8u; }
  if (hi & 4u) { *reinterpret_cast<u32 *>(base + p) = (u32)cur; cur >>= 32; p += 4u; }
  if (hi & 2u) { *reinterpret_cast<unsigned short *>(base + p) = (unsigned short)cur; cur >>= 16; p += 2u; }
  if (hi & 1u) { base[p] = (unsigned char)cur; }
}
/* bytes [lo, hi), both inside the chunk (a payload shorter than its chunk) */
__device__ __forceinline__ void store_sub16(unsigned char *base, const uint4 w, u32 lo, u32 hi) {
  const halves16 sub = halves_of(w);
  u32 p = lo;
  if ((p & 1u) && p + 1u <= hi) { base[p] = (unsigned char)sub(p); p += 1u; }
  if ((p & 2u) && p + 2u <= hi) { *reinterpret_cast<unsigned short *>(base + p) = (unsigned short)sub(p); p += 2u; }
  if ((p & 4u) && p + 4u <= hi) { *reinterpret_cast<u32 *>(base + p) = (u32)sub(p); p += 4u; }
  if ((p & 8u) && p + 8u <= hi) { *reinterpret_cast<u64 *>(base + p) = sub(p); p += 8u; }
  if (p + 8u <= hi) { *reinterpret_cast<u64 *>(base + p) = sub(p); p += 8u; }
  if (p + 4u <= hi) { *reinterpret_cast<u32 *>(base + p) = (u32)sub(p); p += 4u; }
  if (p + 2u <= hi) { *reinterpret_cast<unsigned short *>(base + p) = (unsigned short)sub(p); p += 2u; }
  if (p + 1u <= hi) { base[p] = (unsigned char)sub(p); }
}

/* whole aligned chunk, written once and not read again by the device.  STREAM (one record per wavefront, KiB-sized
 * payloads): non-temporal.  Small records: a plain store -- every other 128-byte line of the output holds a record
 * boundary whose bytes (prefix, partial chunks) arrive from other instructions, and a line the non-temporal store
 * pushed out early is written to memory twice (256-byte payloads, same box: 396 -> 360 us per 2 M records; 4 KiB
 * payloads the other way round, 418 -> 441 us) */
template <bool STREAM>
__device__ __forceinline__ void store16_stream(unsigned char *p, const uint4 w) {
  v4u t; t.x = w.x; t.y = w.y; t.z = w.z; t.w = w.w;
#if defined(RGB_HOST_EMULATION)
  *reinterpret_cast<v4u *>(p) = t;
#else
  if (STREAM) __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p));
  else *reinterpret_cast<v4u *>(p) = t;
#endif
}

/* lane L of a GROUP-lane group receives the value of lane (L - 1) mod GROUP of the same group */
template <int GROUP>
__device__ __forceinline__ u32 rot1(u32 v) {
#ifdef RGB_HOST_EMULATION
  const int t = (int)threadIdx.x;                       /* the emulation's __shfl takes the lane's number in the block */
  return __shfl(v, (t & ~(GROUP - 1)) | ((t - 1) & (GROUP - 1)), 64);
#elif defined(WAL_X_NODPP)
  const int t = (int)(threadIdx.x & 63u);
  return __shfl(v, (t & ~(GROUP - 1)) | ((t - 1) & (GROUP - 1)), 64);
#else
  static_assert(GROUP == 8 || GROUP == 16 || GROUP == 64, "a DPP row is 16 lanes, a wavefront 64");
  /* 8-lane groups: right for every lane but the group's first (rot_last serves that one) */
  if (GROUP <= 16) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
#endif
}
template <int GROUP>
__device__ __forceinline__ uint4 rot1(const uint4 v) {
  return make_uint4(rot1<GROUP>(v.x), rot1<GROUP>(v.y), rot1<GROUP>(v.z), rot1<GROUP>(v.w));
}
/* the group's FIRST lane receives the value of the group's last lane (the other lanes: unspecified) */
template <int GROUP>
__device__ __forceinline__ u32 rot_last(u32 v) {
#if defined(RGB_HOST_EMULATION) || defined(WAL_X_NODPP)
  return rot1<GROUP>(v);
#else
  if (GROUP == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x129 /* row_ror:9 */, 0xF, 0xF, false);
  return rot1<GROUP>(v);
#endif
}
template <int GROUP>
__device__ __forceinline__ uint4 rot_last(const uint4 v) {
  if (GROUP != 8) return rot1<GROUP>(v);
  return make_uint4(rot_last<GROUP>(v.x), rot_last<GROUP>(v.y), rot_last<GROUP>(v.z), rot_last<GROUP>(v.w));
}
/* ({hi, lo} >> 8 sb) & 0xFFFFFFFF, sb = 0..3: v_alignbyte_b32 */
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 sb) {
#ifdef RGB_HOST_EMULATION
  return (u32)((((u64)hi << 32) | lo) >> (8u * (sb & 3u)));
#else
  return __builtin_amdgcn_alignbyte(hi, lo, sb);
#endif
}

/* bytes [o, o + 16) of the 32 bytes p ++ c, o = 0..16 */
template <bool UNIFORM>
__device__ __forceinline__ uint4 window16(const uint4 p, const uint4 c, u32 o) {
  const u32 q = o >> 2, sb = o & 3u;
  if (UNIFORM) {
    /* o is the same in every lane of the wavefront: the dword offset selects straight-line code */
#ifndef RGB_HOST_EMULATION
    const u32 qs = (u32)__builtin_amdgcn_readfirstlane((int)q);
#else
    const u32 qs = q;
#endif
    switch (qs) {
      case 0: return make_uint4(alignbyte(p.y, p.x, sb), alignbyte(p.z, p.y, sb), alignbyte(p.w, p.z, sb), alignbyte(c.x, p.w, sb));
      case 1: return make_uint4(alignbyte(p.z, p.y, sb), alignbyte(p.w, p.z, sb), alignbyte(c.x, p.w, sb), alignbyte(c.y, c.x, sb));
      case 2: return make_uint4(alignbyte(p.w, p.z, sb), alignbyte(c.x, p.w, sb), alignbyte(c.y, c.x, sb), alignbyte(c.z, c.y, sb));
      case 3: return make_uint4(alignbyte(c.x, p.w, sb), alignbyte(c.y, c.x, sb), alignbyte(c.z, c.y, sb), alignbyte(c.w, c.z, sb));
      default: return c;                                  /* o = 16 */
    }
  }
  /* per-lane offset: byte shift of every neighbouring dword pair, then a three-stage dword selector */
  const u32 A0 = alignbyte(p.y, p.x, sb), A1 = alignbyte(p.z, p.y, sb), A2 = alignbyte(p.w, p.z, sb),
            A3 = alignbyte(c.x, p.w, sb), A4 = alignbyte(c.y, c.x, sb), A5 = alignbyte(c.z, c.y, sb),
            A6 = alignbyte(c.w, c.z, sb);
  const bool q0 = (q & 1u) != 0, q1 = (q & 2u) != 0, q2 = (q & 4u) != 0;
  const u32 B0 = q0 ? A1 : A0, B1 = q0 ? A2 : A1, B2 = q0 ? A3 : A2, B3 = q0 ? A4 : A3, B4 = q0 ? A5 : A4, B5 = q0 ? A6 : A5;
  uint4 e;
  e.x = q2 ? c.x : q1 ? B2 : B0; e.y = q2 ? c.y : q1 ? B3 : B1;
  e.z = q2 ? c.z : q1 ? B4 : B2; e.w = q2 ? c.w : q1 ? B5 : B3;
  return e;
}

/* destination chunk at stream position dpos (a multiple of 16) of the record whose payload sits at stream positions
 * [ps, pe) = f: a whole chunk is streamed, the chunks that hold the payload's first / last byte go as aligned
 * power-of-two stores.  Everything by value (see halves16). */
template <bool STREAM>
__device__ __forceinline__ void emit_chunk(unsigned char *rec_al, u32 ps, u32 pe, u32 dpos, const uint4 f) {
  if (dpos + 16u > ps && dpos < pe) {                   /* the chunk holds payload bytes */
    unsigned char *to = rec_al + dpos;
    const bool head = dpos < ps, tail = dpos + 16u > pe;
    if (!head && !tail) store16_stream<STREAM>(to, f);
    else if (!head) store_tail16(to, f, pe - dpos);
    else if (!tail) store_head16(to, f, ps - dpos);
    else store_sub16(to, f, ps - dpos, pe - dpos);
  }
}

/* mean payload up to which a batch is framed with eight lanes per record (then sixteen up to 1 KiB, then a wavefront) */
#ifndef WAL_FRAME_EIGHT_MAX
#define WAL_FRAME_EIGHT_MAX 320u
#endif
template <int GROUP>
__global__ __launch_bounds__(WAL_WAVES_PER_BLOCK * 64) void rgb_wal_frame_kernel(
    const rgb_wal_record *__restrict__ recs, u32 n, const unsigned char *__restrict__ data,
    unsigned char *__restrict__ out, u32 *__restrict__ sums_out, u32 flags) {
  constexpr u32 PER_BLOCK = WAL_WAVES_PER_BLOCK * 64 / GROUP;
  constexpr bool UNI = GROUP == 64;                     /* one record per wavefront: its shifts are wave-uniform */
  constexpr int UNROLL = GROUP == 64 ? WAL_UNROLL : 2;  /* small records: 512 / 256 bytes per round and group */
  const u32 lane = threadIdx.x & (GROUP - 1);
  u32 e = blockIdx.x * PER_BLOCK + threadIdx.x / GROUP;
#ifndef RGB_HOST_EMULATION
  /* one record per wavefront: say so, and the descriptor and everything derived from it live in scalar registers */
  if (UNI) e = (u32)__builtin_amdgcn_readfirstlane((int)e);
#endif
  const bool live = e < n;
  rgb_wal_record r;
  r.index = r.term = r.data_offset = r.hdr_offset = r.out_offset = 0; r.data_len = r.hdr_len = 0;
  if (live) r = recs[e];
  const u32 len = r.data_len;
  const u32 prefix = r.hdr_len + 24u;                   /* HeaderData + Checksum, Len, Idx, Term */
  unsigned char *rec = out + r.out_offset;
  const u32 lr = (u32)((uintptr_t)rec & 15u);           /* record byte b sits at destination stream position lr + b */
  unsigned char *rec_al = rec - lr;                     /* destination chunk c is rec_al[16 c .. 16 c + 16) */
  const unsigned char *pay = data + r.data_offset;
  const u32 ls = (u32)((uintptr_t)pay & 15u);           /* payload byte p sits at source stream position ls + p */
  const v4u *src = reinterpret_cast<const v4u *>(pay - ls);
  const u32 span = ls + len;
  const u32 n_src = (live && len) ? (span + 15u) >> 4 : 0u;
  const u32 span_q = span % ADLER_MOD;
  const u32 ps = lr + prefix, pe = ps + len;            /* the payload at destination positions [ps, pe) */
  const u32 delta = ps - ls;                            /* destination position = source position + delta (> 0) */
  const u32 qd = delta >> 4, dm = delta & 15u;
  /* funnel i = (source chunks i - 1, i) is destination chunk i + qd; one more than the source chunks when the last
   * source chunk's tail spills into a further destination chunk */
  const u32 n_fun = n_src ? n_src + (((n_src + qd) << 4) < pe ? 1u : 0u) : 0u;
  /* HeaderData: one byte per lane, requested now so that it arrives under the payload loads */
  const unsigned char *hdr = data + r.hdr_offset;
  u32 hbyte = 0;
  if (live && lane < r.hdr_len) hbyte = hdr[lane];
  u32 a_acc = 0, b_acc = 0;
  uint4 carry = make_uint4(0, 0, 0, 0);                 /* lane 0: the group's last lane's chunk of the previous round */
  /* weight of the byte behind source chunk i, (span - 16 i - 16) mod 65521, kept per lane and stepped down by
   * 16 GROUP per chunk instead of two divisions per chunk */
  constexpr u32 STEP = (16u * GROUP) % ADLER_MOD;
  u32 wq = (span_q + 2u * ADLER_MOD - ((lane << 4) % ADLER_MOD) - 16u) % ADLER_MOD;
  for (u32 c0 = 0; c0 < n_fun; c0 += GROUP * UNROLL) {
    uint4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const u32 i = c0 + (u32)k * GROUP + lane;
      v[k] = make_uint4(0, 0, 0, 0);
#ifdef WAL_X_NOREAD           /* EXPERIMENT (breaks the output): the write side alone */
      if (i < n_src) v[k] = make_uint4(i, lane, e, len);
#else
      if (i < n_src) { const v4u t = __builtin_nontemporal_load(src + i); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
#endif
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      if (c0 + (u32)k * GROUP >= n_fun) break;          /* the same for the whole group: a short last round is cheap */
      const u32 i = c0 + (u32)k * GROUP + lane;
      const u32 s = i << 4;
      if (i < n_src) {
        uint4 w = v[k];
        if (s < ls || s + 16u > span) {                 /* bytes in front of the payload / behind it: zero */
          w = keep_bytes(w, s < ls ? ls - s : 0u, span - s < 16u ? span - s : 16u);
          v[k] = w;
        }
        u32 a, b;
        chunk_sums(w, a, b);
        a_acc += a;
        b_acc += (wq * a + b) % ADLER_MOD;
        if (b_acc >= 0x7FFF0000u) b_acc %= ADLER_MOD;
        if (a_acc >= 0x7FFF0000u) a_acc %= ADLER_MOD;
      }
      wq = wq >= STEP ? wq - STEP : wq + (ADLER_MOD - STEP);
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      if (c0 + (u32)k * GROUP >= n_fun) break;
      const u32 i = c0 + (u32)k * GROUP + lane;
      /* every lane of the group takes part in the exchange, whatever it loaded */
      const uint4 rot = rot1<GROUP>(v[k]);
      const uint4 prev = lane == 0u ? carry : rot;
      carry = rot_last<GROUP>(v[k]);
      const uint4 f = window16<UNI>(prev, v[k], 16u - dm);
#ifdef WAL_X_NOWRITE          /* EXPERIMENT (breaks the output): the read + compute side alone */
      if (i < n_fun && f.x == 0x12345678u && f.y == 0x9ABCDEF0u) emit_chunk<UNI>(rec_al, ps, pe, (i + qd) << 4, f);
#else
      if (i < n_fun) emit_chunk<UNI>(rec_al, ps, pe, (i + qd) << 4, f);
#endif
    }
  }
  const u32 a_sum = group_sum<GROUP>(a_acc % ADLER_MOD);
  const u32 b_sum = group_sum<GROUP>(b_acc % ADLER_MOD);
  if (!live) return;
  const u32 ih = (u32)(r.index >> 32), il = (u32)r.index, th = (u32)(r.term >> 32), tl = (u32)r.term;
  u32 pa = dot4(ih, 0x01010101u, 0); pa = dot4(il, 0x01010101u, pa);
  pa = dot4(th, 0x01010101u, pa); pa = dot4(tl, 0x01010101u, pa);
  u32 pb = dot4(ih, 0x100F0E0Du, 0); pb = dot4(il, 0x0C0B0A09u, pb);
  pb = dot4(th, 0x08070605u, pb); pb = dot4(tl, 0x04030201u, pb);
  const u32 len_q = len % ADLER_MOD;
  const u32 A = (1u + pa + a_sum) % ADLER_MOD;
  const u32 B = ((16u + len_q) + (len_q * pa + pb) % ADLER_MOD + b_sum) % ADLER_MOD;
  const u32 cs = (flags & RGB_WAL_NO_CHECKSUMS) ? 0u : ((B << 16) | A);
  /* HeaderData verbatim (a known writer's is 3 bytes, a new writer's carries its uid) */
  if (lane < r.hdr_len) rec[lane] = (unsigned char)hbyte;
  for (u32 j = GROUP + lane; j < r.hdr_len; j += GROUP) rec[j] = hdr[j];
  if (lane == 0u) {
    if (sums_out) sums_out[e] = cs;
    /* <<Checksum:32, EntryDataLen:32, Idx:64, Term:64>> big endian: 16 + 8 bytes at any alignment (gfx950 global
     * stores take any alignment) */
    struct __attribute__((packed)) fixed24 { v4u a; u64 b; };
    fixed24 fx;
    fx.a.x = __builtin_bswap32(cs); fx.a.y = __builtin_bswap32(len);
    fx.a.z = __builtin_bswap32(ih); fx.a.w = __builtin_bswap32(il);
    fx.b = (u64)__builtin_bswap32(th) | ((u64)__builtin_bswap32(tl) << 32);
    __builtin_memcpy(rec + r.hdr_len, &fx, 24);
  }
}

/* ---- small records, the DIRECT form (WAL_X_DIRECT): no funnel ----
 * Eight lanes per record; lane j handles payload bytes [16 j, 16 j + 16) as they lie: one 16-byte load at the
 * payload's own alignment, the checksum on exactly those bytes (the piece that starts p bytes into the payload weighs
 * (len - p - 16) a + b, no masks), one 16-byte store at the destination's own alignment -- gfx950 global loads and
 * stores take any byte alignment.  A payload that does not end on a piece boundary ends with the piece
 * [len - 16, len), which overlaps its predecessor: the overlapped bytes are masked out of the sums (what is left
 * weighs exactly b) and stored twice with the same value.  Payloads under 16 bytes go byte by byte.  Against the
 * funnel form this trades requests that straddle a 64-byte boundary (one lane in four, both directions) for ~3/4 of
 * the vector instructions: no rotation, no window, no partial-store chains. */
typedef v4u v4u_any __attribute__((aligned(1)));
#ifndef WAL_X_DIRECT
#define WAL_X_DIRECT 1      /* lane groups that frame in the direct form: 1 = eight lanes per record (mean payload <= 320 B:
                               0.55 against 0.42-0.44 of the roofline on 256-byte payloads, same box), 2 = sixteen too,
                               3 = every size; 0 = the funnel form everywhere */
#endif
template <int GROUP>
__global__ __launch_bounds__(WAL_WAVES_PER_BLOCK * 64) void rgb_wal_frame_direct_kernel(
    const rgb_wal_record *__restrict__ recs, u32 n, const unsigned char *__restrict__ data,
    unsigned char *__restrict__ out, u32 *__restrict__ sums_out, u32 flags) {
  constexpr u32 PER_BLOCK = WAL_WAVES_PER_BLOCK * 64 / GROUP;
  constexpr bool UNI = GROUP == 64;                     /* one record per wavefront: descriptor in scalar registers */
  constexpr int UNROLL = GROUP == 64 ? WAL_UNROLL : 2;
  const u32 lane = threadIdx.x & (GROUP - 1);
  u32 e = blockIdx.x * PER_BLOCK + threadIdx.x / GROUP;
#ifndef RGB_HOST_EMULATION
  if (UNI) e = (u32)__builtin_amdgcn_readfirstlane((int)e);
#endif
  const bool live = e < n;
  rgb_wal_record r;
  r.index = r.term = r.data_offset = r.hdr_offset = r.out_offset = 0; r.data_len = r.hdr_len = 0;
  if (live) r = recs[e];
  const u32 len = r.data_len;
  unsigned char *rec = out + r.out_offset;
  unsigned char *dst = rec + r.hdr_len + 24u;           /* the payload's place in the record */
  const unsigned char *pay = data + r.data_offset;
  const unsigned char *hdr = data + r.hdr_offset;
  u32 hbyte = 0;
  if (live && lane < r.hdr_len) hbyte = hdr[lane];
  const u32 n_full = live ? len >> 4 : 0u, rem = live ? len & 15u : 0u;
  const u32 len_q = len % ADLER_MOD;
  u32 a_acc = 0, b_acc = 0;
  constexpr u32 STEP = (16u * GROUP) % ADLER_MOD;
  /* weight of the byte behind piece j: (len - 16 j - 16) mod 65521, stepped down per round */
  u32 wq = (len_q + 2u * ADLER_MOD - ((lane << 4) % ADLER_MOD) - 16u) % ADLER_MOD;
  for (u32 j0 = 0; j0 < n_full; j0 += GROUP * UNROLL) {
    uint4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const u32 j = j0 + (u32)k * GROUP + lane;
      v[k] = make_uint4(0, 0, 0, 0);
      if (j < n_full) {
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u_any *>(pay + ((size_t)j << 4)));
        v[k] = make_uint4(t.x, t.y, t.z, t.w);
      }
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const u32 j = j0 + (u32)k * GROUP + lane;
      if (j < n_full) {
        v4u t; t.x = v[k].x; t.y = v[k].y; t.z = v[k].z; t.w = v[k].w;
#if defined(RGB_HOST_EMULATION)
        *reinterpret_cast<v4u_any *>(dst + ((size_t)j << 4)) = t;
#else
        /* streamed past the caches for a record per wavefront, plain for the small-record groups (see store16_stream) */
        if (UNI) __builtin_nontemporal_store(t, reinterpret_cast<v4u_any *>(dst + ((size_t)j << 4)));
        else *reinterpret_cast<v4u_any *>(dst + ((size_t)j << 4)) = t;
#endif
        u32 a, b;
        chunk_sums(v[k], a, b);
        a_acc += a;
        b_acc += (wq * a + b) % ADLER_MOD;
        if (b_acc >= 0x7FFF0000u) b_acc %= ADLER_MOD;
        if (a_acc >= 0x7FFF0000u) a_acc %= ADLER_MOD;
      }
      wq = wq >= STEP ? wq - STEP : wq + (ADLER_MOD - STEP);
    }
  }
  if (rem && lane == (n_full & (GROUP - 1))) {           /* the last len mod 16 bytes: they weigh rem .. 1 */
    uint4 w = make_uint4(0, 0, 0, 0);
    if (n_full) {
      const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u_any *>(pay + (len - 16u)));
      *reinterpret_cast<v4u_any *>(dst + (len - 16u)) = t;
      w = keep_bytes(make_uint4(t.x, t.y, t.z, t.w), 16u - rem, 16u);
    } else {
      u64 lo = 0, hi = 0;                                /* bytes at chunk positions [16 - rem, 16) */
      for (u32 k = 0; k < rem; ++k) {
        const u32 c = pay[k], q = 16u - rem + k;
        dst[k] = (unsigned char)c;
        if (q < 8u) lo |= (u64)c << (8u * q); else hi |= (u64)c << (8u * (q - 8u));
      }
      w = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    }
    u32 a, b;
    chunk_sums(w, a, b);
    a_acc += a; b_acc += b;                              /* both far from wrapping: one piece */
  }
  const u32 a_sum = group_sum<GROUP>(a_acc % ADLER_MOD);
  const u32 b_sum = group_sum<GROUP>(b_acc % ADLER_MOD);
  if (!live) return;
  const u32 ih = (u32)(r.index >> 32), il = (u32)r.index, th = (u32)(r.term >> 32), tl = (u32)r.term;
  u32 pa = dot4(ih, 0x01010101u, 0); pa = dot4(il, 0x01010101u, pa);
  pa = dot4(th, 0x01010101u, pa); pa = dot4(tl, 0x01010101u, pa);
  u32 pb = dot4(ih, 0x100F0E0Du, 0); pb = dot4(il, 0x0C0B0A09u, pb);
  pb = dot4(th, 0x08070605u, pb); pb = dot4(tl, 0x04030201u, pb);
  const u32 A = (1u + pa + a_sum) % ADLER_MOD;
  const u32 B = ((16u + len_q) + (len_q * pa + pb) % ADLER_MOD + b_sum) % ADLER_MOD;
  const u32 cs = (flags & RGB_WAL_NO_CHECKSUMS) ? 0u : ((B << 16) | A);
  if (lane < r.hdr_len) rec[lane] = (unsigned char)hbyte;
  for (u32 j = GROUP + lane; j < r.hdr_len; j += GROUP) rec[j] = hdr[j];
  if (lane == 0u) {
    if (sums_out) sums_out[e] = cs;
    struct __attribute__((packed)) fixed24 { v4u a; u64 b; };
    fixed24 fx;
    fx.a.x = __builtin_bswap32(cs); fx.a.y = __builtin_bswap32(len);
    fx.a.z = __builtin_bswap32(ih); fx.a.w = __builtin_bswap32(il);
    fx.b = (u64)__builtin_bswap32(th) | ((u64)__builtin_bswap32(tl) << 32);
    __builtin_memcpy(rec + r.hdr_len, &fx, 24);
  }
}

}  // namespace

/* the context only supplies the default stream; rgb_api.hip exports the accessor */
extern "C" void *rgb_ctx_stream(rgb_ctx *ctx);
extern "C" int rgb_ctx_device(rgb_ctx *ctx);          /* the HIP device the context (and its stream) lives on */

/* off + len <= bytes without the u64 wrap-around of the sum (descriptors come from the caller) */
static inline bool wal_slice_ok(uint64_t off, uint64_t len, uint64_t bytes) { return off <= bytes && len <= bytes - off; }

extern "C" int rgb_wal_adler32_device(rgb_ctx *ctx, const void *d_entries, uint32_t n, const void *d_data,
                                      uint64_t data_bytes, void *d_checksums, void *stream) {
  if (!ctx || (n && (!d_entries || !d_checksums))) return RGB_E_INVAL;
  if (n == 0) return RGB_OK;
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)rgb_ctx_stream(ctx);
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  /* lanes per entry by the batch's mean payload: four entries per wavefront below 1 KiB */
  if (data_bytes / n < 1024u) {
    const u32 per = WAL_WAVES_PER_BLOCK * 64 / 16;
    hipLaunchKernelGGL(rgb_wal_adler32_kernel<16>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                       (const rgb_wal_entry *)d_entries, n, (const unsigned char *)d_data, (u32 *)d_checksums);
  } else {
    const u32 per = WAL_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(rgb_wal_adler32_kernel<64>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                       (const rgb_wal_entry *)d_entries, n, (const unsigned char *)d_data, (u32 *)d_checksums);
  }
  return hipGetLastError() == hipSuccess ? RGB_OK : RGB_E_HIP;
}

extern "C" int rgb_wal_frame_device(rgb_ctx *ctx, const void *d_records, uint32_t n, const void *d_data,
                                    uint64_t data_bytes, void *d_out, uint64_t out_bytes, void *d_checksums,
                                    uint32_t flags, void *stream) {
  if (!ctx || (n && (!d_records || !d_out)) || (flags & ~RGB_WAL_NO_CHECKSUMS)) return RGB_E_INVAL;
  if (n == 0) return RGB_OK;
  if (out_bytes < 27ull * n) return RGB_E_INVAL;       /* the shortest record is 3 + 24 bytes */
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)rgb_ctx_stream(ctx);
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (data_bytes / n <= WAL_FRAME_EIGHT_MAX) {
    /* the smallest payloads: eight lanes per record, eight records per wavefront (the per-record work that does not
     * shrink with the payload -- descriptor, reduction, prefix -- is paid per WAVEFRONT instruction) */
    const u32 per = WAL_WAVES_PER_BLOCK * 64 / 8;
    if (WAL_X_DIRECT >= 1)
      hipLaunchKernelGGL(rgb_wal_frame_direct_kernel<8>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
    else
      hipLaunchKernelGGL(rgb_wal_frame_kernel<8>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
  } else if (data_bytes / n < 1024u) {
    const u32 per = WAL_WAVES_PER_BLOCK * 64 / 16;
    if (WAL_X_DIRECT >= 2)
      hipLaunchKernelGGL(rgb_wal_frame_direct_kernel<16>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
    else
      hipLaunchKernelGGL(rgb_wal_frame_kernel<16>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
  } else {
    const u32 per = WAL_WAVES_PER_BLOCK;
    if (WAL_X_DIRECT >= 3)
      hipLaunchKernelGGL(rgb_wal_frame_direct_kernel<64>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
    else
      hipLaunchKernelGGL(rgb_wal_frame_kernel<64>, dim3((n + per - 1) / per), dim3(WAL_WAVES_PER_BLOCK * 64), 0, st,
                         (const rgb_wal_record *)d_records, n, (const unsigned char *)d_data,
                         (unsigned char *)d_out, (u32 *)d_checksums, flags);
  }
  return hipGetLastError() == hipSuccess ? RGB_OK : RGB_E_HIP;
}

/* ---- host-buffer form: staging buffers live beside the context (keyed by it), grown on demand ---- */
#include <mutex>
#include <unordered_map>
#include <vector>
namespace {
struct wal_stage {
  void *d_entries = nullptr; void *d_data = nullptr; void *d_out = nullptr; size_t cap_e = 0, cap_d = 0;
  void *d_records = nullptr; void *d_frame = nullptr; size_t cap_r = 0, cap_f = 0;   /* rgb_wal_frame */
};
std::mutex g_stage_mu;
std::unordered_map<rgb_ctx *, wal_stage> g_stage;
int grow(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = need + need / 2 + 4096;
  if (hipMalloc(p, want) != hipSuccess) return -1;
  *cap = want;
  return 0;
}
}  // namespace

extern "C" void rgb_wal_release(rgb_ctx *ctx) {      /* called by rgb_close */
  std::lock_guard<std::mutex> lk(g_stage_mu);
  auto it = g_stage.find(ctx);
  if (it == g_stage.end()) return;
  if (it->second.d_entries) (void)hipFree(it->second.d_entries);
  if (it->second.d_data) (void)hipFree(it->second.d_data);
  if (it->second.d_out) (void)hipFree(it->second.d_out);
  if (it->second.d_records) (void)hipFree(it->second.d_records);
  if (it->second.d_frame) (void)hipFree(it->second.d_frame);
  g_stage.erase(it);
}

extern "C" int rgb_wal_adler32(rgb_ctx *ctx, const rgb_wal_entry *entries, uint32_t n, const void *data,
                               uint64_t data_bytes, uint32_t *checksums) {
  if (!ctx || (n && (!entries || !checksums)) || (data_bytes && !data)) return RGB_E_INVAL;
  if (n == 0) return RGB_OK;
  for (uint32_t i = 0; i < n; ++i)
    if (!wal_slice_ok(entries[i].data_offset, entries[i].data_len, data_bytes)) return RGB_E_INVAL;
  /* a dirty-scheduler thread or a multi-context process: allocations must land on the context's device */
  if (hipSetDevice(rgb_ctx_device(ctx)) != hipSuccess) return RGB_E_HIP;
  std::lock_guard<std::mutex> lk(g_stage_mu);
  wal_stage &s = g_stage[ctx];
  size_t cap_o = s.cap_e / sizeof(rgb_wal_entry) * sizeof(u32);
  const size_t need_e = (size_t)n * sizeof(rgb_wal_entry);
  if (need_e > s.cap_e) {
    if (s.d_out) { (void)hipFree(s.d_out); s.d_out = nullptr; }
    if (grow(&s.d_entries, &s.cap_e, need_e)) return RGB_E_NOMEM;
    cap_o = s.cap_e / sizeof(rgb_wal_entry) * sizeof(u32);
    if (hipMalloc(&s.d_out, cap_o) != hipSuccess) return RGB_E_NOMEM;
  }
  if (grow(&s.d_data, &s.cap_d, (size_t)data_bytes + 16)) return RGB_E_NOMEM;
  hipStream_t st = (hipStream_t)rgb_ctx_stream(ctx);
  if (hipMemcpyAsync(s.d_entries, entries, need_e, hipMemcpyHostToDevice, st) != hipSuccess) return RGB_E_HIP;
  if (data_bytes && hipMemcpyAsync(s.d_data, data, data_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return RGB_E_HIP;
  int rc = rgb_wal_adler32_device(ctx, s.d_entries, n, s.d_data, data_bytes, s.d_out, st);
  if (rc) return rc;
  if (hipMemcpyAsync(checksums, s.d_out, (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, st) != hipSuccess) return RGB_E_HIP;
  return hipStreamSynchronize(st) == hipSuccess ? RGB_OK : RGB_E_HIP;
}

extern "C" int rgb_wal_frame(rgb_ctx *ctx, const rgb_wal_record *records, uint32_t n, const void *data,
                             uint64_t data_bytes, void *out, uint64_t out_bytes, uint32_t flags) {
  if (!ctx || (n && (!records || !out)) || (data_bytes && !data)) return RGB_E_INVAL;
  if (n == 0) return RGB_OK;
  /* every slice inside its buffer, records ascending and disjoint in the output */
  uint64_t floor_off = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const rgb_wal_record &r = records[i];
    if (!wal_slice_ok(r.data_offset, r.data_len, data_bytes) || !wal_slice_ok(r.hdr_offset, r.hdr_len, data_bytes))
      return RGB_E_INVAL;
    if (r.hdr_len < 3u || r.out_offset < floor_off) return RGB_E_INVAL;
    const uint64_t rec_len = (uint64_t)r.hdr_len + 24u + (uint64_t)r.data_len;   /* u32 + u32 + 24: no wrap */
    if (!wal_slice_ok(r.out_offset, rec_len, out_bytes)) return RGB_E_INVAL;
    floor_off = r.out_offset + rec_len;
  }
  if (hipSetDevice(rgb_ctx_device(ctx)) != hipSuccess) return RGB_E_HIP;
  std::lock_guard<std::mutex> lk(g_stage_mu);
  wal_stage &s = g_stage[ctx];
  const size_t need_r = (size_t)n * sizeof(rgb_wal_record);
  if (grow(&s.d_records, &s.cap_r, need_r)) return RGB_E_NOMEM;
  if (grow(&s.d_data, &s.cap_d, (size_t)data_bytes + 16)) return RGB_E_NOMEM;
  if (grow(&s.d_frame, &s.cap_f, (size_t)out_bytes + 16)) return RGB_E_NOMEM;
  hipStream_t st = (hipStream_t)rgb_ctx_stream(ctx);
  if (hipMemcpyAsync(s.d_records, records, need_r, hipMemcpyHostToDevice, st) != hipSuccess) return RGB_E_HIP;
  if (data_bytes && hipMemcpyAsync(s.d_data, data, data_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return RGB_E_HIP;
  /* gaps between records (none when laid out by rgb_wal_layout) read back as zeros */
  if (hipMemsetAsync(s.d_frame, 0, out_bytes, st) != hipSuccess) return RGB_E_HIP;
  int rc = rgb_wal_frame_device(ctx, s.d_records, n, s.d_data, data_bytes, s.d_frame, out_bytes, nullptr, flags, st);
  if (rc) return rc;
  if (hipMemcpyAsync(out, s.d_frame, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return RGB_E_HIP;
  return hipStreamSynchronize(st) == hipSuccess ? RGB_OK : RGB_E_HIP;
}

/* rgb_wal_layout and rgb_wal_scan are plain host code: rgb_wal_host.cpp */

extern "C" int rgb_wal_validate(rgb_ctx *ctx, const void *bytes, uint64_t n_bytes, const rgb_wal_scanned *recs,
                                uint32_t n, uint32_t *n_ok, uint32_t *status) {
  if (!ctx || !n_ok || !status || (n && (!recs || !bytes))) return RGB_E_INVAL;
  *n_ok = n;
  *status = RGB_WAL_CLEAN;
  if (n == 0) return RGB_OK;
  std::vector<rgb_wal_entry> entries(n);
  std::vector<uint32_t> sums(n);
  for (uint32_t i = 0; i < n; ++i) {
    entries[i].index = recs[i].index; entries[i].term = recs[i].term;
    entries[i].data_offset = recs[i].data_offset; entries[i].data_len = recs[i].data_len; entries[i]._pad = 0;
  }
  int rc = rgb_wal_adler32(ctx, entries.data(), n, bytes, n_bytes, sums.data());
  if (rc) return rc;
  const unsigned char *b = (const unsigned char *)bytes;
  for (uint32_t i = 0; i < n; ++i) {
    if (!(recs[i].flags & RGB_WAL_REC_VALIDATE)) continue;
    if (recs[i].checksum == 0 || recs[i].checksum == sums[i]) continue;   /* validate_checksum/4 :1022-1033 */
    /* is_last_record/3 (:994-1010): 104 zero bits behind it, or fewer than 13 bytes to the end */
    *n_ok = i;
    const uint64_t rest = recs[i].next_offset;
    if (rest > n_bytes) return RGB_E_INVAL;               /* a descriptor that points past the file */
    bool last = true;
    if (n_bytes - rest >= 13) {
      for (int k = 0; k < 13; ++k) if (b[rest + k] != 0) { last = false; break; }
    }
    *status = last ? RGB_WAL_DROPPED_LAST : RGB_WAL_CORRUPT;
    return RGB_OK;
  }
  return RGB_OK;
}
