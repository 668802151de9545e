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
 * GROUP lanes per record (8 up to a mean payload of 320 bytes, 16 up to 1 KiB, then a wavefront); lane j handles
 * payload bytes [16 j, 16 j + 16) AS THEY LIE: one 16-byte load at the payload's own alignment (non-temporal), the
 * checksum on exactly those bytes (the piece that starts p bytes into the payload weighs (len - p - 16) a + b, no
 * masks), one 16-byte store at the destination's own alignment -- gfx950 global loads and stores take any byte
 * alignment.  A payload that does not end on a piece boundary ends with the piece [len - 16, len), which overlaps its
 * predecessor: the overlapped bytes are masked out of the sums (what is left weighs exactly b) and are stored twice
 * with the same value.  Payloads under 16 bytes go byte by byte.  The 24 fixed bytes are two unaligned vector stores by
 * the group's first lane once the checksum is known; HeaderData is copied byte per lane (3 bytes for a known writer).
 *
 * Rounds 1-3 framed through a FUNNEL (source-aligned loads, destination-aligned stores, the byte shift through DPP
 * and v_alignbyte, partial chunks as power-of-two store chains) because a first version with misaligned loads was
 * 30 % slower on 4 KiB payloads.  Measured again in round 4 with the funnel's instruction count out of the way
 * (profiles/EXPERIMENTS.md): this form is 24 % faster on 256-byte payloads (the funnel was vector-issue bound: ~900
 * instructions per eight records), 4 % faster on 4 KiB payloads and the same on 0.4-16 KiB mixes, at a third of the code. */

/* mean payload up to which a batch is framed with eight lanes per record (then sixteen up to 1 KiB, then a wavefront) */
#ifndef WAL_FRAME_EIGHT_MAX
#define WAL_FRAME_EIGHT_MAX 320u
#endif
typedef v4u v4u_any __attribute__((aligned(1)));
template <int GROUP>
__global__ __launch_bounds__(WAL_WAVES_PER_BLOCK * 64) void rgb_wal_frame_kernel(
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
        /* streamed past the caches for a record per wavefront.  Small records: a plain store -- every other 128-byte
         * line of the output holds a record boundary whose bytes arrive from other instructions, and a line a
         * non-temporal store pushed out early is written to memory twice (round 3: 396 -> 360 us per 2 M records) */
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
  /* lanes per record by the batch's mean payload: the per-record work that does not shrink with the payload --
   * descriptor, reduction, prefix -- is paid per WAVEFRONT instruction */
#define WAL_LAUNCH_FRAME(G)                                                                                          \
  hipLaunchKernelGGL(rgb_wal_frame_kernel<G>, dim3((n + (WAL_WAVES_PER_BLOCK * 64 / G) - 1) / (WAL_WAVES_PER_BLOCK * 64 / G)), \
                     dim3(WAL_WAVES_PER_BLOCK * 64), 0, st, (const rgb_wal_record *)d_records, n,                   \
                     (const unsigned char *)d_data, (unsigned char *)d_out, (u32 *)d_checksums, flags)
  if (data_bytes / n <= WAL_FRAME_EIGHT_MAX) WAL_LAUNCH_FRAME(8);
  else if (data_bytes / n < 1024u) WAL_LAUNCH_FRAME(16);
  else WAL_LAUNCH_FRAME(64);
#undef WAL_LAUNCH_FRAME
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
