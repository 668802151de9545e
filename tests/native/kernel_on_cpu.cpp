/*
 * kernel_on_cpu.cpp -- TEST INFRASTRUCTURE: the device transition code of ra_amd/csrc/rgb_kernels.hip
 * compiled as x86 C++ (tests/native/fake_hip/hip/hip_runtime.h) and run one lane at a time.
 *
 * What runs is the product source itself: rgb_pack_kernel / rgb_unpack_kernel (one lane per server) and
 * process_message<N, KIND, false> (one lane per message; KIND = -1 is the generic path of rgb_tick_kernel,
 * KIND >= 0 the clause-folded specialisations the class-dispatch kernel instantiates) -- and, through the
 * product's own launchers, the whole kernels: every lane of a block is a fiber, so the LDS staging, the
 * cooperative hot-line fetch, the class dispatch, the checksum / leaderboard kernels and the load generator
 * execute as on one workgroup.  What the CPU cannot show is timing, occupancy and cross-workgroup races.
 */
#define RGB_HOST_EMULATION 1
/* Split build (tests/conftest.py): this unit is compiled once per group size with -DRGB_EMU_ONLY_N=<n> -- only that N
 * is instantiated and every external name carries the suffix __N<n> (emu_rename.h); kernel_dispatch_on_cpu.cpp owns
 * the plain names and forwards by n_members.  Eight small units in parallel instead of one three-minute unit. */
#ifdef RGB_EMU_ONLY_N
#define RGB_X_ONLY_N RGB_EMU_ONLY_N
#include "emu_rename.h"
#endif
#include <stdlib.h>
#include <hip/hip_runtime.h>
#ifndef RGB_EMU_ONLY_N
#include "emu_runtime.inc"
#endif
#include "../../ra_amd/csrc/rgb_kernels.hip"

namespace {
struct Emu {
  rgb_dev dev;
  rgb_rpc *slots;
  u32 slot_cap;
};

template <int N>
void run_one(const rgb_dev &dev, const rgb_msg &m, u32 i, rgb_rpc *slots, Dec &d, int specialised) {
  ulonglong2 q[4];
  memcpy(q, &m, sizeof q);
#define EMU_CASE(K) case K: process_message<N, K, false>(dev, q[0], q[1], q[2], q[3], i, slots, 0, 0, d); return;
  if (specialised) {
    switch (m.kind) {
      EMU_CASE(RGB_MSG_AER) EMU_CASE(RGB_MSG_AER_REPLY) EMU_CASE(RGB_MSG_REQUEST_VOTE) EMU_CASE(RGB_MSG_VOTE_RESULT)
      EMU_CASE(RGB_MSG_WRITTEN) EMU_CASE(RGB_MSG_PIPELINE_RPCS) EMU_CASE(RGB_MSG_APPEND) EMU_CASE(RGB_MSG_AWAIT_TIMEOUT)
      EMU_CASE(RGB_MSG_ELECTION_TIMEOUT) EMU_CASE(RGB_MSG_PRE_VOTE_RPC) EMU_CASE(RGB_MSG_PRE_VOTE_RESULT)
      EMU_CASE(RGB_MSG_SNAPSHOT_WRITTEN) EMU_CASE(RGB_MSG_HEARTBEAT_RPC) EMU_CASE(RGB_MSG_HEARTBEAT_REPLY)
      EMU_CASE(RGB_MSG_CONSISTENT_QUERY)
      default: break;
    }
  }
#undef EMU_CASE
  process_message<N, -1, false>(dev, q[0], q[1], q[2], q[3], i, slots, 0, 0, d);
}
}  // namespace

extern "C" {

void *emu_new(uint32_t n_groups, uint32_t n_members, uint32_t max_runs, uint32_t max_pipeline_count,
              uint32_t max_aer_batch) {
  Emu *e = (Emu *)calloc(1, sizeof(Emu));
  rgb_dev &d = e->dev;
  const size_t S = (size_t)n_groups * n_members;
  d.n_servers = (u32)S; d.n_members = n_members; d.max_runs = max_runs;
  d.peer_stride = rgb_peer_stride(n_members);
  d.max_pipeline_count = max_pipeline_count ? max_pipeline_count : 4096;
  d.max_aer_batch = max_aer_batch ? max_aer_batch : 128;
  d.hot = (u64 *)calloc(S * RGB_HOT_WORDS, sizeof(u64));
  d.peers = (u64 *)calloc(S * d.peer_stride, sizeof(u64));
  d.runs = (u64 *)calloc(S * max_runs * 2, sizeof(u64));
  d.cond = (u64 *)calloc(S * 4, sizeof(u64));
  d.qry = (u64 *)calloc(S * RGB_QRY_WORDS, sizeof(u64));
  d.seq_stride = (((n_groups + RGB_TRAIN_SHARDS - 1u) / RGB_TRAIN_SHARDS) * n_members + 255u) & ~255u;
  d.seq = (unsigned char *)calloc((size_t)d.seq_stride * RGB_TRAIN_SHARDS, 1);
  return e;
}

void emu_free(void *h) {
  Emu *e = (Emu *)h;
  free(e->dev.hot); free(e->dev.peers); free(e->dev.runs); free(e->dev.cond); free(e->dev.qry); free(e->dev.seq);
  free(e->slots);
  free(e);
}

void emu_set_state(void *h, uint32_t first, uint32_t n, const rgb_server_state *in) {
  Emu *e = (Emu *)h;
  blockDim = dim3(1); gridDim = dim3(n);
  for (u32 k = 0; k < n; ++k) { blockIdx = dim3(k); threadIdx = dim3(0); rgb_pack_kernel(e->dev, in, first, n); }
}

void emu_get_state(void *h, uint32_t first, uint32_t n, rgb_server_state *out) {
  Emu *e = (Emu *)h;
  blockDim = dim3(1); gridDim = dim3(n);
  for (u32 k = 0; k < n; ++k) { blockIdx = dim3(k); threadIdx = dim3(0); rgb_unpack_kernel(e->dev, out, first, n); }
}

/* messages in order, one lane each; rpc records compacted by (message, slot) like rgb_collect does */
int emu_step(void *h, const rgb_msg *msgs, uint32_t n, rgb_decision *dec, rgb_rpc *rpcs, uint32_t rpc_cap,
             uint32_t *n_rpcs_out, int specialised) {
  Emu *e = (Emu *)h;
  const u32 N = e->dev.n_members, per = N > 1 ? N - 1 : 1;
  if ((size_t)n * per > e->slot_cap) {
    free(e->slots);
    e->slot_cap = n * per;
    e->slots = (rgb_rpc *)calloc(e->slot_cap, sizeof(rgb_rpc));
  }
  u32 out = 0;
  for (u32 i = 0; i < n; ++i) {
    Dec d;
    memset(&d, 0, sizeof d);
    switch (N) {
#define EMU_N(NN) case NN: run_one<NN>(e->dev, msgs[i], i, e->slots, d, specialised); break;
#ifdef RGB_EMU_ONLY_N
      EMU_N(RGB_EMU_ONLY_N)
#else
      EMU_N(1) EMU_N(2) EMU_N(3) EMU_N(4) EMU_N(5) EMU_N(6) EMU_N(7) EMU_N(8)
#endif
#undef EMU_N
      default: return -1;
    }
    memcpy(&dec[i], &d, sizeof(rgb_decision));
    for (u32 k = 0; k < dec[i].n_rpcs; ++k) {
      if (out < rpc_cap) rpcs[out] = e->slots[(size_t)i * per + k];
      out++;
    }
  }
  *n_rpcs_out = out;
  return 0;
}

/* ---- whole kernels through the product's own launchers (grid computation, class dispatch, LDS staging,
 * cooperative hot-line fetch) ---- */
static int compact_rpcs(Emu *e, const rgb_decision *dec, uint32_t n, rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_out) {
  const u32 N = e->dev.n_members, per = N > 1 ? N - 1 : 1;
  u32 out = 0;
  for (u32 i = 0; i < n; ++i)
    for (u32 k = 0; k < dec[i].n_rpcs; ++k) {
      if (out < rpc_cap) rpcs[out] = e->slots[(size_t)i * per + k];
      out++;
    }
  *n_out = out;
  return 0;
}
static void need_slots(Emu *e, uint32_t n) {
  const u32 N = e->dev.n_members, per = N > 1 ? N - 1 : 1;
  if ((size_t)n * per > e->slot_cap) {
    free(e->slots);
    e->slot_cap = n * per;
    e->slots = (rgb_rpc *)calloc(e->slot_cap, sizeof(rgb_rpc));
  }
}

/* cls = -1: the kind-generic kernel; 0..3: the single-kind kernels */
int emu_launch_tick(void *h, int cls, const rgb_msg *msgs, uint32_t n, rgb_decision *dec, rgb_rpc *rpcs,
                    uint32_t rpc_cap, uint32_t *n_rpcs_out) {
  Emu *e = (Emu *)h;
  need_slots(e, n);
  int rc = rgb_launch_tick(e->dev, cls, msgs, n, nullptr, dec, e->slots, 0, 0, nullptr);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i) rgb_decision_expand(&dec[i]);     /* device streams hold compact records */
  return compact_rpcs(e, dec, n, rpcs, rpc_cap, n_rpcs_out);
}

/* messages ordered by clause family, counts per class: the class-dispatch kernel, one launch */
int emu_launch_classes(void *h, const rgb_msg *msgs, const uint32_t *counts, uint32_t n, rgb_decision *dec,
                       rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs_out) {
  Emu *e = (Emu *)h;
  need_slots(e, n);
  int rc = rgb_launch_tick_classes(e->dev, msgs, counts, nullptr, n, dec, e->slots, 0, 0, nullptr);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i) rgb_decision_expand(&dec[i]);
  return compact_rpcs(e, dec, n, rpcs, rpc_cap, n_rpcs_out);
}

/* the tick the load generator just wrote: class sizes come from its per-family totals (device side) */
int emu_launch_classes_dev(void *h, const rgb_msg *msgs, const uint32_t *family_totals, uint32_t max_msgs,
                           rgb_decision *dec) {
  Emu *e = (Emu *)h;
  need_slots(e, max_msgs);
  int rc = rgb_launch_tick_classes(e->dev, msgs, nullptr, family_totals, max_msgs, dec, e->slots, 0, 0, nullptr);
  if (rc) return rc;
  u32 total = 0;
  for (unsigned f = 0; f < RGB_N_FAMILIES; ++f) total += family_totals[f];
  for (uint32_t i = 0; i < total && i < max_msgs; ++i) rgb_decision_expand(&dec[i]);
  return 0;
}

int emu_launch_pack(void *h, uint32_t first, uint32_t n, const rgb_server_state *in) {
  return rgb_launch_pack(((Emu *)h)->dev, in, first, n, nullptr);
}
int emu_launch_unpack(void *h, uint32_t first, uint32_t n, rgb_server_state *out) {
  return rgb_launch_unpack(((Emu *)h)->dev, out, first, n, nullptr);
}
int emu_launch_checksum(void *h, uint32_t first, uint32_t n, uint64_t *out) {
  return rgb_launch_checksum(((Emu *)h)->dev, first, n, (u64 *)out, nullptr);
}
int emu_launch_leaderboard(void *h, rgb_leaderboard_row *rows) {
  return rgb_launch_leaderboard(((Emu *)h)->dev, rows, nullptr);
}
/* the load generator: scratch = emu_synth_scratch_words(h) u32, kind_counts = RGB_MSG_KIND_MAX + 1 u32,
 * bucket_counts (may be NULL) = RGB_N_BUCKETS u32 */
uint32_t emu_synth_scratch_words(void *h) { const rgb_dev &d = ((Emu *)h)->dev; return rgb_synth_scratch_words(d.n_servers / d.n_members); }
int emu_launch_synth(void *h, uint64_t seed, uint64_t tick, rgb_msg *msgs, uint32_t *scratch, uint32_t *kind_counts,
                     uint32_t *n_out, uint32_t *bucket_counts) {
  return rgb_launch_synth(((Emu *)h)->dev, seed, tick, msgs, scratch, kind_counts, n_out, bucket_counts, nullptr, nullptr, nullptr);
}

}  // extern "C"
