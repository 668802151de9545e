/*
 * rgb_api.hip -- the C ABI of include/ra_gpu_batch.h on top of the HIP kernels.
 *
 * Host path (what the Erlang NIF drives): rgb_submit copies the caller's messages into a
 * slot of a pinned staging ring, then enqueues  H2D -> transition kernel(s) -> D2H  on the
 * context's stream and returns; rgb_collect waits on the slot's event and hands decisions back
 * in submission order.  Messages addressed to the same server inside one batch are serialised
 * into sub-ticks (round r holds every server's r-th message), one kernel launch per round.
 *
 * There is no CPU fallback: without a HIP device rgb_open fails with RGB_E_NODEVICE.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "rgb_internal.h"
#include "../../include/ra_gpu_batch_synth.h"

/* Host copies of the staging ring.  One rgb_submit + rgb_collect moves every message twice and every decision once
 * through the calling thread; with plain stores each destination line is first READ (read for ownership) -- a third of
 * the thread's memory traffic, and one thread of the host path runs at the memory bandwidth of its core.  Batches too
 * big to stay in the cache anyway are written with streaming stores (no ownership read; the device's DMA and the
 * consumer read them from memory either way). */
#define RGB_STREAM_COPY_MIN (1u << 20)      /* bytes: smaller batches are consumed from the cache */
#ifndef RGB_COPY_STREAM_MIN
#define RGB_COPY_STREAM_MIN (2u << 20)      /* bytes: batches below it keep their copy on the decision stream (two API calls and an event hand-over less on a latency-bound round trip) */
#endif
static inline void copy_msg(rgb_msg *dst, const rgb_msg *src, bool stream) {
#if defined(__x86_64__)
  if (stream) {                                              /* dst: a 64-byte slot of the pinned buffer (aligned) */
    const __m128i *q = reinterpret_cast<const __m128i *>(src);
    __m128i *d = reinterpret_cast<__m128i *>(dst);
    const __m128i a = _mm_loadu_si128(q), b = _mm_loadu_si128(q + 1), c = _mm_loadu_si128(q + 2), e = _mm_loadu_si128(q + 3);
    _mm_stream_si128(d, a); _mm_stream_si128(d + 1, b); _mm_stream_si128(d + 2, c); _mm_stream_si128(d + 3, e);
    return;
  }
#endif
  (void)stream;
  *dst = *src;
}
static inline void copy_out(void *dst, const void *src, size_t bytes) {
#if defined(__x86_64__)
  if (bytes >= RGB_STREAM_COPY_MIN && ((uintptr_t)dst & 15u) == 0 && (bytes & 63u) == 0) {
    const __m128i *q = reinterpret_cast<const __m128i *>(src);
    __m128i *d = reinterpret_cast<__m128i *>(dst);
    for (size_t k = 0; k < bytes / 16u; k += 4) {
      const __m128i a = _mm_loadu_si128(q + k), b = _mm_loadu_si128(q + k + 1), c = _mm_loadu_si128(q + k + 2), e = _mm_loadu_si128(q + k + 3);
      _mm_stream_si128(d + k, a); _mm_stream_si128(d + k + 1, b); _mm_stream_si128(d + k + 2, c); _mm_stream_si128(d + k + 3, e);
    }
    _mm_sfence();
    return;
  }
#endif
  memcpy(dst, src, bytes);
}

#define HIPCHK(ctx, expr)                          \
  do {                                             \
    hipError_t e__ = (expr);                       \
    if (e__ != hipSuccess) {                       \
      (ctx)->last_hip.store((int)e__, std::memory_order_relaxed); \
      return RGB_E_HIP;                            \
    }                                              \
  } while (0)

struct rgb_slot {
  rgb_msg *h_msgs = nullptr;        /* pinned: the batch in device order, then h_pos -- ONE copy to the device */
  rgb_decision *h_dec = nullptr;    /* pinned: WRITTEN BY THE DEVICE (rgb_results_kernel): full records, submission order */
  rgb_msg *d_msgs = nullptr;        /* (capacity x (64 + 4) bytes: d_pos lies behind the batch's messages) */
  rgb_decision *d_dec = nullptr;
  u32 *h_pos = nullptr, *d_pos = nullptr;   /* = (u32 *)(h_msgs + n) / (d_msgs + n): device position of submitted message i */
  rgb_rpc *d_rpcs = nullptr;        /* the fixed rpc slots: (N-1) per device position */
  rgb_rpc *h_rpcs = nullptr;        /* pinned: WRITTEN BY THE DEVICE: the batch's records compacted, (message, slot) order, msg_index set */
  u32 *h_nrpc = nullptr;            /* pinned header, written by the device: [0] records, [1] train error word, [2] over-count flag */
  u32 *d_res = nullptr;             /* rgb_launch_results' scratch: block sums + error word */
  /* sub-tick rounds as ONE train launch: stamps, per-round plan and row table (pinned staging + device) */
  unsigned char *h_stamps = nullptr, *d_stamps = nullptr;
  rgb_train_tick *h_plan = nullptr, *d_plan = nullptr;
  u32 *h_rows = nullptr, *d_rows = nullptr;
  u32 rows_cap = 0;                 /* words of h_rows / d_rows */
  u32 *d_ctl = nullptr;             /* RGB_TRAIN_CTL_WORDS: this slot's train error word + per-launch counters */
  /* rgb_submit_seq: the batch's range list (written events of more than two ranges, RGB_MF_SEQX) */
  u64 *h_ranges = nullptr, *d_ranges = nullptr;   /* pinned / device: (first, last) pairs; allocated with the first batch that has any */
  u32 ranges_cap = 0, n_ranges = 0;
  bool has_seqx = false;            /* the batch holds a RGB_MF_SEQX record: its written class runs in the written-only kernel that takes range lists */
  /* fail-safe: the servers this batch touches and their rows as they were before it (saved while a train is in flight) */
  u32 *h_touched = nullptr, *d_touched = nullptr;   /* pinned / device: ring_capacity ids */
  u32 n_touched = 0;
  void *d_undo = nullptr;           /* allocated with the first batch that needs it */
  bool has_undo = false;
  u32 n_rounds = 0;
  std::vector<u32> round_start;     /* n_rounds + 1: first device position of every round            */
  std::vector<u32> round_cc;        /* n_rounds x RGB_N_CLASSES: class sizes (one launch per round)  */
  bool used_train = false;
  int enqueue_error = 0;            /* the enqueue of this batch failed (RGB_E_*): rgb_collect reports it once */
  /* 0 free | 1 reserved by a producer (being filled) | 2 in flight (published) | 3 reserved by a consumer (being
   * copied out).  Producers take slots in ring order and consumers hand them back in any order, so fullness is the
   * state of the NEXT slot, not a count. */
  std::atomic<int> state{0};
  u32 n = 0;
  uint64_t tick = 0;
  hipEvent_t done = nullptr;
  hipEvent_t copied = nullptr;      /* the batch's messages are on the device (recorded on the copy stream) */
};

/* Threading contract of the staging ring: any number of threads may call rgb_submit -- they
 * prepare their batches in parallel (validation, rounds, the bucket sort into the slot's pinned buffer) and meet in
 * two short critical sections: submit_mu hands out the next ring slot and a ticket, enqueue_mu lets the tickets
 * through in order for the stream's work and the publication (batches reach the device in the order their submits
 * took their slots).  Any number may call rgb_collect: collect_mu covers waiting for the oldest batch, the size check
 * and taking the slot; the copy back to submission order runs outside it, so consumers copy different batches at
 * once; every batch is handed out exactly once.  The two sides meet in the slots' atomic states (release on publish,
 * acquire on collect; release when the slot is given back, acquire by the producer whose turn it is) and in
 * `in_flight` = published and not yet taken, on which rgb_wait parks a consumer (no polling).  head belongs to the
 * producers' lock, tail to the consumers'. */
#define RGB_SUBMIT_TRAIN_ROUNDS 16u    /* a batch with more sub-tick rounds than this takes one launch per round */
#define RGB_SUBMIT_TRAIN_MIN 4096u     /* and so does a small one: a train's fixed costs are those of a big launch */

struct rgb_ctx {
  rgb_config cfg;
  rgb_dev dev;
  hipStream_t stream = nullptr;
  std::atomic<int> last_hip{0};
  bool registered = false;
  /* state transfer staging */
  rgb_server_state *d_stage = nullptr;
  rgb_server_state *h_stage = nullptr;   /* pinned */
  u32 stage_cap = 0;
  /* ring */
  std::unique_ptr<rgb_slot[]> ring_mem;   /* rgb_slot holds an atomic: not movable */
  u32 ring_size = 0;
  u32 head = 0;                          /* guarded by submit_mu  */
  u32 tail = 0;                          /* guarded by collect_mu */
  std::atomic<u32> in_flight{0};         /* published and not yet taken by a consumer */
  /* producers prepare their batches in parallel; the stream's work is enqueued in the order the slots were taken */
  uint64_t next_ticket = 0;              /* guarded by submit_mu  */
  uint64_t enqueue_turn = 0;             /* guarded by enqueue_mu */
  std::mutex enqueue_mu, train_mu;
  std::condition_variable enqueue_cv;
  std::mutex submit_mu, collect_mu, state_mu;   /* state_mu: the h_stage / d_stage transfer staging */
  std::mutex wait_mu;
  std::condition_variable wait_cv;
  std::atomic<u32> wake_gen{0};
  u32 rpc_cap = 0;      /* records per ring slot = ring_capacity * rpc_stride */
  u32 rpc_stride = 1;   /* fixed rpc slots per message = max(n_members-1, 1) */

  /* snapshot / checksum scratch */
  rgb_leaderboard_row *d_rows = nullptr;
  u64 *d_sums = nullptr;
  void *d_lb_gather = nullptr;      /* rgb_leaderboard_allgather_host: this rank's padded rows | the gathered rows | status words */
  size_t lb_gather_bytes = 0;
  void *h_lb_pinned = nullptr;      /* pinned twin of the copy-outs (status words | gathered rows): the side stream's device-to-host
                                       copies land HERE, never in a caller's pageable buffer that a timed-out call has given back */
  size_t lb_pinned_bytes = 0;
  std::mutex lb_mu;                 /* .. one call at a time per context: it owns the buffer, the side stream, the event */
  hipStream_t lb_stream = nullptr;  /* the collective and its copy-out run HERE, outside the decision path's locks */
  /* the H2D copy of a BIG batch runs here and the decision stream waits for its event: PCIe is full duplex, and the
   * copy in of batch k + 1 then runs under the kernels of batch k and under its results kernel's stores to the host
   * (one stream serialised them: 131 072-message batches were 180 us in + 15 us of kernels + 170 us out) */
  hipStream_t copy_stream = nullptr;
  hipEvent_t lb_event = nullptr;    /* the snapshot on the context's stream -> the side stream */
  u32 synth_hint = 2;         /* rgb_synth_set_hint */
  u32 *d_synth = nullptr;     /* load-generator scratch (family and bucket counters) */
  unsigned char *d_synth_sent = nullptr;   /* load generator: messages addressed to every server so far, mod 256 (its stamps) */
  /* train launches */
  u32 *d_train_ctl = nullptr;           /* RGB_TRAIN_CTL_WORDS: sticky error flags | calibration scratch */
  unsigned char *d_seq_cnt = nullptr;   /* running stamp counters of rgb_train_stamp_device (a copy of dev.seq) */
  std::atomic<u32> n_submit_trains{0};  /* batches whose sub-tick rounds ran as one train launch */
  std::atomic<u32> trains_in_flight{0}; /* such batches enqueued and not yet verified by rgb_collect: while there are
                                           any, every batch saves an undo log (settle_trains) */
  std::atomic<u32> n_train_recoveries{0};
  std::atomic<u32> inject_fault{0};     /* tests: RGB_FAULT_* applied to the next train batch (rgb_debug_inject_train_fault) */
  u32 n_xcc = 0;                        /* XCCs of the device (calibration launch): a train block serves the shard of its XCC */
  u32 train_blocks = 0;                 /* blocks of a persistent train launch: every wavefront slot of the device */
  std::atomic<bool> train_dealt{false}; /* trains run in the dealt form (one block per row): the calibration launch was dealt
                                           round robin over 8 XCCs and no launch has failed its placement check since */
  int xcc_state = 0;                    /* 0 = not calibrated, 1 = usable, -1 = the XCC ids are not 0 .. n-1 with n | 8 */
};

/* the device plan of a train: one rgb_train_tick per tick */
struct rgb_train_plan {
  rgb_train_tick *d_ticks = nullptr;
  u32 *d_rows = nullptr;  /* [n_ticks][bpt / RGB_TRAIN_SHARDS]: class << 24 | row of the class */
  u32 n_ticks = 0;
  u32 bpt = 0;            /* blocks per tick: RGB_TRAIN_SHARDS x the longest tick's rows */
  u32 snap_every = 0;     /* > 0: ticks k x snap_every (k >= 1) carry the rows of a leaderboard snapshot in front of them */
  bool on_device = false; /* rgb_train_plan_create_device: the tables are filled by rgb_train_plan_build_device; the table's
                             stride is the rows BOUND of a tick */
  std::vector<u32> rows_fit;  /* on_device: the rows of every tick the host has been TOLD (rgb_train_plan_fit), else
                                 0xFFFFFFFF: a launch whose ticks are all known takes their rows as its grid, not the bound */
};

/* A tick ordered by clause family: ONE launch of the class-dispatch kernel. */
static int launch_tick_classes(rgb_ctx *ctx, const rgb_dev &dev, const rgb_msg *m, rgb_decision *d, rgb_rpc *rpcs,
                               const u32 counts[RGB_N_CLASSES], u32 rpc_slot_base, u32 msg_index_base,
                               hipStream_t main) {
  int rc = rgb_launch_tick_classes(dev, m, counts, nullptr, 0, d, rpcs, rpc_slot_base, msg_index_base, main);
  if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

extern "C" void rgb_wal_release(rgb_ctx *ctx);   /* rgb_wal.hip: staging buffers of the host-buffer form */

extern "C" {

uint32_t rgb_abi_version(void) { return RGB_ABI_VERSION; }

size_t rgb_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(rgb_msg);
    case 1: return sizeof(rgb_decision);
    case 2: return sizeof(rgb_rpc);
    case 3: return sizeof(rgb_server_state);
    case 4: return sizeof(rgb_leaderboard_row);
    case 5: return sizeof(rgb_config);
    case 6: return sizeof(rgb_view);
    default: return 0;
  }
}

const char *rgb_strerror(int code) {
  switch (code) {
    case RGB_OK: return "ok";
    case RGB_E_INVAL: return "invalid argument";
    case RGB_E_NOMEM: return "out of memory";
    case RGB_E_HIP: return "HIP runtime error (see rgb_last_hip_error)";
    case RGB_E_STATE: return "call out of order";
    case RGB_E_FULL: return "staging ring full";
    case RGB_E_EMPTY: return "nothing submitted";
    case RGB_E_UNSUPPORTED: return "unsupported";
    case RGB_E_NODEVICE: return "no HIP device (there is no CPU fallback)";
    case RGB_E_COMM: return "RCCL error (see rgb_comm_last_error)";
    default: return "unknown error";
  }
}

void rgb_default_config(rgb_config *cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof *cfg);
  cfg->abi_version = RGB_ABI_VERSION;
  cfg->device = 0;
  cfg->max_runs = 8;
  cfg->ring_slots = 4;
  cfg->ring_capacity = 65536;
  cfg->max_pipeline_count = RGB_DEFAULT_MAX_PIPELINE_COUNT;
  cfg->max_aer_batch = RGB_AER_CHUNK_SIZE;
  cfg->flags = 0;
}

int rgb_last_hip_error(const rgb_ctx *ctx) { return ctx ? ctx->last_hip.load(std::memory_order_relaxed) : 0; }

/* The host-buffer state calls (upload / download / snapshot / checksum) share staging buffers (state_mu) and put
 * work on the context's stream: they take the enqueue lock as well, so their work never lands between the rounds of
 * a batch another thread is enqueuing -- it is ordered against WHOLE batches, after every rgb_submit that has
 * returned.  Lock order: state_mu, collect_mu, enqueue_mu (rgb_submit holds enqueue_mu alone, rgb_collect collect_mu
 * then enqueue_mu). */
static int settle_trains(rgb_ctx *ctx);
struct rgb_stream_turn {
  std::lock_guard<std::mutex> a, c, b;
  int rc;      /* a failed train launch among the batches in flight is repaired first (settle_trains) */
  explicit rgb_stream_turn(rgb_ctx *ctx) : a(ctx->state_mu), c(ctx->collect_mu), b(ctx->enqueue_mu), rc(settle_trains(ctx)) {}
};

static void free_slot(rgb_slot &s) {
  if (s.h_msgs) (void)hipHostFree(s.h_msgs);
  if (s.h_dec) (void)hipHostFree(s.h_dec);
  if (s.d_msgs) (void)hipFree(s.d_msgs);
  if (s.d_dec) (void)hipFree(s.d_dec);
  if (s.d_rpcs) (void)hipFree(s.d_rpcs);
  if (s.h_rpcs) (void)hipHostFree(s.h_rpcs);
  if (s.h_nrpc) (void)hipHostFree(s.h_nrpc);
  if (s.d_res) (void)hipFree(s.d_res);
  if (s.h_stamps) (void)hipHostFree(s.h_stamps);
  if (s.d_stamps) (void)hipFree(s.d_stamps);
  if (s.h_plan) (void)hipHostFree(s.h_plan);
  if (s.d_plan) (void)hipFree(s.d_plan);
  if (s.h_rows) (void)hipHostFree(s.h_rows);
  if (s.d_rows) (void)hipFree(s.d_rows);
  if (s.d_ctl) (void)hipFree(s.d_ctl);
  if (s.h_ranges) (void)hipHostFree(s.h_ranges);
  if (s.d_ranges) (void)hipFree(s.d_ranges);
  s.h_ranges = s.d_ranges = nullptr; s.ranges_cap = 0;
  if (s.h_touched) (void)hipHostFree(s.h_touched);
  if (s.d_touched) (void)hipFree(s.d_touched);
  if (s.d_undo) (void)hipFree(s.d_undo);
  if (s.done) (void)hipEventDestroy(s.done);
  if (s.copied) (void)hipEventDestroy(s.copied);
  s.copied = nullptr;
  s.h_msgs = nullptr; s.h_dec = nullptr; s.d_msgs = nullptr; s.d_dec = nullptr; s.d_rpcs = nullptr; s.h_rpcs = nullptr;
  s.h_nrpc = nullptr; s.d_res = nullptr;
  s.h_stamps = s.d_stamps = nullptr; s.h_plan = s.d_plan = nullptr; s.h_rows = s.d_rows = nullptr; s.done = nullptr;
  s.d_ctl = nullptr; s.h_touched = s.d_touched = nullptr; s.d_undo = nullptr;
  s.h_pos = s.d_pos = nullptr;
}

void rgb_close(rgb_ctx *ctx) {
  if (!ctx) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  rgb_wal_release(ctx);
  for (u32 k = 0; k < ctx->ring_size; ++k) free_slot(ctx->ring_mem[k]);
  if (ctx->dev.dbg_buf) (void)hipFree(ctx->dev.dbg_buf);
  if (ctx->dev.hot) (void)hipFree(ctx->dev.hot);
  if (ctx->dev.peers) (void)hipFree(ctx->dev.peers);
  if (ctx->dev.runs) (void)hipFree(ctx->dev.runs);
  if (ctx->dev.cond) (void)hipFree(ctx->dev.cond);
  if (ctx->dev.qry) (void)hipFree(ctx->dev.qry);
  if (ctx->dev.seq) (void)hipFree(ctx->dev.seq);
  if (ctx->d_stage) (void)hipFree(ctx->d_stage);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->d_rows) (void)hipFree(ctx->d_rows);
  if (ctx->d_sums) (void)hipFree(ctx->d_sums);
  if (ctx->lb_stream) { (void)hipStreamSynchronize(ctx->lb_stream); (void)hipStreamDestroy(ctx->lb_stream); }
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->d_lb_gather) (void)hipFree(ctx->d_lb_gather);
  if (ctx->h_lb_pinned) (void)hipHostFree(ctx->h_lb_pinned);
  if (ctx->lb_event) (void)hipEventDestroy(ctx->lb_event);
  if (ctx->d_synth) (void)hipFree(ctx->d_synth);
  if (ctx->d_synth_sent) (void)hipFree(ctx->d_synth_sent);
  if (ctx->d_train_ctl) (void)hipFree(ctx->d_train_ctl);
  if (ctx->d_seq_cnt) (void)hipFree(ctx->d_seq_cnt);

  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int rgb_open(const rgb_config *cfg_in, rgb_ctx **out) {
  if (!out) return RGB_E_INVAL;
  *out = nullptr;
  rgb_config cfg;
  if (cfg_in) cfg = *cfg_in; else rgb_default_config(&cfg);
  if (cfg.abi_version != RGB_ABI_VERSION) return RGB_E_INVAL;
  if (cfg.max_runs < 2 || cfg.max_runs > RGB_MAX_RUNS) return RGB_E_INVAL;
  if (cfg.ring_slots < 1 || cfg.ring_slots > 64 || cfg.ring_capacity < 1) return RGB_E_INVAL;
  if (cfg.max_pipeline_count == 0) cfg.max_pipeline_count = RGB_DEFAULT_MAX_PIPELINE_COUNT;
  if (cfg.max_aer_batch == 0) cfg.max_aer_batch = RGB_AER_CHUNK_SIZE;
  if (cfg.max_aer_batch > 65535) return RGB_E_INVAL;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return RGB_E_NODEVICE;
  if (cfg.device < 0 || cfg.device >= ndev) return RGB_E_INVAL;
  rgb_ctx *ctx = new (std::nothrow) rgb_ctx();
  if (!ctx) return RGB_E_NOMEM;
  ctx->cfg = cfg;
  memset(&ctx->dev, 0, sizeof ctx->dev);
  e = hipSetDevice(cfg.device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);

  if (e != hipSuccess) { delete ctx; return RGB_E_HIP; }
  *out = ctx;
  return RGB_OK;
}

uint32_t rgb_n_servers(const rgb_ctx *ctx) { return ctx ? ctx->dev.n_servers : 0; }

static int alloc_slot(rgb_ctx *ctx, rgb_slot &s) {
  const u32 cap = ctx->cfg.ring_capacity;
  /* messages + positions: one pinned block, one device block, one copy per batch (the positions of a batch of n
   * messages lie behind its n-th message) */
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_msgs, (size_t)cap * (sizeof(rgb_msg) + sizeof(u32)), hipHostMallocDefault));
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_dec, (size_t)cap * sizeof(rgb_decision), hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_msgs, (size_t)cap * (sizeof(rgb_msg) + sizeof(u32))));
  HIPCHK(ctx, hipMalloc((void **)&s.d_dec, (size_t)cap * sizeof(rgb_decision)));
  s.h_pos = reinterpret_cast<u32 *>(s.h_msgs + cap); s.d_pos = reinterpret_cast<u32 *>(s.d_msgs + cap);
  HIPCHK(ctx, hipMalloc((void **)&s.d_rpcs, (size_t)ctx->rpc_cap * sizeof(rgb_rpc)));
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_rpcs, (size_t)ctx->rpc_cap * sizeof(rgb_rpc), hipHostMallocDefault));
  HIPCHK(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_nrpc, 64, hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_res, ((size_t)rgb_results_blocks(cap) + 1u) * sizeof(u32)));
  HIPCHK(ctx, hipMemsetAsync(s.d_res, 0, ((size_t)rgb_results_blocks(cap) + 1u) * sizeof(u32), ctx->stream));   /* (the error word) */
  /* fused sub-tick rounds (at most RGB_SUBMIT_TRAIN_ROUNDS of them per batch) */
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_stamps, cap, hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_stamps, cap));
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_plan, RGB_SUBMIT_TRAIN_ROUNDS * sizeof(rgb_train_tick), hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_plan, RGB_SUBMIT_TRAIN_ROUNDS * sizeof(rgb_train_tick)));
  s.rows_cap = (cap / 32u + 2u * RGB_N_CLASSES * RGB_SUBMIT_TRAIN_ROUNDS + 64u);   /* every row holds >= 32 x 8 messages or closes a class */
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_rows, (size_t)s.rows_cap * RGB_SUBMIT_TRAIN_ROUNDS * sizeof(u32), hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_rows, (size_t)s.rows_cap * RGB_SUBMIT_TRAIN_ROUNDS * sizeof(u32)));
  HIPCHK(ctx, hipMalloc((void **)&s.d_ctl, RGB_TRAIN_CTL_WORDS * sizeof(u32)));
  /* on the context's stream: it is a non-blocking stream, which the null stream's memset would not be ordered with */
  HIPCHK(ctx, hipMemsetAsync(s.d_ctl, 0, RGB_TRAIN_CTL_WORDS * sizeof(u32), ctx->stream));
  HIPCHK(ctx, hipHostMalloc((void **)&s.h_touched, (size_t)cap * sizeof(u32), hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&s.d_touched, (size_t)cap * sizeof(u32)));
  return RGB_OK;
}

int rgb_upload_state(rgb_ctx *ctx, uint32_t first, uint32_t n, const rgb_server_state *in);

int rgb_register_groups(rgb_ctx *ctx, uint32_t n_groups, uint32_t n_members) {
  if (!ctx) return RGB_E_INVAL;
  if (ctx->registered) return RGB_E_STATE;
  if (n_members < 1 || n_members > RGB_MAX_MEMBERS || n_groups < 1) return RGB_E_INVAL;
  const uint64_t S64 = (uint64_t)n_groups * n_members;
  if (S64 > 0x7FFFFFFFull) return RGB_E_INVAL;
  const u32 S = (u32)S64;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  rgb_dev &d = ctx->dev;
  d.n_servers = S; d.n_members = n_members; d.max_runs = ctx->cfg.max_runs;
  d.peer_stride = rgb_peer_stride(n_members);
  d.max_pipeline_count = ctx->cfg.max_pipeline_count;
  d.max_aer_batch = ctx->cfg.max_aer_batch;
  d.dbg = 0; d.dbg_buf = nullptr;
  d.synth_hint = ctx->synth_hint;
  d.fuse_pipeline = (ctx->cfg.flags & RGB_CFG_FUSE_PIPELINE) ? 1u : 0u;
  d.seq_ranges = nullptr; d.n_seq_ranges = 0;
#ifdef RGB_PROFILE
  /* the profiling build only (libra_gpu_batch_prof.so): knobs from the environment */
  { const char *e = getenv("RGB_DEBUG"); d.dbg = e ? (u32)atoi(e) : 0u; }
  if (d.dbg & 16u) {
    HIPCHK(ctx, hipMalloc((void **)&d.dbg_buf, (size_t)(S / 64 + 16) * 8 * sizeof(u64)));
    HIPCHK(ctx, hipMemsetAsync(d.dbg_buf, 0, (size_t)(S / 64 + 16) * 8 * sizeof(u64), ctx->stream));
  }
#endif
#ifdef RGB_X_DECLINE_HIST
  /* EXPERIMENT build (tools/decline_hist.py): 3 classes x 32 reason counters */
  HIPCHK(ctx, hipMalloc((void **)&d.dbg_buf, 128 * sizeof(u64)));
  HIPCHK(ctx, hipMemsetAsync(d.dbg_buf, 0, 128 * sizeof(u64), ctx->stream));
#endif
#ifdef RGB_X_TRAIN_TIMELINE
  /* EXPERIMENT build (tools/train_timeline.py): 8 words per block of a train launch */
  HIPCHK(ctx, hipMalloc((void **)&d.dbg_buf, (size_t)(1u << 20) * 8 * sizeof(u64)));
  HIPCHK(ctx, hipMemsetAsync(d.dbg_buf, 0, (size_t)(1u << 20) * 8 * sizeof(u64), ctx->stream));
#endif
  HIPCHK(ctx, hipMalloc((void **)&d.hot, (size_t)S * RGB_HOT_WORDS * sizeof(u64)));
  HIPCHK(ctx, hipMalloc((void **)&d.peers, (size_t)S * d.peer_stride * sizeof(u64)));
  HIPCHK(ctx, hipMalloc((void **)&d.runs, (size_t)S * d.max_runs * 2 * sizeof(u64)));
  HIPCHK(ctx, hipMalloc((void **)&d.cond, (size_t)S * 4 * sizeof(u64)));
  HIPCHK(ctx, hipMalloc((void **)&d.qry, (size_t)S * RGB_QRY_WORDS * sizeof(u64)));
  HIPCHK(ctx, hipMemsetAsync(d.runs, 0, (size_t)S * d.max_runs * 2 * sizeof(u64), ctx->stream));
  d.seq_stride = (((n_groups + RGB_TRAIN_SHARDS - 1u) / RGB_TRAIN_SHARDS) * n_members * RGB_SEQ_SPREAD + 255u) & ~255u;
  HIPCHK(ctx, hipMalloc((void **)&d.seq, (size_t)d.seq_stride * RGB_TRAIN_SHARDS));
  HIPCHK(ctx, hipMemsetAsync(d.seq, 0, (size_t)d.seq_stride * RGB_TRAIN_SHARDS, ctx->stream));
  ctx->stage_cap = S < 16384u ? S : 16384u;
  HIPCHK(ctx, hipMalloc((void **)&ctx->d_stage, (size_t)ctx->stage_cap * sizeof(rgb_server_state)));
  HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_stage, (size_t)ctx->stage_cap * sizeof(rgb_server_state),
                            hipHostMallocDefault));
  HIPCHK(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)n_groups * sizeof(rgb_leaderboard_row)));
  HIPCHK(ctx, hipMalloc((void **)&ctx->d_sums, (size_t)S * sizeof(u64)));
  ctx->rpc_stride = n_members > 1 ? n_members - 1 : 1;
  ctx->rpc_cap = ctx->cfg.ring_capacity * ctx->rpc_stride;
  ctx->ring_mem.reset(new (std::nothrow) rgb_slot[ctx->cfg.ring_slots]);
  if (!ctx->ring_mem) return RGB_E_NOMEM;
  ctx->ring_size = ctx->cfg.ring_slots;
  for (u32 k = 0; k < ctx->ring_size; ++k) {
    rgb_slot &s = ctx->ring_mem[k];
    int rc = alloc_slot(ctx, s);
    if (rc) return rc;
  }
  ctx->registered = true;
  /* every server starts as ra_server:init/1 on an empty log (empty_state of the reference
   * tests, test/ra_server_SUITE.erl:4139-4149; new_peer/0 src/ra_server.erl:2990-2995) */
  std::vector<rgb_server_state> init(ctx->stage_cap);
  for (u32 base = 0; base < S; base += ctx->stage_cap) {
    u32 cnt = S - base < ctx->stage_cap ? S - base : ctx->stage_cap;
    for (u32 k = 0; k < cnt; ++k) {
      rgb_server_state &h = init[k];
      memset(&h, 0, sizeof h);
      h.snapshot_index = RGB_UNDEF; h.snapshot_term = RGB_UNDEF;
      h.pending_first = 1;   /* [0:0] is written: nothing pending (last_index + 1) */
      for (unsigned i = 0; i < n_members; ++i) h.next_index[i] = 1;
      h.role = RGB_ROLE_FOLLOWER; h.self = (uint8_t)((base + k) % n_members);
      h.n_members = (uint8_t)n_members; h.voted_for = RGB_NONE; h.leader_id = RGB_NONE;
      h.cond_leader = RGB_NONE; h.n_runs = 1;
      h.present_mask = (uint8_t)((1u << n_members) - 1u); h.voter_mask = h.present_mask;
      h.status_mask = 0xFF;
    }
    int rc = rgb_upload_state(ctx, base, cnt, init.data());
    if (rc) return rc;
  }
  return RGB_OK;
}

static int validate_state(const rgb_ctx *ctx, const rgb_server_state &h) {
  if (h.n_members != ctx->dev.n_members) return RGB_E_INVAL;
  if (h.self >= RGB_MAX_MEMBERS) return RGB_E_INVAL;
  if (h.role > RGB_ROLE_AWAIT_CONDITION || h.cond_reason > RGB_COND_WAL_DOWN_LEADER) return RGB_E_INVAL;
  if (h.n_runs > RGB_MAX_RUNS || h.votes > 15) return RGB_E_INVAL;
  if (h.voted_for != RGB_NONE && h.voted_for >= RGB_MAX_MEMBERS) return RGB_E_INVAL;
  if (h.leader_id != RGB_NONE && h.leader_id >= RGB_MAX_MEMBERS) return RGB_E_INVAL;
  if (h.cond_leader != RGB_NONE && h.cond_leader >= RGB_MAX_MEMBERS) return RGB_E_INVAL;
  if (h.first_index <= h.last_index) {
    if (h.n_runs == 0 || h.run_start[0] != h.first_index) return RGB_E_INVAL;
    for (unsigned r = 1; r < h.n_runs; ++r)
      if (!(h.run_start[r] > h.run_start[r - 1] && h.run_start[r] <= h.last_index)) return RGB_E_INVAL;
    if (h.run_term[h.n_runs - 1] != h.last_term) return RGB_E_INVAL;
  }
  /* sparse pending: old ranges ascending, non-empty, non-adjacent, strictly below the newest range */
  if (h.n_pending_old > 2) return RGB_E_INVAL;
  uint64_t above = h.pending_first;                /* first index of the range above the one being checked */
  for (int k = (int)h.n_pending_old - 1; k >= 0; --k) {
    const uint64_t s = h.pending_old[k][0], e = h.pending_old[k][1];
    if (s > e || e == RGB_UNDEF || !(e + 1 < above)) return RGB_E_INVAL;
    above = s;
  }
  return RGB_OK;
}

int rgb_upload_state(rgb_ctx *ctx, uint32_t first, uint32_t n, const rgb_server_state *in) {
  if (!ctx || (!in && n)) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  if ((uint64_t)first + n > ctx->dev.n_servers) return RGB_E_INVAL;
  for (u32 k = 0; k < n; ++k) {
    int rc = validate_state(ctx, in[k]);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  rgb_stream_turn turn(ctx);
  if (turn.rc) return turn.rc;
  for (u32 base = 0; base < n; base += ctx->stage_cap) {
    u32 cnt = n - base < ctx->stage_cap ? n - base : ctx->stage_cap;
    memcpy(ctx->h_stage, in + base, (size_t)cnt * sizeof(rgb_server_state));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage, ctx->h_stage, (size_t)cnt * sizeof(rgb_server_state),
                               hipMemcpyHostToDevice, ctx->stream));
    int rc = rgb_launch_pack(ctx->dev, ctx->d_stage, first + base, cnt, ctx->stream);
    if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return RGB_OK;
}

int rgb_download_state(rgb_ctx *ctx, uint32_t first, uint32_t n, rgb_server_state *out) {
  if (!ctx || (!out && n)) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  if ((uint64_t)first + n > ctx->dev.n_servers) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  rgb_stream_turn turn(ctx);
  if (turn.rc) return turn.rc;
  for (u32 base = 0; base < n; base += ctx->stage_cap) {
    u32 cnt = n - base < ctx->stage_cap ? n - base : ctx->stage_cap;
    int rc = rgb_launch_unpack(ctx->dev, ctx->d_stage, first + base, cnt, ctx->stream);
    if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage, ctx->d_stage, (size_t)cnt * sizeof(rgb_server_state),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out + base, ctx->h_stage, (size_t)cnt * sizeof(rgb_server_state));
  }
  return RGB_OK;
}

static int validate_msg(const rgb_ctx *ctx, const rgb_msg &m) {
  if (m.kind > RGB_MSG_KIND_MAX) return RGB_E_INVAL;
  if (m.kind == RGB_MSG_NOP) return RGB_OK;
  if (m.server >= ctx->dev.n_servers) return RGB_E_INVAL;
  if (m.from != RGB_NONE && m.from >= RGB_MAX_MEMBERS) return RGB_E_INVAL;
  if (m.kind == RGB_MSG_AER && m.n_run0 > m.n_entries) return RGB_E_INVAL;
  if (m.kind == RGB_MSG_WRITTEN && m.a > m.b) return RGB_E_INVAL;
  if (m.kind == RGB_MSG_WRITTEN && (m.flags & RGB_MF_SEQ2) &&
      !(m.run0_term <= m.run1_term && m.run1_term != RGB_UNDEF && m.run1_term + 1 < m.a)) return RGB_E_INVAL;
  if (m.kind == RGB_MSG_WRITTEN && (m.flags & RGB_MF_SEQX) && !(m.flags & RGB_MF_SEQ2)) return RGB_E_INVAL;
  return RGB_OK;
}
/* a RGB_MF_SEQX record against the batch's range list: entries c .. c + n_entries - 1 exist, are ranges, ascending and
 * non-adjacent, and the last one ends below the record's lower inline range */
static int validate_seqx(const rgb_msg &m, const uint64_t *ranges, uint32_t n_ranges) {
  if (m.kind != RGB_MSG_WRITTEN || !(m.flags & RGB_MF_SEQX)) return RGB_OK;
  if (!ranges || m.n_entries == 0 || m.c > n_ranges || (uint64_t)m.c + m.n_entries > n_ranges) return RGB_E_INVAL;
  uint64_t prev_last = 0; bool have = false;
  for (uint32_t k = 0; k < m.n_entries; ++k) {
    const uint64_t f = ranges[2 * (m.c + k)], l = ranges[2 * (m.c + k) + 1];
    if (f > l || l == RGB_UNDEF || (have && !(prev_last + 1 < f))) return RGB_E_INVAL;
    prev_last = l; have = true;
  }
  return prev_last + 1 < m.run0_term ? RGB_OK : RGB_E_INVAL;
}

static int train_scratch(rgb_ctx *ctx);

/* fault injection for the fail-safe tests (rgb_debug_inject_train_fault): applied to the next train batch */
#define RGB_FAULT_STAMP 1u   /* its first message carries a stamp that never comes up: RGB_TRAIN_ERR_SPIN            */
#define RGB_FAULT_SHARD 2u   /* two messages of one class and round swap shards: RGB_TRAIN_ERR_PLACEMENT            */

/* the rounds of a batch with one launch per round (the shape of every batch that is not a train, and the replay of
 * one whose train launch failed) */
/* the device view of one batch: the context's, plus the batch's own range list */
static rgb_dev slot_dev(const rgb_ctx *ctx, const rgb_slot &s) {
  rgb_dev d = ctx->dev;
  d.seq_ranges = s.n_ranges ? s.d_ranges : nullptr;
  d.n_seq_ranges = s.n_ranges;
  return d;
}

static int enqueue_rounds(rgb_ctx *ctx, rgb_slot &s) {
  const rgb_dev dv = slot_dev(ctx, s);
  auto fail = [&](int lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; };
  for (u32 r = 0; r < s.n_rounds; ++r) {
    const u32 off = s.round_start[r], cnt = s.round_start[r + 1] - s.round_start[r];
    /* every round, whatever its size, runs the class-dispatch kernel -- a path specialised per message kind, no
     * scratch for any group size (round 6: rounds under 4 096 messages used to take the kind-generic kernel, the one
     * kernel of the library that spills: 110 .. 1 402 VGPRs for groups of 5 .. 8) */
    u32 cc[RGB_N_CLASSES], real = 0, before_written = 0;
    for (int c = 0; c < RGB_N_CLASSES; ++c) {
      cc[c] = s.round_cc[(size_t)r * RGB_N_CLASSES + c];
      real += cc[c];
      if (c < 2) before_written += cc[c];
    }
    int lr;
    if (!s.has_seqx || cc[2] == 0) {
      lr = launch_tick_classes(ctx, dv, s.d_msgs + off, s.d_dec + off, s.d_rpcs, cc, off, off, ctx->stream);
      if (lr) return lr;
    } else {
      /* a batch of rgb_submit_seq: its written events may carry range lists (RGB_MF_SEQX), which the class kernel's
       * written path does not take -- the classes in front of the written class, the written class through the
       * written-only kernel that does, the classes behind it (each launch sees its own part of the round: the class
       * offsets of a plan count from the first class that has messages) */
      u32 head[RGB_N_CLASSES] = {0}, tail[RGB_N_CLASSES] = {0};
      head[0] = cc[0]; head[1] = cc[1];
      for (int c = 3; c < RGB_N_CLASSES; ++c) tail[c] = cc[c];
      if (before_written) {
        lr = launch_tick_classes(ctx, dv, s.d_msgs + off, s.d_dec + off, s.d_rpcs, head, off, off, ctx->stream);
        if (lr) return lr;
      }
      const u32 w0 = off + before_written, t0 = w0 + cc[2];
      lr = rgb_launch_tick(dv, RGB_TICK_CLS_WRITTEN_SEQX, s.d_msgs + w0, cc[2], nullptr, s.d_dec + w0, s.d_rpcs, w0, w0, ctx->stream);
      if (lr) return fail(lr);
      if (real > before_written + cc[2]) {
        lr = launch_tick_classes(ctx, dv, s.d_msgs + t0, s.d_dec + t0, s.d_rpcs, tail, t0, t0, ctx->stream);
        if (lr) return lr;
      }
    }
    if (real < cnt) {      /* NOP slots sort last: their empty decisions */
      lr = rgb_launch_tick(dv, RGB_TICK_CLS_NOP, s.d_msgs + off + real, cnt - real, nullptr, s.d_dec + off + real,
                           s.d_rpcs, off + real, off + real, ctx->stream);
      if (lr) return fail(lr);
    }
  }
  return RGB_OK;
}

/* what comes back, written by the device into the slot's pinned buffers (rgb_launch_results: no copy command): the
 * decisions in submission order, the rpc records compacted, the header (records, a train's error word -- the launch's
 * placement marks went into it: rgb_launch_train); then the slot's event */
static int enqueue_results(rgb_ctx *ctx, rgb_slot &s) {
  int lr = rgb_launch_results(s.d_dec, s.d_pos, s.n, ctx->cfg.ring_capacity, s.d_rpcs, ctx->rpc_stride, s.d_res, s.used_train ? s.d_ctl : nullptr,
                              s.h_dec, s.h_rpcs, s.h_nrpc, ctx->stream);
  if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
  HIPCHK(ctx, hipEventRecord(s.done, ctx->stream));
  return RGB_OK;
}

/* step 3 of rgb_submit (under enqueue_mu, in ticket order): everything the batch puts on the stream */
static int enqueue_batch(rgb_ctx *ctx, rgb_slot &s, bool as_train, u32 rows_max) {
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  const u32 n = s.n;
  s.h_nrpc[0] = 0; s.h_nrpc[1] = 0; s.h_nrpc[2] = 0;
  if (!n) { HIPCHK(ctx, hipEventRecord(s.done, ctx->stream)); return RGB_OK; }
  u32 fault = 0;
  if (as_train) {
    fault = ctx->inject_fault.exchange(0u, std::memory_order_relaxed);
    if (fault == RGB_FAULT_STAMP) s.h_stamps[0] = (unsigned char)(s.h_stamps[0] + 7u);   /* (round + 7: never comes up) */
    if (fault == RGB_FAULT_SHARD) {
      /* the first messages of two shards of one (round, class) change places (with their stamps and their places in
       * the permutation): the tick is still class-ordered, so the per-round replay computes it */
      bool done = false;
      for (u32 c = 0; c < RGB_N_PCLASSES && !done; ++c)
        for (u32 x = 0; x + 1 < RGB_TRAIN_SHARDS && !done; ++x) {
          const rgb_train_tick &t0 = s.h_plan[0];
          if (t0.cnt[c][x] == 0 || t0.cnt[c][x + 1] == 0) continue;
          const u32 a = t0.msg_base + t0.off[c][x], b = t0.msg_base + t0.off[c][x + 1];
          std::swap(s.h_msgs[a], s.h_msgs[b]); std::swap(s.h_stamps[a], s.h_stamps[b]);
          for (u32 i = 0; i < n; ++i) {                       /* (the submitted messages that sit at a and b) */
            if (s.h_pos[i] == a) s.h_pos[i] = b;
            else if (s.h_pos[i] == b) s.h_pos[i] = a;
          }
          done = true;
        }
    }
  }
  {
    const size_t bytes = (size_t)n * (sizeof(rgb_msg) + sizeof(u32));                               /* messages + h_pos */
    if (bytes >= RGB_COPY_STREAM_MIN) {
      /* (the slot's device buffers are free: its previous batch was collected, i.e. its event had completed) */
      HIPCHK(ctx, hipMemcpyAsync(s.d_msgs, s.h_msgs, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
      HIPCHK(ctx, hipEventRecord(s.copied, ctx->copy_stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    } else {
      HIPCHK(ctx, hipMemcpyAsync(s.d_msgs, s.h_msgs, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if (s.n_ranges)
    HIPCHK(ctx, hipMemcpyAsync(s.d_ranges, s.h_ranges, (size_t)s.n_ranges * 2u * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
  /* the undo log: the rows of the touched servers as they are before this batch -- while a train is in flight
   * (this batch included) a failed launch must be repairable */
  if (as_train || ctx->trains_in_flight.load(std::memory_order_acquire) != 0) {
    if (s.n_touched) {
      if (!s.d_undo) {
        const size_t servers = ctx->cfg.ring_capacity < ctx->dev.n_servers ? ctx->cfg.ring_capacity : ctx->dev.n_servers;
        HIPCHK(ctx, hipMalloc(&s.d_undo, servers * rgb_undo_pieces(ctx->dev) * 16u));
      }
      HIPCHK(ctx, hipMemcpyAsync(s.d_touched, s.h_touched, (size_t)s.n_touched * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
      int lr = rgb_launch_undo(ctx->dev, s.d_touched, s.n_touched, s.d_undo, 0, ctx->stream);
      if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
    }
    s.has_undo = true;
  }
  if (as_train) {
    HIPCHK(ctx, hipMemcpyAsync(s.d_stamps, s.h_stamps, n, hipMemcpyHostToDevice, ctx->stream));
    {
      int lr = rgb_launch_stamp_rounds(ctx->dev, s.d_msgs, n, s.d_stamps, ctx->stream);
      if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
    }
    HIPCHK(ctx, hipMemcpyAsync(s.d_plan, s.h_plan, (size_t)s.n_rounds * sizeof(rgb_train_tick), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(s.d_rows, s.h_rows, (size_t)s.n_rounds * rows_max * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(s.d_ctl, 0, sizeof(u32), ctx->stream));    /* this slot's own error word */
    int lr = rgb_launch_train(slot_dev(ctx, s), s.d_msgs, s.d_stamps, 0, s.d_plan, s.d_rows, s.n_rounds, rows_max * RGB_TRAIN_SHARDS,
                              s.d_dec, s.d_rpcs, 1, 0, s.d_ctl, ctx->n_xcc,
                              ctx->train_dealt.load(std::memory_order_relaxed) ? 0u : ctx->train_blocks, ctx->stream);
    if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
    s.used_train = true;
    ctx->trains_in_flight.fetch_add(1, std::memory_order_release);
    ctx->n_submit_trains.fetch_add(1, std::memory_order_relaxed);
  } else {
    int rc = enqueue_rounds(ctx, s);
    if (rc) return rc;
  }
  int rc = enqueue_results(ctx, s);
  if (rc && s.used_train) { s.used_train = false; ctx->trains_in_flight.fetch_sub(1, std::memory_order_release); }
  return rc;
}

/* pass 1 of rgb_submit, compiled per group size: the bucket key holds the server's shard = (server / n_members) mod 8,
 * a division by a constant here (a multiply) instead of one by a run-time value per message; the per-thread scratch
 * comes in as plain pointers (a thread_local std::vector is reached through its guard function at every use) */
struct submit_scan { u32 n_rounds = 0; bool any_nop = false, too_many = false, any_seqx = false; int bad = RGB_OK; };
extern "C++" {
template <unsigned NM>
static void submit_pass1(const rgb_ctx *ctx, const rgb_msg *msgs, u32 n, const uint64_t *ranges, uint32_t n_ranges,
                         uint16_t *seen, uint16_t *key_of, u32 *round_of, std::vector<u32> &touched, submit_scan &sc) {
  u32 n_rounds = sc.n_rounds;
  for (u32 i = 0; i < n; ++i) {
    const rgb_msg &m = msgs[i];
    int bad = validate_msg(ctx, m);
    if (!bad) bad = validate_seqx(m, ranges, n_ranges);
    if (bad) { sc.bad = bad; break; }
    if (m.kind == RGB_MSG_WRITTEN && (m.flags & RGB_MF_SEQX)) sc.any_seqx = true;
    u32 r = 0;
    if (m.kind != RGB_MSG_NOP) {
      uint16_t &c = seen[m.server];
      if (c == 0) touched.push_back(m.server);
      r = c;
      if (c == 0xFFFF) { sc.too_many = true; break; }
      c++;
    } else sc.any_nop = true;
    round_of[i] = r;
    key_of[i] = (uint16_t)rgb_bucket(m.kind, m.flags, m.kind != RGB_MSG_NOP ? m.server : 0u, NM);
    if (r + 1 > n_rounds) n_rounds = r + 1;
  }
  sc.n_rounds = n_rounds;
}
}  // extern "C++"

/* The sub-tick rounds of one batch run as ONE train launch (reference: the mailbox of a member is FIFO,
 * src/ra_server_proc.erl:1356-1397 -- a leader's N-1 replies land in one batch, so rounds > 1 are the normal shape):
 * round r = tick r of the train, every round in bucket order, the per-server sequence bytes order a server's
 * messages instead of a kernel boundary per round. */
/* rgb_submit in three steps, so that any number of producers prepare their batches IN PARALLEL:
 *   1. no lock: validation, the sub-tick rounds (thread-local scratch), the bucket counts;
 *   2. submit_mu, O(1): take the next ring slot and a ticket -- RGB_E_FULL when that slot is not free;
 *      then, no lock: the bucket sort of the batch into the slot's pinned buffer, the train plan;
 *   3. enqueue_mu, in ticket order: the stream's work (H2D, kernels, D2H, event), the train stamps (they count on from
 *      the previous batch), publication.  Batches reach the device in the order their submits took their slots.
 * Everything that can fail because of the INPUT fails before step 2; a HIP error in step 3 publishes the batch as
 * failed (rgb_collect returns the error once and the ring moves on). */
int rgb_submit(rgb_ctx *ctx, const rgb_msg *msgs, uint32_t n, uint64_t tick) {
  return rgb_submit_seq(ctx, msgs, n, tick, nullptr, 0);
}

int rgb_set_seq_ranges_device(rgb_ctx *ctx, const void *d_ranges, uint32_t n_ranges) {
  if (!ctx || (!d_ranges && n_ranges)) return RGB_E_INVAL;
  /* submit threads copy ctx->dev for their launches (slot_dev) under the enqueue lock: the list changes under it too */
  std::lock_guard<std::mutex> b(ctx->enqueue_mu);
  ctx->dev.seq_ranges = n_ranges ? (const u64 *)d_ranges : nullptr;
  ctx->dev.n_seq_ranges = n_ranges;
  return RGB_OK;
}

int rgb_submit_seq(rgb_ctx *ctx, const rgb_msg *msgs, uint32_t n, uint64_t tick, const uint64_t *ranges, uint32_t n_ranges) {
  if (!ctx || (!msgs && n) || (!ranges && n_ranges)) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  if (n > ctx->cfg.ring_capacity) return RGB_E_INVAL;
  /* the calling thread's current device may be any (one context per GPU, scheduler threads default to device 0):
   * everything below -- the one-off train calibration included -- runs on the context's */
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  /* ---- 1. ONE pass over the batch: validation, the rounds (round r = every server's r-th message of this batch,
   * in order; per-thread scratch) and every message's bucket key -- the later passes read 6 bytes per message, not 64 */
  thread_local std::vector<uint16_t> seen, key_of;
  thread_local std::vector<u32> touched, round_of;
  if (seen.size() < ctx->dev.n_servers) seen.assign(ctx->dev.n_servers, 0);
  round_of.resize(n);
  key_of.resize(n);
  touched.clear();
  if (touched.capacity() < n) touched.reserve(n);
  const unsigned n_members = ctx->dev.n_members;
  submit_scan sc;
  sc.n_rounds = n ? 1 : 0;
  switch (n_members) {
#define RGB_SCAN_N(NM) case NM: submit_pass1<NM>(ctx, msgs, n, ranges, n_ranges, seen.data(), key_of.data(), round_of.data(), touched, sc); break;
    RGB_SCAN_N(1) RGB_SCAN_N(2) RGB_SCAN_N(3) RGB_SCAN_N(4) RGB_SCAN_N(5) RGB_SCAN_N(6) RGB_SCAN_N(7) RGB_SCAN_N(8)
#undef RGB_SCAN_N
    default: return RGB_E_STATE;
  }
  u32 n_rounds = sc.n_rounds;
  const bool any_nop = sc.any_nop, too_many = sc.too_many, any_seqx = sc.any_seqx;
  const int bad = sc.bad;
  for (u32 t : touched) seen[t] = 0;
  if (bad) return bad;
  if (too_many) return RGB_E_UNSUPPORTED;
  /* several rounds, a batch worth a big launch, no NOP padding, a device that keeps a shard on one XCD: the rounds
   * run as ONE train launch, in bucket order (class, shard, success flag) -- a finer key of the same family order */
  bool as_train = n_rounds >= 2 && n_rounds <= RGB_SUBMIT_TRAIN_ROUNDS && n >= RGB_SUBMIT_TRAIN_MIN && !any_nop && !any_seqx &&
                  (ctx->cfg.flags & RGB_CFG_SUBMIT_TRAINS) && !(ctx->cfg.flags & RGB_CFG_ROUNDS_PER_LAUNCH);
  if (as_train) {
    std::lock_guard<std::mutex> tl(ctx->train_mu);          /* the one-off calibration uses the stream */
    const int ts = train_scratch(ctx);
    as_train = ts == RGB_OK;
    if (!as_train && getenv("RGB_TRACE_TRAIN_SETUP"))
      fprintf(stderr, "[rgb] rgb_submit: no train (setup rc %d, hip %d)\n", ts, ctx->last_hip.load());
  } else if (getenv("RGB_TRACE_TRAIN_SETUP")) {
    fprintf(stderr, "[rgb] rgb_submit: n %u rounds %u nop %d -> launch per round\n", n, n_rounds, (int)any_nop);
  }
  /* device order: by round, then by clause family = (message kind, success flag) (a round holds
   * at most one message per server, so its order is free; family-homogeneous wavefronts do not
   * diverge across clause families), stable inside a bucket */
  const u32 NK = as_train ? (u32)RGB_N_BUCKETS : (u32)RGB_N_FAMILIES;
  /* the family (kind rank, success flag) is the bucket key without its shard bits */
  auto family = [as_train](uint16_t key) -> u32 { return as_train ? (u32)key : (((u32)key >> 4) << 1) | ((u32)key & 1u); };
  std::vector<u32> bucket_counts;
  std::vector<u32> start(n_rounds + 1, 0);
  std::vector<u32> bucket((size_t)n_rounds * NK + 1, 0);
  const u32 *const ro = round_of.data();                 /* (plain pointers: see submit_pass1) */
  const uint16_t *const ko = key_of.data();
  /* (runs of one bucket -- a mailbox drain is clustered by kind -- are counted in a register: an increment per
   * message on ONE counter is a store-to-load chain, five cycles a message) */
  {
    size_t cur = (size_t)-1; u32 run = 0;
    for (u32 i = 0; i < n; ++i) {
      const size_t b = (size_t)ro[i] * NK + family(ko[i]) + 1;
      if (b != cur) { if (run) bucket[cur] += run; cur = b; run = 0; }
      run++;
    }
    if (run) bucket[cur] += run;
  }
  for (u32 r = 0; r < n_rounds; ++r) {                   /* a round = its buckets */
    u32 t = 0;
    for (u32 k = 0; k < NK; ++k) t += bucket[(size_t)r * NK + k + 1];
    start[r + 1] = start[r] + t;
  }
  if (as_train) bucket_counts.assign(bucket.begin() + 1, bucket.end());      /* per (round, bucket), before the scan */
  for (size_t b = 0; b < (size_t)n_rounds * NK; ++b) bucket[b + 1] += bucket[b];

  /* per round: the class sizes (what one launch per round needs -- the fall-back of a train and its replay after a
   * failed launch) and the span of device positions whose kind can emit rpc records.  A bucket holds one class, so
   * both follow from the bucket bounds (bucket[b] .. bucket[b + 1] before the scatter below moves them) */
  std::vector<u32> class_counts((size_t)n_rounds * RGB_N_CLASSES, 0);
  for (size_t b = 0; b < (size_t)n_rounds * NK; ++b) {
    const u32 b0 = bucket[b], b1 = bucket[b + 1];
    if (b1 == b0) continue;
    const u32 cls = (u32)(b % NK) / (as_train ? 2u * RGB_TRAIN_SHARDS : 2u);      /* kind rank; 15 = NOP */
    if (cls < RGB_N_CLASSES) class_counts[(b / NK) * RGB_N_CLASSES + cls] += b1 - b0;
  }
  /* ---- 2. the slot and the ticket.  From here to the publication nothing returns early and whatever throws is
   * caught: the ticket must be honoured (enqueue_turn advances, the slot is published -- as failed if need be) or
   * every later rgb_submit would block for ever ---- */
  rgb_slot *sp;
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> lk(ctx->submit_mu);
    sp = &ctx->ring_mem[ctx->head];
    int free_state = 0;
    /* acquire: a slot a consumer gave back (release store in rgb_collect) is really free */
    if (!sp->state.compare_exchange_strong(free_state, 1, std::memory_order_acquire)) return RGB_E_FULL;
    ctx->head = (ctx->head + 1) % ctx->ring_size;
    ticket = ctx->next_ticket++;
  }
  rgb_slot &s = *sp;
  int rc = RGB_OK;
  u32 rows_max = 0;
  try {
    s.h_pos = reinterpret_cast<u32 *>(s.h_msgs + n); s.d_pos = reinterpret_cast<u32 *>(s.d_msgs + n);
    const bool stream_copy = (size_t)n * sizeof(rgb_msg) >= RGB_STREAM_COPY_MIN;
    {
      size_t cur = (size_t)-1; u32 next = 0;                  /* (the current bucket's cursor lives in a register) */
      for (u32 i = 0; i < n; ++i) {
        const size_t b = (size_t)ro[i] * NK + family(ko[i]);
        if (b != cur) { if (cur != (size_t)-1) bucket[cur] = next; cur = b; next = bucket[b]; }
        const u32 p = next++;
        s.h_pos[i] = p;
        copy_msg(&s.h_msgs[p], &msgs[i], stream_copy);
        if (as_train) s.h_stamps[p] = (unsigned char)ro[i];   /* a train's stamps: the device adds the sequence bytes */
      }
      if (cur != (size_t)-1) bucket[cur] = next;
    }
#if defined(__x86_64__)
    if (stream_copy) _mm_sfence();                        /* (the streaming stores are ordered before the copy command) */
#endif
    s.n = n; s.tick = tick;
    s.n_ranges = 0; s.has_seqx = any_seqx;
    if (n_ranges) {                                            /* the batch's range list travels with it */
      if (s.ranges_cap < n_ranges) {
        if (s.h_ranges) (void)hipHostFree(s.h_ranges);
        if (s.d_ranges) (void)hipFree(s.d_ranges);
        s.h_ranges = s.d_ranges = nullptr; s.ranges_cap = 0;
        const u32 cap = n_ranges < 1024u ? 1024u : n_ranges;
        if (hipHostMalloc((void **)&s.h_ranges, (size_t)cap * 2u * sizeof(u64), hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&s.d_ranges, (size_t)cap * 2u * sizeof(u64)) != hipSuccess) {
          rc = RGB_E_NOMEM;
        } else s.ranges_cap = cap;
      }
      if (rc == RGB_OK) { memcpy(s.h_ranges, ranges, (size_t)n_ranges * 2u * sizeof(u64)); s.n_ranges = n_ranges; }
    }
    s.used_train = false; s.enqueue_error = 0; s.has_undo = false;
    s.n_rounds = n_rounds;
    s.round_start.swap(start);                                  /* no allocation: the vectors change hands */
    s.round_cc.swap(class_counts);
    s.n_touched = (u32)touched.size();
    if (s.n_touched) memcpy(s.h_touched, touched.data(), (size_t)s.n_touched * sizeof(u32));
    if (as_train) {                                            /* the plan of every round: slot-local, no lock */
      for (u32 r = 0; r < n_rounds; ++r) {
        const u32 rows = rgb_train_make_tick(bucket_counts.data() + (size_t)r * RGB_N_BUCKETS, n_members, &s.h_plan[r], nullptr, 0);
        if (rows > rows_max) rows_max = rows;
      }
      if (rows_max == 0 || rows_max > s.rows_cap) as_train = false;   /* more rows than the slot's table holds */
      for (u32 r = 0; as_train && r < n_rounds; ++r) {
        for (u32 k = 0; k < rows_max; ++k) s.h_rows[(size_t)r * rows_max + k] = 0xFFFFFFFFu;
        rgb_train_make_tick(bucket_counts.data() + (size_t)r * RGB_N_BUCKETS, n_members, &s.h_plan[r], s.h_rows + (size_t)r * rows_max, rows_max);
        s.h_plan[r].msg_base = s.round_start[r];
      }
    }
  } catch (...) {
    rc = RGB_E_NOMEM;
  }

  /* ---- 3. the stream's work, in ticket order ---- */
  {
    std::unique_lock<std::mutex> el(ctx->enqueue_mu);
    ctx->enqueue_cv.wait(el, [&] { return ctx->enqueue_turn == ticket; });
    if (rc == RGB_OK) {
      try {
        rc = enqueue_batch(ctx, s, as_train, rows_max);
      } catch (...) {
        rc = RGB_E_NOMEM;
      }
    }
    if (rc != RGB_OK) {
      /* published as failed: rgb_collect reports the error once and the ring moves on */
      s.enqueue_error = rc; s.n = 0; s.used_train = false; s.has_undo = false;
      (void)hipEventRecord(s.done, ctx->stream);
    }
    /* publish: everything written to the slot above happens-before the consumer's acquire load */
    s.state.store(2, std::memory_order_release);
    ctx->in_flight.fetch_add(1, std::memory_order_release);
    ctx->enqueue_turn += 1;
  }
  ctx->enqueue_cv.notify_all();
  { std::lock_guard<std::mutex> wl(ctx->wait_mu); }
  ctx->wait_cv.notify_one();
  return rc;
}

/* A train launch that failed (RGB_TRAIN_ERR_*: a dependency that never committed, a tick out of bucket order) has
 * applied SOME of its batch's messages, and every batch enqueued behind it ran on that state.  The engine repairs this
 * itself (reference semantics to keep: a member's messages apply in order, exactly once,
 * src/ra_server_proc.erl:1356-1397): the undo logs of the in-flight batches go back newest first -- every batch that
 * was enqueued while a train was in flight carries one -- which is the state before the failed batch, and the batches
 * run again, oldest first, with one launch per round.  Caller holds collect_mu and enqueue_mu. */
static int settle_trains(rgb_ctx *ctx) {
  if (ctx->trains_in_flight.load(std::memory_order_acquire) == 0) return RGB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  /* the published batches, oldest first (producers take slots in ring order) */
  u32 idx[64], m = 0, first_bad = 0xFFFFFFFFu;
  for (u32 k = 0; k < ctx->ring_size; ++k) {
    const u32 j = (ctx->tail + k) % ctx->ring_size;
    rgb_slot &s = ctx->ring_mem[j];
    if (s.state.load(std::memory_order_acquire) != 2) break;
    if (first_bad == 0xFFFFFFFFu && s.used_train && !s.enqueue_error && s.h_nrpc[1] != 0) first_bad = m;
    idx[m++] = j;
  }
  if (first_bad == 0xFFFFFFFFu) return RGB_OK;
  ctx->n_train_recoveries.fetch_add(1, std::memory_order_relaxed);
  /* a block on the wrong XCD: this device does not (always) deal round robin -- persistent trains from now on */
  if (ctx->ring_mem[idx[first_bad]].h_nrpc[1] & RGB_TRAIN_ERR_PLACEMENT) ctx->train_dealt.store(false, std::memory_order_relaxed);
  for (u32 k = m; k-- > first_bad;) {
    rgb_slot &s = ctx->ring_mem[idx[k]];
    if (s.enqueue_error || !s.n) continue;
    if (!s.has_undo) return RGB_E_STATE;                    /* cannot happen: enqueued behind a train in flight */
    int lr = rgb_launch_undo(ctx->dev, s.d_touched, s.n_touched, s.d_undo, 1, ctx->stream);
    if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
  }
  for (u32 k = first_bad; k < m; ++k) {
    rgb_slot &s = ctx->ring_mem[idx[k]];
    if (s.enqueue_error || !s.n) continue;
    if (s.used_train) { s.used_train = false; ctx->trains_in_flight.fetch_sub(1, std::memory_order_release); }
    s.h_nrpc[1] = 0;
    int rc = enqueue_rounds(ctx, s);
    if (rc == RGB_OK) rc = enqueue_results(ctx, s);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   /* (the sequence bytes went back with the rows) */
  return RGB_OK;
}

/* The oldest published batch, taken out of the ring (state 3: the caller's until it is given back).  Under collect_mu
 * only: wait for the batch, size check, take the slot.  cap / rpc_cap: the caller's buffers (rgb_collect), or
 * 0xFFFFFFFF for a caller that reads the slot in place (rgb_collect_view). */
static int take_oldest(rgb_ctx *ctx, bool have_out, uint32_t cap, bool have_rpc_out, uint32_t rpc_cap, uint32_t *n_out,
                       uint32_t *n_rpc_out, rgb_slot **taken, u32 *n_rpc_taken) {
  std::lock_guard<std::mutex> lk(ctx->collect_mu);
  if (ctx->in_flight.load(std::memory_order_acquire) == 0) return RGB_E_EMPTY;
  rgb_slot &s = ctx->ring_mem[ctx->tail];
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  HIPCHK(ctx, hipEventSynchronize(s.done));
  int fail = s.enqueue_error;
  if (!fail && s.used_train && s.h_nrpc[1] != 0) {
    /* the train launch of this batch failed (its error word came back with the results): repair the device
     * state and run this batch and everything enqueued behind it again, one launch per round */
    std::lock_guard<std::mutex> el(ctx->enqueue_mu);
    fail = settle_trains(ctx);
  }
  if (!fail && s.used_train) {
    s.used_train = false;
    ctx->trains_in_flight.fetch_sub(1, std::memory_order_release);
  }
  /* a message reported more records than it has slots (a kind that cannot emit rpcs did): unrecoverable for this
   * batch -- it is consumed all the same, so the ring moves on and the caller sees the error once */
  if (!fail && s.n && s.h_nrpc[2] != 0) fail = RGB_E_STATE;
  if (fail) {
    if (s.used_train) { s.used_train = false; ctx->trains_in_flight.fetch_sub(1, std::memory_order_release); }
    s.enqueue_error = 0;
    ctx->tail = (ctx->tail + 1) % ctx->ring_size;
    ctx->in_flight.fetch_sub(1, std::memory_order_release);
    s.state.store(0, std::memory_order_release);
    return fail;
  }
  const u32 n_rpc = s.n ? s.h_nrpc[0] : 0u;                /* counted on the device */
  /* a buffer that is too small leaves the batch in the ring: the sizes it needs are reported and the
   * caller retries (nothing is dropped, the ring is not wedged) */
  if (s.n > cap || (s.n && !have_out) || (have_rpc_out && n_rpc > rpc_cap)) {
    if (n_out) *n_out = s.n;
    if (n_rpc_out) *n_rpc_out = n_rpc;
    return (s.n > cap || (s.n && !have_out)) ? RGB_E_INVAL : RGB_E_FULL;
  }
  s.state.store(3, std::memory_order_relaxed);              /* mine: the next consumer takes the next slot */
  ctx->tail = (ctx->tail + 1) % ctx->ring_size;
  ctx->in_flight.fetch_sub(1, std::memory_order_release);
  *taken = &s;
  *n_rpc_taken = n_rpc;
  return RGB_OK;
}

/* rgb_collect: the oldest published batch, copied out.  The device wrote both parts into the pinned slot in the order
 * they are handed out (rgb_results_kernel): the decisions in submission order, the rpc records by (msg_index, peer) --
 * two sequential copies, outside the lock, so several consumers copy different batches at once. */
int rgb_collect(rgb_ctx *ctx, rgb_decision *out, uint32_t cap, uint32_t *n_out, rgb_rpc *rpc_out,
                uint32_t rpc_cap, uint32_t *n_rpc_out, uint64_t *tick_out) {
  if (!ctx) return RGB_E_INVAL;
  if (n_out) *n_out = 0;
  if (n_rpc_out) *n_rpc_out = 0;
  rgb_slot *sp = nullptr;
  u32 n_rpc = 0;
  const int rc = take_oldest(ctx, out != nullptr, cap, rpc_out != nullptr, rpc_cap, n_out, n_rpc_out, &sp, &n_rpc);
  if (rc != RGB_OK) return rc;
  rgb_slot &s = *sp;
  if (s.n) copy_out(out, s.h_dec, (size_t)s.n * sizeof(rgb_decision));
  if (n_rpc && rpc_out) memcpy(rpc_out, s.h_rpcs, (size_t)n_rpc * sizeof(rgb_rpc));
  if (n_out) *n_out = s.n;
  if (n_rpc_out) *n_rpc_out = n_rpc;
  if (tick_out) *tick_out = s.tick;
  /* release: the slot's buffers are free for the producer whose turn it is */
  s.state.store(0, std::memory_order_release);
  return RGB_OK;
}

/* rgb_collect_view (ABI v9): the oldest published batch IN PLACE -- pointers into the pinned slot the device wrote, no
 * copy.  The slot is the caller's until rgb_release(view.slot). */
int rgb_collect_view(rgb_ctx *ctx, rgb_view *view) {
  if (!ctx || !view) return RGB_E_INVAL;
  memset(view, 0, sizeof *view);
  rgb_slot *sp = nullptr;
  u32 n_rpc = 0;
  const int rc = take_oldest(ctx, true, 0xFFFFFFFFu, true, 0xFFFFFFFFu, nullptr, nullptr, &sp, &n_rpc);
  if (rc != RGB_OK) return rc;
  view->decisions = sp->h_dec; view->n = sp->n;
  view->rpcs = sp->h_rpcs; view->n_rpcs = n_rpc;
  view->tick = sp->tick;
  view->slot = (uint32_t)(sp - ctx->ring_mem.get());
  return RGB_OK;
}

int rgb_release(rgb_ctx *ctx, uint32_t slot) {
  if (!ctx || slot >= ctx->ring_size) return RGB_E_INVAL;
  int held = 3;
  /* release: the slot's buffers are free for the producer whose turn it is */
  return ctx->ring_mem[slot].state.compare_exchange_strong(held, 0, std::memory_order_release) ? RGB_OK : RGB_E_STATE;
}

/* Sizes of the OLDEST batch in flight (waits for it like rgb_collect, consumes nothing): what a caller needs to
 * allocate exactly the buffers rgb_collect will fill. */
int rgb_peek(rgb_ctx *ctx, uint32_t *n_out, uint32_t *n_rpc_out) {
  if (!ctx) return RGB_E_INVAL;
  if (n_out) *n_out = 0;
  if (n_rpc_out) *n_rpc_out = 0;
  std::lock_guard<std::mutex> lk(ctx->collect_mu);
  if (ctx->in_flight.load(std::memory_order_acquire) == 0) return RGB_E_EMPTY;
  rgb_slot &s = ctx->ring_mem[ctx->tail];
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  HIPCHK(ctx, hipEventSynchronize(s.done));
  if (!s.enqueue_error && s.used_train && s.h_nrpc[1] != 0) {   /* a failed train launch: repaired before it is sized */
    std::lock_guard<std::mutex> el(ctx->enqueue_mu);
    int rc = settle_trains(ctx);
    if (rc) return rc;
  }
  const u32 n_rpc = (s.n && !s.enqueue_error) ? s.h_nrpc[0] : 0u;
  if (n_out) *n_out = s.n;
  if (n_rpc_out) *n_rpc_out = n_rpc;
  return RGB_OK;
}

/* Park until a batch is in flight (RGB_OK), the timeout passes or rgb_wake is called (RGB_E_EMPTY). */
int rgb_wait(rgb_ctx *ctx, uint32_t timeout_ms) {
  if (!ctx) return RGB_E_INVAL;
  std::unique_lock<std::mutex> lk(ctx->wait_mu);
  const u32 gen = ctx->wake_gen.load(std::memory_order_acquire);
  const bool ok = ctx->wait_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] {
    return ctx->in_flight.load(std::memory_order_acquire) != 0 || ctx->wake_gen.load(std::memory_order_acquire) != gen;
  });
  if (!ok) return RGB_E_EMPTY;
  return ctx->in_flight.load(std::memory_order_acquire) != 0 ? RGB_OK : RGB_E_EMPTY;
}

void rgb_wake(rgb_ctx *ctx) {
  if (!ctx) return;
  ctx->wake_gen.fetch_add(1, std::memory_order_release);
  { std::lock_guard<std::mutex> wl(ctx->wait_mu); }
  ctx->wait_cv.notify_all();
}

uint32_t rgb_submit_trains(const rgb_ctx *ctx) { return ctx ? ctx->n_submit_trains.load(std::memory_order_relaxed) : 0; }

uint32_t rgb_in_flight(const rgb_ctx *ctx) { return ctx ? ctx->in_flight.load(std::memory_order_acquire) : 0; }

/* splitmix64 finaliser: the hash of the group partition (ra_amd/shard.py computes the same) */
static inline uint64_t rgb_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

uint32_t rgb_route(uint64_t group_uid, uint32_t n_contexts) {
  return n_contexts <= 1 ? 0u : (uint32_t)(rgb_mix64(group_uid) % n_contexts);
}

int rgb_run_ticks_device(rgb_ctx *ctx, const void *d_msgs, uint32_t tick_stride,
                         const uint32_t *tick_counts, const void *d_tick_counts,
                         const uint32_t *kind_counts, uint32_t n_ticks, void *d_decisions, void *d_rpcs,
                         void *stream) {
  if (!ctx || !d_msgs || !d_decisions) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const rgb_msg *m = (const rgb_msg *)d_msgs;
  rgb_decision *d = (rgb_decision *)d_decisions;
  const u32 *dn = (const u32 *)d_tick_counts;
  const u32 NKIND = RGB_MSG_KIND_MAX + 1;
  for (u32 t = 0; t < n_ticks; ++t) {
    size_t off = (size_t)t * tick_stride;
    if (kind_counts) {
      /* family-ordered tick with host-known per-kind counts: the class-dispatch kernel */
      u32 cc[RGB_N_CLASSES] = {0};
      u64 total = 0;
      for (u32 k = 1; k < NKIND; ++k) { cc[rgb_class_of_kind(k)] += kind_counts[t * NKIND + k]; total += kind_counts[t * NKIND + k]; }
      if (total > tick_stride || kind_counts[t * NKIND + RGB_MSG_NOP]) return RGB_E_INVAL;
      int rc = launch_tick_classes(ctx, ctx->dev, m + off, d + off, (rgb_rpc *)d_rpcs, cc, 0, (u32)off, st);
      if (rc) return rc;
      continue;
    }
    u32 cnt = tick_counts ? tick_counts[t] : tick_stride;
    if (cnt > tick_stride) return RGB_E_INVAL;
    int rc = rgb_launch_tick(ctx->dev, -1, m + off, cnt, dn ? dn + t : nullptr, d + off, (rgb_rpc *)d_rpcs, 0,
                             (u32)off, st);
    if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  }
  return RGB_OK;
}

/* internal: the context's default stream (rgb_wal.hip) */
void *rgb_ctx_stream(rgb_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int rgb_ctx_device(rgb_ctx *ctx) { return ctx ? ctx->cfg.device : 0; }
int rgb_ctx_set_device(rgb_ctx *ctx) {
  if (!ctx) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  return RGB_OK;
}

int rgb_synth_tick_stamped_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs, void *d_kind_counts,
                                  void *d_n, void *d_bucket_counts, void *d_stamps, void *stream) {
  if (!ctx || !d_msgs) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  void *st = stream ? stream : (void *)ctx->stream;
  if (!ctx->d_synth)
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_synth, (size_t)rgb_synth_scratch_words(ctx->dev.n_servers / ctx->dev.n_members) * sizeof(u32)));
  if (d_stamps && !ctx->d_synth_sent) {
    /* first stamped tick: the generator counts on from what the servers hold now */
    const size_t bytes = (size_t)ctx->dev.seq_stride * RGB_TRAIN_SHARDS;
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_synth_sent, bytes));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_synth_sent, ctx->dev.seq, bytes, hipMemcpyDeviceToDevice, (hipStream_t)st));
  }
  int rc = rgb_launch_synth(ctx->dev, seed, tick, (rgb_msg *)d_msgs, ctx->d_synth, (u32 *)d_kind_counts,
                            (u32 *)d_n, (u32 *)d_bucket_counts, (unsigned char *)d_stamps,
                            d_stamps ? ctx->d_synth_sent : nullptr, st);
  if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

int rgb_synth_snapshot_mark_device(rgb_ctx *ctx, void *d_snap_stamps, void *stream) {
  if (!ctx) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t bytes = (size_t)ctx->dev.seq_stride * RGB_TRAIN_SHARDS;
  if (!ctx->d_synth_sent) {
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_synth_sent, bytes));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_synth_sent, ctx->dev.seq, bytes, hipMemcpyDeviceToDevice, st));
  }
  int lr = rgb_launch_seq_bump(ctx->d_synth_sent, (unsigned char *)d_snap_stamps, (u32)bytes, (void *)st);
  if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

int rgb_synth_stamps_resync_device(rgb_ctx *ctx, void *stream) {
  if (!ctx) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t bytes = (size_t)ctx->dev.seq_stride * RGB_TRAIN_SHARDS;
  if (!ctx->d_synth_sent) HIPCHK(ctx, hipMalloc((void **)&ctx->d_synth_sent, bytes));
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_synth_sent, ctx->dev.seq, bytes, hipMemcpyDeviceToDevice, st));
  return RGB_OK;
}

int rgb_synth_set_hint(rgb_ctx *ctx, uint32_t level) {
  if (!ctx || level > 2u) return RGB_E_INVAL;
  ctx->synth_hint = level;
  ctx->dev.synth_hint = level;       /* (by value in every later launch of the generator) */
  return RGB_OK;
}

int rgb_synth_tick_buckets_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs, void *d_kind_counts,
                                  void *d_n, void *d_bucket_counts, void *stream) {
  return rgb_synth_tick_stamped_device(ctx, seed, tick, d_msgs, d_kind_counts, d_n, d_bucket_counts, nullptr, stream);
}

int rgb_synth_tick_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs, void *d_kind_counts,
                          void *d_n, void *stream) {
  return rgb_synth_tick_stamped_device(ctx, seed, tick, d_msgs, d_kind_counts, d_n, nullptr, nullptr, stream);
}

/* ---- train launches (include/ra_gpu_batch.h) ---- */
uint32_t rgb_train_bucket(uint32_t kind, uint32_t flags, uint32_t server, uint32_t n_members) {
  return n_members ? rgb_bucket(kind, flags, server, n_members) : 0u;
}

static int train_scratch(rgb_ctx *ctx) {
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  if (!ctx->d_train_ctl) {
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_train_ctl, RGB_TRAIN_CTL_WORDS * sizeof(u32)));
    /* (on the context's non-blocking stream: a null-stream memset is not ordered with the calibration launch below and
     * wiped some of its marks -- found by the fail-safe test running behind other tests of one process) */
    HIPCHK(ctx, hipMemsetAsync(ctx->d_train_ctl, 0, RGB_TRAIN_CTL_WORDS * sizeof(u32), ctx->stream));
  }
  if (!ctx->d_seq_cnt) HIPCHK(ctx, hipMalloc((void **)&ctx->d_seq_cnt, (size_t)ctx->dev.seq_stride * RGB_TRAIN_SHARDS));
  if (ctx->xcc_state == 0) {
    /* which XCCs does the device have?  A train block serves the shard(s) of the XCC it runs on (placement by
     * construction); the ids must be 0 .. n-1 with n dividing the 8 shards */
    HIPCHK(ctx, hipMemsetAsync(ctx->d_train_ctl + 1, 0, 2 * sizeof(u32), ctx->stream));
    {
      int rc = rgb_launch_train_calibrate(ctx->d_train_ctl + 1, ctx->stream);
      if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
    }
    u32 cal[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(cal, ctx->d_train_ctl + 1, sizeof cal, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_train_ctl + 1, 0, 2 * sizeof(u32), ctx->stream));
    const u32 w = cal[0];
    u32 n = 0;
    while ((w >> n) & 1u) ++n;
    ctx->train_blocks = rgb_train_resident_blocks(ctx->dev.n_members);
    const bool ok = n >= 1 && n <= RGB_TRAIN_SHARDS && (n & (n - 1u)) == 0 && (w >> n) == 0 &&
                    ctx->train_blocks >= RGB_TRAIN_SHARDS;
    ctx->n_xcc = ok ? n : 0;
    ctx->xcc_state = ok ? 1 : -1;
    if (getenv("RGB_TRACE_TRAIN_SETUP"))
      fprintf(stderr, "[rgb] train setup: xcc mask %#x rotations %#x resident blocks %u -> %s\n", cal[0], cal[1],
              ctx->train_blocks, ok ? "ok" : "unsupported");
    /* the dealt form only where the calibration launch was dealt round robin over eight XCCs (one rotation seen) */
    ctx->train_dealt.store(ok && n == RGB_TRAIN_SHARDS && cal[1] != 0 && (cal[1] & (cal[1] - 1u)) == 0 &&
                           !(ctx->cfg.flags & RGB_CFG_TRAIN_PERSISTENT), std::memory_order_relaxed);
  }
  return ctx->xcc_state == 1 ? RGB_OK : RGB_E_UNSUPPORTED;
}

int rgb_train_plan_create(rgb_ctx *ctx, const uint32_t *bucket_counts, uint32_t n_ticks, rgb_train_plan **out) {
  return rgb_train_plan_create_snap(ctx, bucket_counts, n_ticks, 0, out);
}

int rgb_train_plan_create_snap(rgb_ctx *ctx, const uint32_t *bucket_counts, uint32_t n_ticks, uint32_t snapshot_every,
                               rgb_train_plan **out) {
  if (!ctx || !out || (!bucket_counts && n_ticks)) return RGB_E_INVAL;
  *out = nullptr;
  if (!ctx->registered) return RGB_E_STATE;
  int rc;
  {
    std::lock_guard<std::mutex> tl(ctx->train_mu);          /* the one-off calibration uses the stream */
    rc = train_scratch(ctx);
  }
  if (rc) return rc;
  rgb_train_plan *p = new (std::nothrow) rgb_train_plan();
  if (!p) return RGB_E_NOMEM;
  std::vector<rgb_train_tick> ticks(n_ticks);
  u32 rows = 0;
  const u32 snap_rows = rgb_train_snap_rows(ctx->dev.n_servers / ctx->dev.n_members);
  auto snap_of = [&](u32 t) -> u32 { return (snapshot_every && t && t % snapshot_every == 0) ? snap_rows : 0u; };
  for (u32 t = 0; t < n_ticks; ++t) {
    const u32 r = rgb_train_make_tick(bucket_counts + (size_t)t * RGB_N_BUCKETS, ctx->dev.n_members, &ticks[t], nullptr, 0, snap_of(t));
    if (r > rows) rows = r;
  }
  std::vector<u32> tab((size_t)n_ticks * rows, 0xFFFFFFFFu);
  for (u32 t = 0; t < n_ticks; ++t) {
    rgb_train_make_tick(bucket_counts + (size_t)t * RGB_N_BUCKETS, ctx->dev.n_members, &ticks[t], tab.data() + (size_t)t * rows, rows, snap_of(t));
    ticks[t].snap = snap_of(t) ? t / snapshot_every : 0u;        /* 1 + ordinal: the snapshot in front of tick k x every is k - 1 */
  }
  p->snap_every = snapshot_every;
  p->n_ticks = n_ticks;
  p->bpt = rows * RGB_TRAIN_SHARDS;
  if (n_ticks && rows) {
    hipError_t e = hipMalloc((void **)&p->d_ticks, (size_t)n_ticks * sizeof(rgb_train_tick));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_rows, tab.size() * sizeof(u32));
    if (e == hipSuccess)
      e = hipMemcpy(p->d_ticks, ticks.data(), (size_t)n_ticks * sizeof(rgb_train_tick), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_rows, tab.data(), tab.size() * sizeof(u32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      ctx->last_hip.store((int)e, std::memory_order_relaxed);
      if (p->d_ticks) (void)hipFree(p->d_ticks);
      if (p->d_rows) (void)hipFree(p->d_rows);
      delete p;
      return RGB_E_HIP;
    }
  }
  *out = p;
  return RGB_OK;
}

int rgb_train_plan_create_device(rgb_ctx *ctx, uint32_t n_ticks, uint32_t snapshot_every, rgb_train_plan **out) {
  if (!ctx || !out) return RGB_E_INVAL;
  *out = nullptr;
  if (!ctx->registered) return RGB_E_STATE;
  int rc;
  {
    std::lock_guard<std::mutex> tl(ctx->train_mu);          /* the one-off calibration uses the stream */
    rc = train_scratch(ctx);
  }
  if (rc) return rc;
  rgb_train_plan *p = new (std::nothrow) rgb_train_plan();
  if (!p) return RGB_E_NOMEM;
  const u32 rows = rgb_train_rows_bound(ctx->dev.n_servers, ctx->dev.n_members, snapshot_every != 0);
  p->snap_every = snapshot_every;
  p->n_ticks = n_ticks;
  p->bpt = rows * RGB_TRAIN_SHARDS;
  p->on_device = true;
  try { p->rows_fit.assign(n_ticks, 0xFFFFFFFFu); } catch (...) { delete p; return RGB_E_NOMEM; }
  if (n_ticks) {
    hipError_t e = hipMalloc((void **)&p->d_ticks, (size_t)n_ticks * sizeof(rgb_train_tick));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_rows, (size_t)n_ticks * rows * sizeof(u32));
    /* an unbuilt tick has no rows (a launch over it does nothing) */
    if (e == hipSuccess) e = hipMemset(p->d_ticks, 0, (size_t)n_ticks * sizeof(rgb_train_tick));
    if (e == hipSuccess) e = hipMemset(p->d_rows, 0xFF, (size_t)n_ticks * rows * sizeof(u32));
    if (e != hipSuccess) {
      ctx->last_hip.store((int)e, std::memory_order_relaxed);
      if (p->d_ticks) (void)hipFree(p->d_ticks);
      if (p->d_rows) (void)hipFree(p->d_rows);
      delete p;
      return RGB_E_HIP;
    }
  }
  *out = p;
  return RGB_OK;
}

int rgb_train_plan_build_device(rgb_ctx *ctx, rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                                const void *d_bucket_counts, void *stream) {
  if (!ctx || !plan || (!d_bucket_counts && n_ticks)) return RGB_E_INVAL;
  if (!ctx->registered || ctx->xcc_state != 1) return RGB_E_STATE;
  if (!plan->on_device || (uint64_t)first_tick + n_ticks > plan->n_ticks) return RGB_E_INVAL;
  void *st = stream ? stream : (void *)ctx->stream;
  int lr = rgb_launch_train_plan((const u32 *)d_bucket_counts, plan->d_ticks, plan->d_rows, plan->bpt / RGB_TRAIN_SHARDS,
                                 first_tick, n_ticks, plan->snap_every, ctx->dev.n_servers / ctx->dev.n_members,
                                 ctx->dev.n_members, ctx->d_train_ctl, st);
  if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
  for (u32 t = first_tick; t < first_tick + n_ticks; ++t) plan->rows_fit[t] = 0xFFFFFFFFu;      /* rebuilt: not known any more */
  return RGB_OK;
}

/* Optional, outside any timed path: tell the HOST how many rows the built ticks [first_tick, first_tick + n_ticks) have --
 * four bytes per tick come back, nothing else of the plan -- so that launches over them take a grid of their rows instead
 * of the rows bound (a dealt launch of a device-built plan otherwise starts ~2 x as many blocks as it has rows: measured
 * +2.5 % per tick in long launches, +4..6 % in a 20-tick launch).  Synchronises `stream`. */
int rgb_train_plan_fit(rgb_ctx *ctx, rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks, void *stream) {
  if (!ctx || !plan) return RGB_E_INVAL;
  if (!plan->on_device || (uint64_t)first_tick + n_ticks > plan->n_ticks) return RGB_E_INVAL;
  if (n_ticks == 0) return RGB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  std::vector<u32> rows(n_ticks);
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipMemcpy2D(rows.data(), sizeof(u32), &plan->d_ticks[first_tick].n_rows, sizeof(rgb_train_tick), sizeof(u32),
                          n_ticks, hipMemcpyDeviceToHost));
  for (u32 t = 0; t < n_ticks; ++t) plan->rows_fit[first_tick + t] = rows[t];
  return RGB_OK;
}

int rgb_train_plan_download(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t tick, void *out_tick, uint32_t *out_rows,
                            uint32_t rows_cap) {
  if (!ctx || !plan || !out_tick || tick >= plan->n_ticks) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      /* (a build on another stream: the caller has synchronised it) */
  rgb_train_tick h;
  HIPCHK(ctx, hipMemcpy(&h, plan->d_ticks + tick, sizeof h, hipMemcpyDeviceToHost));
  memcpy(out_tick, &h, sizeof h);
  const u32 rpt = plan->bpt / RGB_TRAIN_SHARDS;
  const u32 n = h.n_rows < rows_cap ? h.n_rows : rows_cap;
  if (out_rows && n && h.n_rows <= rpt)
    HIPCHK(ctx, hipMemcpy(out_rows, plan->d_rows + (size_t)tick * rpt, (size_t)n * sizeof(u32), hipMemcpyDeviceToHost));
  return (int)h.n_rows;
}

void rgb_train_plan_destroy(rgb_train_plan *plan) {
  if (!plan) return;
  if (plan->d_ticks) (void)hipFree(plan->d_ticks);
  if (plan->d_rows) (void)hipFree(plan->d_rows);
  delete plan;
}

uint32_t rgb_train_plan_blocks_per_tick(const rgb_train_plan *plan) {
  if (!plan) return 0;
  if (plan->on_device && !plan->rows_fit.empty()) {
    /* a device-built plan whose ticks the host has all been told (rgb_train_plan_fit): its longest tick, not the bound */
    u32 mx = 0;
    for (u32 r : plan->rows_fit) { if (r == 0xFFFFFFFFu) return plan->bpt; if (r > mx) mx = r; }
    if (mx * RGB_TRAIN_SHARDS <= plan->bpt) return (mx ? mx : 1u) * RGB_TRAIN_SHARDS;
  }
  return plan->bpt;
}

int rgb_train_stamp_device(rgb_ctx *ctx, const void *d_msgs, void *d_stamps, uint32_t tick_stride,
                           const uint32_t *tick_counts, uint32_t n_ticks, void *stream) {
  if (!ctx || !d_msgs || !d_stamps || (!tick_counts && n_ticks)) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  if (!ctx->d_seq_cnt || ctx->xcc_state != 1) return RGB_E_STATE;      /* rgb_train_plan_create comes first */
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const rgb_msg *m = (const rgb_msg *)d_msgs;
  unsigned char *sp = (unsigned char *)d_stamps;
  /* the counters start from what the servers hold now (stream order: after every train enqueued before) */
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_seq_cnt, ctx->dev.seq, (size_t)ctx->dev.seq_stride * RGB_TRAIN_SHARDS,
                             hipMemcpyDeviceToDevice, st));
  for (u32 t = 0; t < n_ticks; ++t) {
    if (tick_counts[t] > tick_stride) return RGB_E_INVAL;
    int rc = rgb_launch_train_seq(ctx->dev, m + (size_t)t * tick_stride, tick_counts[t], ctx->d_seq_cnt,
                                  sp + (size_t)t * tick_stride, st);
    if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  }
  return RGB_OK;
}

int rgb_train_run_device(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                         const void *d_msgs, const void *d_stamps, uint32_t tick_stride, void *d_decisions,
                         void *d_rpcs, uint32_t rpc_ring, void *stream) {
  return rgb_train_run_snap_device(ctx, plan, first_tick, n_ticks, d_msgs, d_stamps, tick_stride, d_decisions, d_rpcs,
                                   rpc_ring, nullptr, nullptr, stream);
}

uint32_t rgb_train_seq_bytes(const rgb_ctx *ctx) {
  return (ctx && ctx->registered) ? ctx->dev.seq_stride * RGB_TRAIN_SHARDS : 0u;
}

int rgb_snapshot_train_device(rgb_ctx *ctx, void *d_rows, void *stream) {
  int rc = rgb_snapshot_device(ctx, d_rows, stream);
  if (rc) return rc;
  void *st = stream ? stream : (void *)ctx->stream;
  int lr = rgb_launch_seq_bump(ctx->dev.seq, nullptr, ctx->dev.seq_stride * RGB_TRAIN_SHARDS, st);
  if (lr) { ctx->last_hip.store(lr, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

int rgb_train_run_snap_device(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                              const void *d_msgs, const void *d_stamps, uint32_t tick_stride, void *d_decisions,
                              void *d_rpcs, uint32_t rpc_ring, const void *d_snap_stamps, void *d_snap_rows,
                              void *stream) {
  if (!ctx || !plan || !d_msgs || !d_stamps || !d_decisions) return RGB_E_INVAL;
  /* a plan with snapshots advances the sequence bytes at its boundaries: it cannot run without them */
  if (plan->snap_every && (!d_snap_stamps || !d_snap_rows)) return RGB_E_INVAL;
  if (!plan->snap_every && (d_snap_stamps || d_snap_rows)) return RGB_E_INVAL;
  if (!ctx->registered || ctx->xcc_state != 1) return RGB_E_STATE;
  if ((uint64_t)first_tick + n_ticks > plan->n_ticks) return RGB_E_INVAL;
  if (n_ticks == 0 || plan->bpt == 0) return RGB_OK;
  void *st = stream ? stream : (void *)ctx->stream;
  /* launches of at most RGB_TRAIN_MAX_TICKS ticks (and of a grid the runtime accepts) */
  u32 per = RGB_TRAIN_MAX_TICKS;
  if ((uint64_t)per * plan->bpt > 0x7FFFFFFFull) per = (u32)(0x7FFFFFFFull / plan->bpt);
  if (per == 0) return RGB_E_INVAL;
  /* (a snapshot in front of a launch's first tick is outside the launch: the caller splits at tick counts that keep
   * every boundary inside -- one launch of at most 255 ticks -- or takes that snapshot itself) */
  if (plan->snap_every) {
    if (n_ticks > per) return RGB_E_INVAL;
    /* ticks + the snapshots inside the launch: a server's sequence byte must not come round within it */
    const u32 last = first_tick + n_ticks - 1u;
    const u32 inside = last / plan->snap_every - first_tick / plan->snap_every;
    if (n_ticks + inside > RGB_TRAIN_MAX_TICKS) return RGB_E_INVAL;
  }
  for (u32 t = first_tick; t < first_tick + n_ticks; t += per) {
    const u32 n = first_tick + n_ticks - t < per ? first_tick + n_ticks - t : per;
    const size_t off = (size_t)t * tick_stride;
    /* the grid's rows per tick: the table's, or -- a device-built plan whose ticks the host has been told
     * (rgb_train_plan_fit) -- the rows of the launch's longest tick */
    u32 grid_bpt = plan->bpt;
    if (plan->on_device) {
      u32 mx = 0; bool known = true;
      for (u32 k = t; k < t + n && known; ++k) { known = plan->rows_fit[k] != 0xFFFFFFFFu; if (known && plan->rows_fit[k] > mx) mx = plan->rows_fit[k]; }
      if (known && mx * RGB_TRAIN_SHARDS <= plan->bpt) grid_bpt = (mx ? mx : 1u) * RGB_TRAIN_SHARDS;
    }
    int rc = rgb_launch_train(ctx->dev, (const rgb_msg *)d_msgs + off, (const unsigned char *)d_stamps + off,
                              tick_stride, plan->d_ticks + t, plan->d_rows + (size_t)t * (plan->bpt / RGB_TRAIN_SHARDS), n,
                              grid_bpt, (rgb_decision *)d_decisions + off,
                              (rgb_rpc *)d_rpcs, rpc_ring, (u32)off, ctx->d_train_ctl, ctx->n_xcc,
                              /* (a device-built plan runs in the dealt form too: its grid is the rows BOUND of a tick --
                               * rgb_train_rows_bound -- the blocks behind a tick's real rows find an empty table entry and exit) */
                              ctx->train_dealt.load(std::memory_order_relaxed) ? 0u : ctx->train_blocks, st,
                              (const unsigned char *)d_snap_stamps, (rgb_leaderboard_row *)d_snap_rows,
                              plan->bpt / RGB_TRAIN_SHARDS);
    if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  }
  return RGB_OK;
}

int rgb_train_status(rgb_ctx *ctx, uint32_t *flags_out, uint32_t *xcc_of_shard) {
  if (!ctx) return RGB_E_INVAL;
  if (flags_out) *flags_out = 0;
  if (xcc_of_shard)
    for (u32 x = 0; x < RGB_TRAIN_SHARDS; ++x)
      xcc_of_shard[x] = ctx->xcc_state == 1 ? x % ctx->n_xcc : 0xFFFFFFFFu;
  if (!ctx->d_train_ctl) return RGB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  /* (every launch verifies its own placement marks behind itself: the error word is final once the stream is idle) */
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  u32 w = 0;
  HIPCHK(ctx, hipMemcpy(&w, ctx->d_train_ctl, sizeof w, hipMemcpyDeviceToHost));
  if (flags_out) *flags_out = w;
  if (w & RGB_TRAIN_ERR_PLACEMENT) ctx->train_dealt.store(false, std::memory_order_relaxed);
  if (w) {
    HIPCHK(ctx, hipMemsetAsync(ctx->d_train_ctl, 0, sizeof(u32), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RGB_E_STATE;
  }
  return RGB_OK;
}

int rgb_synth_apply_tick_device(rgb_ctx *ctx, const void *d_msgs, uint32_t max_msgs, void *d_decisions,
                                void *d_rpcs, void *stream) {
  if (!ctx || !d_msgs || !d_decisions) return RGB_E_INVAL;
  if (!ctx->registered || !ctx->d_synth) return RGB_E_STATE;
  void *st = stream ? stream : (void *)ctx->stream;
  int rc = rgb_launch_tick_classes(ctx->dev, (const rgb_msg *)d_msgs, nullptr, ctx->d_synth, max_msgs,
                                   (rgb_decision *)d_decisions, (rgb_rpc *)d_rpcs, 0, 0, st);
  if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

int rgb_snapshot_device(rgb_ctx *ctx, void *d_rows, void *stream) {
  if (!ctx || !d_rows) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  void *st = stream ? stream : (void *)ctx->stream;
  int rc = rgb_launch_leaderboard(ctx->dev, (rgb_leaderboard_row *)d_rows, st);
  if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  return RGB_OK;
}

int rgb_snapshot(rgb_ctx *ctx, rgb_leaderboard_row *out) {
  if (!ctx || !out) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  rgb_stream_turn turn(ctx);                               /* d_rows is shared */
  if (turn.rc) return turn.rc;
  int rc = rgb_snapshot_device(ctx, ctx->d_rows, ctx->stream);
  if (rc) return rc;
  u32 g = ctx->dev.n_servers / ctx->dev.n_members;
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_rows, (size_t)g * sizeof(rgb_leaderboard_row), hipMemcpyDeviceToHost,
                             ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return RGB_OK;
}

/* host-buffer form of the leaderboard all-gather (what the NIF hands out as a binary): this context's rows are
 * produced on the device, padded to n_rows, gathered and copied to rows_all (n_ranks * n_rows rows).
 *
 * The decision path has no exchange step and this call must not give it one (round 5): the locks of rgb_submit /
 * rgb_collect (rgb_stream_turn) are held only while the snapshot is ENQUEUED on the context's stream; the collective,
 * its copy-out and the wait run on a side stream behind an event, so a slow or missing rank delays this caller, never a
 * producer, a consumer or the collector thread of this context.  Every exit is collective-safe: whatever goes wrong
 * locally (n_rows smaller than this rank's groups, an allocation, the snapshot) is carried into an 8-byte STATUS
 * all-gather that every rank takes part in first; unless every rank reports RGB_OK nobody starts the payload gather and
 * all ranks return an error (the first failing rank's).  A collective that does not complete within
 * RGB_COMM_TIMEOUT_MS (environment, default 30 000) is abandoned with ncclCommAbort: RGB_E_COMM,
 * rgb_comm_last_error() says so, the communicator has to be re-created. */
extern "C" int rgb_comm_allgather_bytes(rgb_comm *comm, const void *d_local, uint64_t bytes, void *d_all, void *stream);
extern "C" int rgb_comm_abort(rgb_comm *comm, const char *why);
extern "C" void rgb_comm_set_error_text(const char *why);

/* the side stream has drained, or the deadline has passed */
static bool lb_wait(rgb_ctx *ctx, unsigned timeout_ms) {
#ifdef RGB_HOST_EMULATION
  /* (the emulated stream is always idle; RGB_EMU_LB_TIMEOUT -- tests/test_c_abi_on_cpu.py -- plays a rank that never
   * arrives, so that the give-up path of rgb_leaderboard_allgather_host runs on a CPU) */
  (void)ctx; (void)timeout_ms;
  return getenv("RGB_EMU_LB_TIMEOUT") == nullptr;
#else
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(ctx->lb_stream);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { ctx->last_hip.store((int)q, std::memory_order_relaxed); return false; }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
#endif
}

int rgb_leaderboard_allgather_host(rgb_ctx *ctx, rgb_comm *comm, uint32_t n_rows, rgb_leaderboard_row *rows_all) {
  if (!ctx || !comm || !rows_all) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  const u32 g = ctx->dev.n_servers / ctx->dev.n_members, world = rgb_comm_n_ranks(comm);
  if (world == 0) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  unsigned timeout_ms = 30000;
  if (const char *e = getenv("RGB_COMM_TIMEOUT_MS")) { const long v = atol(e); if (v > 0) timeout_ms = (unsigned)v; }
  std::lock_guard<std::mutex> one(ctx->lb_mu);
  /* everything that can fail locally happens in front of the collectives and only sets `mine` */
  int64_t mine = RGB_OK;
  if (n_rows < g) mine = RGB_E_INVAL;
  if (!ctx->lb_stream && hipStreamCreateWithFlags(&ctx->lb_stream, hipStreamNonBlocking) != hipSuccess) return RGB_E_HIP;
  if (!ctx->lb_event && hipEventCreateWithFlags(&ctx->lb_event, hipEventDisableTiming) != hipSuccess) return RGB_E_HIP;
  const size_t rows_bytes = (size_t)n_rows * (world + 1u) * sizeof(rgb_leaderboard_row);
  const size_t need = rows_bytes + (size_t)(world + 1u) * sizeof(int64_t);
  if (ctx->lb_gather_bytes < need) {
    (void)hipStreamSynchronize(ctx->lb_stream);
    if (ctx->d_lb_gather) (void)hipFree(ctx->d_lb_gather);
    ctx->d_lb_gather = nullptr; ctx->lb_gather_bytes = 0;
    /* (room for the status words even when the rows cannot be had: the status exchange must still happen) */
    if (hipMalloc(&ctx->d_lb_gather, need) == hipSuccess) ctx->lb_gather_bytes = need;
    else if (hipMalloc(&ctx->d_lb_gather, (size_t)(world + 1u) * sizeof(int64_t)) == hipSuccess) {
      ctx->lb_gather_bytes = (size_t)(world + 1u) * sizeof(int64_t);
      if (mine == RGB_OK) mine = RGB_E_NOMEM;
    } else return RGB_E_NOMEM;               /* (not even 8 bytes per rank: this device is gone) */
  }
  const bool have_rows = ctx->lb_gather_bytes >= need;
  char *base = (char *)ctx->d_lb_gather;
  rgb_leaderboard_row *d_local = (rgb_leaderboard_row *)base, *d_all = d_local + n_rows;
  int64_t *d_status = (int64_t *)(base + (have_rows ? rows_bytes : 0)), *d_status_all = d_status + 1;
  if (mine == RGB_OK) {
    /* the decision path's locks: only around the ENQUEUE of the snapshot on the context's stream */
    rgb_stream_turn turn(ctx);
    if (turn.rc) mine = turn.rc;
    else if (hipMemsetAsync(d_local, 0, (size_t)n_rows * sizeof(rgb_leaderboard_row), ctx->stream) != hipSuccess) mine = RGB_E_HIP;
    else {
      const int rc = rgb_snapshot_device(ctx, d_local, ctx->stream);
      if (rc) mine = rc;
      else if (hipEventRecord(ctx->lb_event, ctx->stream) != hipSuccess) mine = RGB_E_HIP;
    }
  }
  if (mine == RGB_OK && hipStreamWaitEvent(ctx->lb_stream, ctx->lb_event, 0) != hipSuccess) mine = RGB_E_HIP;
  /* the copy-outs go through PINNED memory owned by the context (a pageable destination makes hipMemcpyAsync either
   * synchronous -- a missing rank would block inside the copy, in front of the timeout -- or leaves a copy queued
   * behind the aborted collective that later writes into memory the caller has got back) */
  const size_t pin_need = (size_t)(world + 1u) * sizeof(int64_t) + (size_t)n_rows * world * sizeof(rgb_leaderboard_row);
  if (ctx->lb_pinned_bytes < pin_need) {
    (void)hipStreamSynchronize(ctx->lb_stream);
    if (ctx->h_lb_pinned) (void)hipHostFree(ctx->h_lb_pinned);
    ctx->h_lb_pinned = nullptr; ctx->lb_pinned_bytes = 0;
    if (hipHostMalloc(&ctx->h_lb_pinned, pin_need, hipHostMallocDefault) == hipSuccess) ctx->lb_pinned_bytes = pin_need;
    else if (hipHostMalloc(&ctx->h_lb_pinned, (size_t)(world + 1u) * sizeof(int64_t), hipHostMallocDefault) == hipSuccess) {
      ctx->lb_pinned_bytes = (size_t)(world + 1u) * sizeof(int64_t);
      if (mine == RGB_OK) mine = RGB_E_NOMEM;
    } else return RGB_E_NOMEM;
  }
  int64_t *h_mine = (int64_t *)ctx->h_lb_pinned, *h_status = h_mine + 1;
  rgb_leaderboard_row *h_rows = (rgb_leaderboard_row *)(h_status + world);
  /* after a timeout: the communicator is aborted (its kernels leave the stream), then the side stream is DRAINED, so
   * that nothing of this call is still queued when the caller gets its buffers back */
  auto give_up = [&](const char *why) {
    const int rc = rgb_comm_abort(comm, why);
    (void)hipStreamSynchronize(ctx->lb_stream);
    return rc;
  };
  /* 1. the status of every rank.  (A HIP call that fails HERE -- between the collectives -- means this device is gone;
   * the other ranks then leave through their timeout, which is what it is for.) */
  *h_mine = mine;
  HIPCHK(ctx, hipMemcpyAsync(d_status, h_mine, sizeof mine, hipMemcpyHostToDevice, ctx->lb_stream));
  int rc = rgb_comm_allgather_bytes(comm, d_status, sizeof(int64_t), d_status_all, (void *)ctx->lb_stream);
  if (rc) { (void)hipStreamSynchronize(ctx->lb_stream); return rc; }
  if (hipMemcpyAsync(h_status, d_status_all, (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->lb_stream) != hipSuccess)
    return give_up("leaderboard all-gather: the status copy-out failed (communicator aborted)");
  if (!lb_wait(ctx, timeout_ms)) return give_up("leaderboard all-gather: a rank did not arrive (status exchange timed out; communicator aborted)");
  for (u32 r = 0; r < world; ++r)
    if (h_status[r] != RGB_OK) {
      if (mine == RGB_OK) rgb_comm_set_error_text("leaderboard all-gather: another rank reported an error; nothing was gathered");
      return mine != RGB_OK ? (int)mine : RGB_E_COMM;
    }
  /* 2. the rows */
  rc = rgb_comm_allgather_bytes(comm, d_local, (uint64_t)n_rows * sizeof(rgb_leaderboard_row), d_all, (void *)ctx->lb_stream);
  if (rc) { (void)hipStreamSynchronize(ctx->lb_stream); return rc; }
  if (hipMemcpyAsync(h_rows, d_all, (size_t)n_rows * world * sizeof(rgb_leaderboard_row), hipMemcpyDeviceToHost, ctx->lb_stream) != hipSuccess)
    return give_up("leaderboard all-gather: the copy-out failed (communicator aborted)");
  if (!lb_wait(ctx, timeout_ms)) return give_up("leaderboard all-gather: timed out (communicator aborted)");
  memcpy(rows_all, h_rows, (size_t)n_rows * world * sizeof(rgb_leaderboard_row));     /* only now: the side stream is idle */
  return RGB_OK;
}

int rgb_state_checksum(rgb_ctx *ctx, uint32_t first, uint32_t n, uint64_t *out) {
  if (!ctx || !out) return RGB_E_INVAL;
  if (!ctx->registered) return RGB_E_STATE;
  if ((uint64_t)first + n > ctx->dev.n_servers) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  rgb_stream_turn turn(ctx);                               /* d_sums is shared */
  if (turn.rc) return turn.rc;
  int rc = rgb_launch_checksum(ctx->dev, first, n, ctx->d_sums, ctx->stream);
  if (rc) { ctx->last_hip.store(rc, std::memory_order_relaxed); return RGB_E_HIP; }
  std::vector<u64> sums(n);
  HIPCHK(ctx, hipMemcpyAsync(sums.data(), ctx->d_sums, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost,
                             ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  /* checksum of checksums: position-weighted sum mod 2^64 */
  u64 acc = 0;
  for (u32 k = 0; k < n; ++k) acc += sums[k] * (2ull * (u64)(first + k) + 1ull);
  *out = acc;
  return RGB_OK;
}

/* profiling aid (the -DRGB_PROFILE build with RGB_DEBUG & 16): per-wave timestamps of the last class-dispatch
 * launch; RGB_E_INVAL in the product library */
int rgb_debug_read(rgb_ctx *ctx, uint64_t *out, uint32_t n_words) {
  if (!ctx || !out || !ctx->dev.dbg_buf) return RGB_E_INVAL;
  HIPCHK(ctx, hipMemcpy(out, ctx->dev.dbg_buf, (size_t)n_words * sizeof(u64), hipMemcpyDeviceToHost));
  return RGB_OK;
}

/* fail-safe tests and diagnostics (not part of the boundary): the next train batch of rgb_submit gets a fault
 * (1 = a stamp that never comes up, 2 = two messages bucketed under each other's shard); the batches whose failed
 * train launch the engine repaired so far */
void rgb_debug_inject_train_fault(rgb_ctx *ctx, uint32_t fault) {
  if (ctx) ctx->inject_fault.store(fault, std::memory_order_relaxed);
}
uint32_t rgb_train_form(const rgb_ctx *ctx) {
  if (!ctx || ctx->xcc_state != 1) return RGB_TRAIN_FORM_NONE;
  return ctx->train_dealt.load(std::memory_order_relaxed) ? RGB_TRAIN_FORM_DEALT : RGB_TRAIN_FORM_PERSISTENT;
}
uint32_t rgb_train_recoveries(const rgb_ctx *ctx) { return ctx ? ctx->n_train_recoveries.load(std::memory_order_relaxed) : 0; }

int rgb_synchronize(rgb_ctx *ctx) {
  if (!ctx) return RGB_E_INVAL;
  HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return RGB_OK;
}

}  /* extern "C" */
