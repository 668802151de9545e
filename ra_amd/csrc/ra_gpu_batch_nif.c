/*
 * ra_gpu_batch_nif.c -- Erlang NIF shim over the C ABI of include/ra_gpu_batch.h.
 *
 * No logic lives here: every function unpacks its arguments, calls exactly one rgb_* entry
 * point and packs the result.  Records cross the boundary as binaries with the exact C layout
 * (erlang/ra_gpu_batch.erl builds and parses them), so the hot calls are one memcpy each.
 * Threading follows SURVEY.md section 8b and the reference's interception point, which is per gen_statem
 * (src/ra_server_proc.erl:1356-1397):
 *   - submit/3 may be called by any number of processes at once (rgb_submit serialises them); batches above
 *     SUBMIT_DIRTY_MSGS messages reschedule themselves onto a dirty CPU scheduler (enif_schedule_nif) so that the
 *     validation + family sort + memcpy never holds a normal scheduler;
 *   - results come back either through the dirty NIF collect/1 or through the collector thread started by
 *     start_collector/2.  The thread parks in rgb_wait (condition variable, no polling), and FANS EVERY BATCH BACK
 *     PER OWNER: register_owner(Ctx, FirstServer, N, Pid) says which gen_statem owns servers
 *     [FirstServer, FirstServer+N); each owner that has decisions in a batch receives ONE message
 *     {ra_gpu_batch, Tick, NDecisions, DecisionsBin, RpcsBin} holding only its decisions (submission order) and
 *     their rpc records (msg_index = index into that DecisionsBin); servers nobody registered go to the default
 *     owner given to start_collector/2.  collect/1 is refused ({error, collector_running}) while the thread runs:
 *     one consumer at a time is a contract of the shim, the C ABI itself tolerates several.
 * The resource destructor stops the thread and calls rgb_close.
 *
 * Cannot be built against the real erl_nif.h in this image (no OTP): `make nif-check` syntax-checks it
 * against ra_amd/csrc/nif_stub/erl_nif.h, and tests/test_nif_shim_mock_beam.py EXECUTES it against a functional
 * mock of that API (tests/native/mock_beam) on top of the CPU-emulated library (also under ThreadSanitizer).
 * Build line for a machine with OTP >= 26:
 *   cc -O2 -fPIC -shared -I$ERL_INCLUDE -Iinclude ra_amd/csrc/ra_gpu_batch_nif.c \
 *      -Lra_amd/csrc -lra_gpu_batch -o priv/ra_gpu_batch_nif.so
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c11 */
#endif
#include <stdatomic.h>
#include <string.h>
#include <time.h>

#include "erl_nif.h"
#include "ra_gpu_batch.h"
#include "ra_gpu_wal.h"

#define SUBMIT_DIRTY_MSGS 2048u   /* ~20 us of rgb_submit host work: above this the call goes to a dirty scheduler */

typedef struct {
  rgb_ctx *ctx;
  uint32_t ring_capacity, n_members, n_groups, n_servers;
  ErlNifTid tid;
  ErlNifPid owner;                 /* default owner (start_collector/2) */
  /* per-process fan-back: owner_of[server] = 1 + index into pids, 0 = default owner (guarded by own_mu) */
  ErlNifMutex *own_mu;
  ErlNifPid *pids;
  uint32_t n_pids, cap_pids;
  unsigned char *live;          /* live[k]: slot k of pids holds a registered owner (unregister_owner/2 frees it) */
  uint32_t *ht; uint32_t ht_cap, ht_used;      /* pid -> slot + 1 (open addressing; 0 empty, 0xFFFFFFFF a released entry) */
  uint32_t *free_slots; uint32_t n_free;       /* slots released by unregister_owner/2 */
  uint32_t *owner_of;
  atomic_ullong fb_ns, fb_decisions, fb_batches;   /* time the collector thread spent in fan_back (fan_back_stats/1) */
  uint32_t *tix; uint32_t tix_cap, tix_gen;   /* fan_back scratch: per-slot position in the batch's owner list, generation-stamped */
  atomic_int collector_on, stop;
  rgb_comm *comm;                  /* this context's rank in the node's leaderboard all-gather (comm_init/4), or NULL */
} nif_ctx;

static ErlNifResourceType *CTX_TYPE;

static ERL_NIF_TERM mk_error(ErlNifEnv *env, nif_ctx *c, int rc) {
  if (rc == RGB_E_HIP)   /* {error, {hip, Code}}: the caller falls back to ra_server */
    return enif_make_tuple2(env, enif_make_atom(env, "error"),
                            enif_make_tuple2(env, enif_make_atom(env, "hip"),
                                             enif_make_int(env, c ? rgb_last_hip_error(c->ctx) : 0)));
  static const char *names[] = {"ok", "invalid", "nomem", "hip", "state", "full", "empty", "unsupported", "nodevice", "comm"};
  int k = -rc;
  return enif_make_tuple2(env, enif_make_atom(env, "error"),
                          enif_make_atom(env, (k >= 0 && k <= 9) ? names[k] : "unknown"));
}

static void ctx_dtor(ErlNifEnv *env, void *obj) {
  nif_ctx *c = (nif_ctx *)obj;
  (void)env;
  atomic_store(&c->stop, 1);
  /* the collector thread drops its reference when it ends: if that was the last one this destructor runs ON that
   * thread, and a thread cannot join itself (EDEADLK) */
  if (atomic_load(&c->collector_on) && !enif_equal_tids(enif_thread_self(), c->tid)) {
    rgb_wake(c->ctx);
    enif_thread_join(c->tid, NULL);
  }
  if (c->comm) rgb_comm_destroy(c->comm);
  c->comm = NULL;
  if (c->ctx) rgb_close(c->ctx);
  c->ctx = NULL;
  if (c->own_mu) enif_mutex_destroy(c->own_mu);
  if (c->pids) enif_free(c->pids);
  if (c->live) enif_free(c->live);
  if (c->ht) enif_free(c->ht);
  if (c->free_slots) enif_free(c->free_slots);
  if (c->tix) enif_free(c->tix);
  if (c->owner_of) enif_free(c->owner_of);
}

static int get_ctx(ErlNifEnv *env, ERL_NIF_TERM t, nif_ctx **c) {
  return enif_get_resource(env, t, CTX_TYPE, (void **)c) && (*c)->ctx != NULL;
}

/* open(Device, MaxRuns, RingSlots, RingCapacity) -> {ok, Ctx} | {error, _} */
static ERL_NIF_TERM nif_open(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  rgb_config cfg;
  unsigned dev, runs, slots, cap;
  (void)argc;
  rgb_default_config(&cfg);
  if (!enif_get_uint(env, argv[0], &dev) || !enif_get_uint(env, argv[1], &runs) ||
      !enif_get_uint(env, argv[2], &slots) || !enif_get_uint(env, argv[3], &cap))
    return enif_make_badarg(env);
  cfg.device = (int32_t)dev; cfg.max_runs = runs; cfg.ring_slots = slots; cfg.ring_capacity = cap;
  rgb_ctx *ctx = NULL;
  int rc = rgb_open(&cfg, &ctx);
  if (rc) return mk_error(env, NULL, rc);
  nif_ctx *c = (nif_ctx *)enif_alloc_resource(CTX_TYPE, sizeof *c);
  memset(c, 0, sizeof *c);
  c->ctx = ctx; c->ring_capacity = cap;
  c->own_mu = enif_mutex_create((char *)"rgb_owners");
  ERL_NIF_TERM term = enif_make_resource(env, c);
  enif_release_resource(c);
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), term);
}

static ERL_NIF_TERM nif_register_groups(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned g, n;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &g) || !enif_get_uint(env, argv[2], &n))
    return enif_make_badarg(env);
  int rc = rgb_register_groups(c->ctx, g, n);
  if (rc) return mk_error(env, c, rc);
  c->n_members = n; c->n_groups = g; c->n_servers = rgb_n_servers(c->ctx);
  c->owner_of = (uint32_t *)enif_alloc((size_t)c->n_servers * sizeof(uint32_t));
  if (!c->owner_of) return mk_error(env, c, RGB_E_NOMEM);
  memset(c->owner_of, 0, (size_t)c->n_servers * sizeof(uint32_t));
  return enif_make_atom(env, "ok");
}

/* register_owner(Ctx, FirstServer, N, Pid) -> ok: the gen_statem Pid owns servers [FirstServer, FirstServer+N)
 * (a ra_server_proc registers the one server it is; a batching process may own a range).  A pid that is already in
 * the table keeps its slot, a slot freed by unregister_owner/2 is reused: the table is bounded by the number of LIVE
 * owners, not by the number of restarts. */
/* O(1): a pid -> slot hash (open addressing over enif_hash of the pid, compared with enif_compare_pids) and a free
 * list of the slots unregister_owner/2 released -- one ra_server_proc per server registers itself, so a linear scan of
 * the table made the registration of S owners O(S^2) and held own_mu (which fan_back needs) all the while */
static uint64_t pid_hash(const ErlNifPid *pid) {
  uint64_t x = enif_hash(ERL_NIF_INTERNAL_HASH, pid->pid, 0);    /* (the term's bits are opaque: enif_hash, OTP 20+) */
  x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
  return x;
}
/* Rebuilds the hash without its tombstones; the capacity follows the LIVE owners (round 5: it doubled on every
 * rebuild -- ht_used also counts tombstones -- so register / unregister churn from restarting ra_server_procs grew the
 * table with the number of restarts and made every rebuild an O(slots) walk under own_mu): the same size while the
 * live entries fill at most a quarter of it, twice the size only when they need the room. */
static int owner_ht_grow(nif_ctx *c) {                              /* own_mu held */
  uint32_t n_live = 0;
  for (uint32_t k = 0; k < c->n_pids; ++k) n_live += c->live[k] ? 1u : 0u;
  uint32_t cap = c->ht_cap ? c->ht_cap : 256;
  while ((uint64_t)(n_live + 1u) * 4u > cap) cap *= 2;
  uint32_t *ht = (uint32_t *)enif_alloc((size_t)cap * sizeof(uint32_t));
  if (!ht) return 0;
  memset(ht, 0, (size_t)cap * sizeof(uint32_t));
  for (uint32_t k = 0; k < c->n_pids; ++k) {
    if (!c->live[k]) continue;
    uint32_t h = (uint32_t)(pid_hash(&c->pids[k]) & (cap - 1));
    while (ht[h]) h = (h + 1) & (cap - 1);
    ht[h] = k + 1;
  }
  if (c->ht) enif_free(c->ht);
  c->ht = ht; c->ht_cap = cap; c->ht_used = n_live;           /* occupied cells = live entries (no tombstones left) */
  return 1;
}
static uint32_t owner_find(nif_ctx *c, const ErlNifPid *pid) {      /* own_mu held; slot + 1, 0 = not registered */
  if (!c->ht_cap) return 0;
  uint32_t h = (uint32_t)(pid_hash(pid) & (c->ht_cap - 1));
  while (c->ht[h]) {
    const uint32_t k = c->ht[h] - 1;
    if (k != 0xFFFFFFFEu && c->live[k] && enif_compare_pids(&c->pids[k], pid) == 0) return k + 1;
    h = (h + 1) & (c->ht_cap - 1);
  }
  return 0;
}
static uint32_t owner_slot(nif_ctx *c, const ErlNifPid *pid) {      /* own_mu held; 0 = no memory */
  uint32_t idx = owner_find(c, pid);
  if (idx) return idx;
  if ((c->ht_used + 1) * 2 > c->ht_cap && !owner_ht_grow(c)) return 0;   /* (also clears the tombstones) */
  if (c->n_free) idx = c->free_slots[--c->n_free] + 1;
  else {
    if (c->n_pids == c->cap_pids) {
      uint32_t cap = c->cap_pids ? c->cap_pids * 2 : 64;
      ErlNifPid *p = (ErlNifPid *)enif_alloc((size_t)cap * sizeof(ErlNifPid));
      unsigned char *l = (unsigned char *)enif_alloc(cap);
      uint32_t *f = (uint32_t *)enif_alloc((size_t)cap * sizeof(uint32_t));
      if (!p || !l || !f) { if (p) enif_free(p); if (l) enif_free(l); if (f) enif_free(f); return 0; }
      if (c->n_pids) { memcpy(p, c->pids, (size_t)c->n_pids * sizeof(ErlNifPid)); memcpy(l, c->live, c->n_pids); }
      if (c->n_free) memcpy(f, c->free_slots, (size_t)c->n_free * sizeof(uint32_t));
      if (c->pids) enif_free(c->pids);
      if (c->live) enif_free(c->live);
      if (c->free_slots) enif_free(c->free_slots);
      c->pids = p; c->live = l; c->free_slots = f; c->cap_pids = cap;
    }
    idx = ++c->n_pids;
  }
  c->pids[idx - 1] = *pid; c->live[idx - 1] = 1;
  uint32_t h = (uint32_t)(pid_hash(pid) & (c->ht_cap - 1));
  while (c->ht[h] && c->ht[h] != 0xFFFFFFFFu) h = (h + 1) & (c->ht_cap - 1);
  if (!c->ht[h]) c->ht_used += 1;                             /* (a reused tombstone was counted when it was first filled) */
  c->ht[h] = idx;
  return idx;
}

static ERL_NIF_TERM nif_register_owner(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned first, n; ErlNifPid pid;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &first) || !enif_get_uint(env, argv[2], &n) ||
      !enif_get_local_pid(env, argv[3], &pid) || !c->owner_of || (uint64_t)first + n > c->n_servers)
    return enif_make_badarg(env);
  enif_mutex_lock(c->own_mu);
  const uint32_t idx = owner_slot(c, &pid);
  if (!idx) { enif_mutex_unlock(c->own_mu); return mk_error(env, c, RGB_E_NOMEM); }
  for (unsigned k = 0; k < n; ++k) c->owner_of[first + k] = idx;
  enif_mutex_unlock(c->own_mu);
  return enif_make_atom(env, "ok");
}

/* unregister_owner(Ctx, Pid) -> ok: Pid owns nothing any more (its servers fall back to the default owner of
 * start_collector/2) and its slot is free for the next register_owner/4 -- what a terminating or restarting
 * ra_server_proc calls (src/ra_server_proc.erl terminate/3) */
static ERL_NIF_TERM nif_unregister_owner(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifPid pid;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_local_pid(env, argv[1], &pid) || !c->owner_of) return enif_make_badarg(env);
  enif_mutex_lock(c->own_mu);
  const uint32_t idx = owner_find(c, &pid);
  if (idx) {
    for (uint32_t sv = 0; sv < c->n_servers; ++sv)          /* (dirty CPU scheduler: the one O(servers) walk left) */
      if (c->owner_of[sv] == idx) c->owner_of[sv] = 0;
    c->live[idx - 1] = 0;
    uint32_t h = (uint32_t)(pid_hash(&pid) & (c->ht_cap - 1));
    while (c->ht[h] != idx) h = (h + 1) & (c->ht_cap - 1);
    c->ht[h] = 0xFFFFFFFFu;                                  /* released: probes walk over it, inserts reuse it */
    c->free_slots[c->n_free++] = idx - 1;
  }
  enif_mutex_unlock(c->own_mu);
  return enif_make_atom(env, "ok");
}

/* owner_slots(Ctx) -> {Slots, Live}: size of the owner table and how many of its slots hold a registered process
 * (diagnostics: Slots stays bounded by the peak number of live owners whatever the number of restarts) */
static ERL_NIF_TERM nif_owner_slots(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  if (!get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  enif_mutex_lock(c->own_mu);
  unsigned live = 0;
  for (uint32_t k = 0; k < c->n_pids; ++k) live += c->live[k] ? 1u : 0u;
  const unsigned slots = c->n_pids;
  enif_mutex_unlock(c->own_mu);
  return enif_make_tuple2(env, enif_make_uint(env, slots), enif_make_uint(env, live));
}

/* fan_back_stats(Ctx) -> {Batches, Decisions, Nanoseconds}: what the collector thread has spent splitting batches
 * per owner and sending them (allocation of the per-owner binaries + enif_send included) */
static ERL_NIF_TERM nif_fan_back_stats(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  if (!get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  return enif_make_tuple3(env, enif_make_uint64(env, atomic_load(&c->fb_batches)),
                          enif_make_uint64(env, atomic_load(&c->fb_decisions)), enif_make_uint64(env, atomic_load(&c->fb_ns)));
}

/* upload_state(Ctx, FirstServer, <<rgb_server_state x N>>) */
static ERL_NIF_TERM nif_upload_state(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned first; ErlNifBinary b;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &first) ||
      !enif_inspect_binary(env, argv[2], &b) || b.size % sizeof(rgb_server_state))
    return enif_make_badarg(env);
  int rc = rgb_upload_state(c->ctx, first, (uint32_t)(b.size / sizeof(rgb_server_state)),
                            (const rgb_server_state *)b.data);
  return rc ? mk_error(env, c, rc) : enif_make_atom(env, "ok");
}

static ERL_NIF_TERM nif_download_state(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned first, n; ErlNifBinary b;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &first) || !enif_get_uint(env, argv[2], &n))
    return enif_make_badarg(env);
  if (!enif_alloc_binary((size_t)n * sizeof(rgb_server_state), &b)) return mk_error(env, c, RGB_E_NOMEM);
  int rc = rgb_download_state(c->ctx, first, n, (rgb_server_state *)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, c, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &b));
}

/* submit(Ctx, <<rgb_msg x N>>, Tick): copies into the pinned ring and returns.  Thread-safe: any process may call it. */
static ERL_NIF_TERM submit_body(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary b; uint64_t tick;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &b) || b.size % sizeof(rgb_msg) ||
      !enif_get_uint64(env, argv[2], &tick))
    return enif_make_badarg(env);
  /* submit/4: the batch's range list (written events of more than two ranges, RGB_MF_SEQX): <<First:64/little,
   * Last:64/little>> per entry -- rgb_submit_seq copies it with the batch */
  ErlNifBinary r; r.size = 0; r.data = NULL;
  if (argc == 4 && (!enif_inspect_binary(env, argv[3], &r) || r.size % (2 * sizeof(uint64_t)))) return enif_make_badarg(env);
  int rc = rgb_submit_seq(c->ctx, (const rgb_msg *)b.data, (uint32_t)(b.size / sizeof(rgb_msg)), tick,
                          r.size ? (const uint64_t *)r.data : NULL, (uint32_t)(r.size / (2 * sizeof(uint64_t))));
  return rc ? mk_error(env, c, rc) : enif_make_atom(env, "ok");
}
static ERL_NIF_TERM nif_submit(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  ErlNifBinary b;
  if (enif_inspect_binary(env, argv[1], &b) && b.size / sizeof(rgb_msg) > SUBMIT_DIRTY_MSGS)
    return enif_schedule_nif(env, "submit_dirty", ERL_NIF_DIRTY_JOB_CPU_BOUND, submit_body, argc, argv);
  return submit_body(env, argc, argv);
}

static int do_collect(nif_ctx *c, ErlNifBinary *dec, ErlNifBinary *rpc, uint32_t *n, uint32_t *nr, uint64_t *tick) {
  /* the binaries are sized from the batch itself (rgb_peek waits for it): exactly byte_size(DecisionsBin) div 64
   * decisions and byte_size(RpcsBin) div 56 rpc records, no ring-capacity-sized allocation per batch */
  uint32_t want = 0, want_r = 0;
  int rc = rgb_peek(c->ctx, &want, &want_r);
  if (rc) return rc;
  if (!enif_alloc_binary((size_t)want * sizeof(rgb_decision), dec)) return RGB_E_NOMEM;
  if (!enif_alloc_binary((size_t)want_r * sizeof(rgb_rpc), rpc)) { enif_release_binary(dec); return RGB_E_NOMEM; }
  rc = rgb_collect(c->ctx, (rgb_decision *)dec->data, want, n, (rgb_rpc *)rpc->data, want_r, nr, tick);
  if (rc) { enif_release_binary(dec); enif_release_binary(rpc); return rc; }
  return rc;
}

/* collect(Ctx) -> {ok, Tick, NDecisions, DecisionsBin, RpcsBin}   (dirty IO-bound NIF: it waits) */
static ERL_NIF_TERM nif_collect(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary dec, rpc; uint32_t n = 0, nr = 0; uint64_t tick = 0;
  (void)argc;
  if (!get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  if (atomic_load(&c->collector_on) == 1)     /* one consumer: the collector thread owns rgb_collect while it runs */
    return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "collector_running"));
  int rc = do_collect(c, &dec, &rpc, &n, &nr, &tick);
  if (rc) return mk_error(env, c, rc);
  return enif_make_tuple5(env, enif_make_atom(env, "ok"), enif_make_uint64(env, tick), enif_make_uint64(env, n),
                          enif_make_binary(env, &dec), enif_make_binary(env, &rpc));
}

static void send_batch(ErlNifEnv *env, const ErlNifPid *to, uint64_t tick, uint32_t n, ErlNifBinary *dec, ErlNifBinary *rpc) {
  ERL_NIF_TERM msg = enif_make_tuple5(env, enif_make_atom(env, "ra_gpu_batch"), enif_make_uint64(env, tick),
                                      enif_make_uint64(env, n), enif_make_binary(env, dec), enif_make_binary(env, rpc));
  enif_send(NULL, to, env, msg);
  enif_clear_env(env);
}

/* One collected batch -> one message per owning process, in two linear passes: (1) count decisions and rpc records
 * per owner and list the owners in order of first appearance, (2) append every decision (submission order is kept
 * inside an owner) and its rpc records -- rgb_collect returns them ordered by msg_index -- to the owner's binaries,
 * with msg_index rewritten to the decision's position inside that owner's DecisionsBin. */
/* d / r: the batch where the device wrote it -- the pinned ring slot of an rgb_collect_view (ABI v9), read ONCE here
 * and released by the caller afterwards: the decisions are copied a single time, straight into the per-owner binaries
 * (through rgb_collect they were copied into a batch binary first and regrouped from there). */
static int fan_back(nif_ctx *c, ErlNifEnv *env, uint64_t tick, uint32_t n, uint32_t nr, const rgb_decision *d, const rgb_rpc *r) {
  enif_mutex_lock(c->own_mu);
  if (c->n_pids == 0) {
    enif_mutex_unlock(c->own_mu);
    /* no owner table: the whole batch to the default owner, as two binaries */
    ErlNifBinary dec, rpc;
    if (!enif_alloc_binary((size_t)n * sizeof(rgb_decision), &dec)) return RGB_E_NOMEM;
    if (!enif_alloc_binary((size_t)nr * sizeof(rgb_rpc), &rpc)) { enif_release_binary(&dec); return RGB_E_NOMEM; }
    if (n) memcpy(dec.data, d, (size_t)n * sizeof(rgb_decision));
    if (nr) memcpy(rpc.data, r, (size_t)nr * sizeof(rgb_rpc));
    send_batch(env, &c->owner, tick, n, &dec, &rpc);
    return RGB_OK;
  }
  /* Under the lock only what depends on the owner table: owner of every decision and the pids involved (copied).
   * The binaries are filled and sent after it is released -- register_owner/4 runs on a normal scheduler and must
   * not wait for a 64k-decision fan-out.  tix[] (one word per table slot, generation-stamped: no memset per batch)
   * lives in the context; the per-batch scratch is bounded by the batch. */
  const uint32_t n_own = c->n_pids + 1;                       /* bucket 0 = default owner */
  if (c->tix_cap < n_own) {
    uint32_t *t2 = (uint32_t *)enif_alloc((size_t)n_own * 2 * sizeof(uint32_t));
    if (!t2) { enif_mutex_unlock(c->own_mu); return RGB_E_NOMEM; }
    memset(t2, 0, (size_t)n_own * 2 * sizeof(uint32_t));
    if (c->tix) enif_free(c->tix);
    c->tix = t2; c->tix_cap = n_own; c->tix_gen = 0;
  }
  if (++c->tix_gen == 0) { memset(c->tix, 0, (size_t)c->tix_cap * 2 * sizeof(uint32_t)); c->tix_gen = 1; }
  uint32_t *tix = c->tix, *tgen = c->tix + c->tix_cap;       /* tix[o] valid iff tgen[o] == tix_gen */
  uint32_t *own = (uint32_t *)enif_alloc((size_t)(n + 1) * 6 * sizeof(uint32_t));
  ErlNifPid *to = NULL;
  int rc = RGB_OK;
  if (!own) rc = RGB_E_NOMEM;
  uint32_t n_t = 0;
  uint32_t *cnt_d = NULL, *cnt_r = NULL, *fill_d = NULL, *fill_r = NULL, *slot_of = NULL;
  if (rc == RGB_OK) {
    cnt_d = own + (n + 1); cnt_r = cnt_d + (n + 1); fill_d = cnt_r + (n + 1); fill_r = fill_d + (n + 1);
    slot_of = fill_r + (n + 1);                               /* decision i -> position of its owner in the list */
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t o = d[i].server < c->n_servers ? c->owner_of[d[i].server] : 0;
      if (tgen[o] != c->tix_gen) { tgen[o] = c->tix_gen; own[n_t] = o; cnt_d[n_t] = cnt_r[n_t] = fill_d[n_t] = fill_r[n_t] = 0; tix[o] = ++n_t; }
      slot_of[i] = tix[o] - 1;
      cnt_d[tix[o] - 1] += 1; cnt_r[tix[o] - 1] += d[i].n_rpcs;
    }
    to = (ErlNifPid *)enif_alloc((size_t)(n_t ? n_t : 1) * sizeof(ErlNifPid));
    if (!to) rc = RGB_E_NOMEM;
    for (uint32_t t = 0; rc == RGB_OK && t < n_t; ++t) to[t] = (own[t] && c->live[own[t] - 1]) ? c->pids[own[t] - 1] : c->owner;
  }
  enif_mutex_unlock(c->own_mu);
  if (rc == RGB_OK) {
    /* ONE decisions binary and ONE rpc binary per batch, grouped by owner in first-appearance order; every owner
     * gets SUB-BINARIES of them (no allocation and no copy per owner: with one gen_statem per server that is per
     * decision -- the mock BEAM measured 263 ns per decision at 4 096 owners with two binaries allocated per owner).
     * The messages are sent with msg_env = NULL (copied: the two parent terms stay valid until the env is cleared). */
    ErlNifBinary all_d, all_r;
    uint32_t tot_r = 0;
    for (uint32_t t = 0; t < n_t; ++t) tot_r += cnt_r[t];
    if (!enif_alloc_binary((size_t)n * sizeof(rgb_decision), &all_d)) rc = RGB_E_NOMEM;
    else if (!enif_alloc_binary((size_t)tot_r * sizeof(rgb_rpc), &all_r)) { enif_release_binary(&all_d); rc = RGB_E_NOMEM; }
    if (rc == RGB_OK) {
      /* fill_d / fill_r become the owners' running cursors: start at their offsets */
      uint32_t od = 0, orr = 0;
      for (uint32_t t = 0; t < n_t; ++t) { fill_d[t] = od; fill_r[t] = orr; od += cnt_d[t]; orr += cnt_r[t]; }
      uint32_t *first_d = (uint32_t *)enif_alloc((size_t)(n_t ? n_t : 1) * 2 * sizeof(uint32_t));
      if (!first_d) { enif_release_binary(&all_d); enif_release_binary(&all_r); rc = RGB_E_NOMEM; }
      else {
        uint32_t *first_r = first_d + (n_t ? n_t : 1);
        for (uint32_t t = 0; t < n_t; ++t) { first_d[t] = fill_d[t]; first_r[t] = fill_r[t]; }
        uint32_t k = 0;                                        /* cursor into the rpc records */
        for (uint32_t i = 0; i < n; ++i) {
          const uint32_t t = slot_of[i];
          const uint32_t at = fill_d[t]++;
          ((rgb_decision *)all_d.data)[at] = d[i];
          for (uint32_t q = 0; q < d[i].n_rpcs; ++q) {
            rgb_rpc x = r[k++];
            x.msg_index = at - first_d[t];                     /* position inside that owner's DecisionsBin */
            ((rgb_rpc *)all_r.data)[fill_r[t]++] = x;
          }
        }
        ERL_NIF_TERM td = enif_make_binary(env, &all_d), tr = enif_make_binary(env, &all_r);
        for (uint32_t t = 0; t < n_t; ++t) {
          ERL_NIF_TERM msg = enif_make_tuple5(env, enif_make_atom(env, "ra_gpu_batch"), enif_make_uint64(env, tick),
                                              enif_make_uint64(env, cnt_d[t]),
                                              enif_make_sub_binary(env, td, (size_t)first_d[t] * sizeof(rgb_decision), (size_t)cnt_d[t] * sizeof(rgb_decision)),
                                              enif_make_sub_binary(env, tr, (size_t)first_r[t] * sizeof(rgb_rpc), (size_t)cnt_r[t] * sizeof(rgb_rpc)));
          enif_send(NULL, &to[t], NULL, msg);
        }
        enif_clear_env(env);
        enif_free(first_d);
      }
    }
  }
  if (to) enif_free(to);
  if (own) enif_free(own);
  return rc;
}

/* the collector thread: owns the consumer side of the ring, parks in rgb_wait while nothing is in flight.  It takes
 * every batch as a VIEW (rgb_collect_view: pointers into the pinned slot the device wrote), fans it out from there and
 * gives the slot back */
static void *collector_main(void *arg) {
  nif_ctx *c = (nif_ctx *)arg;
  ErlNifEnv *env = enif_alloc_env();
  while (!atomic_load(&c->stop)) {
    if (rgb_wait(c->ctx, 250) != RGB_OK) continue;            /* timeout or rgb_wake: re-check stop */
    rgb_view v;
    int rc = rgb_collect_view(c->ctx, &v);
    if (rc == RGB_E_EMPTY) continue;
    if (rc == RGB_OK) {
      const uint32_t n = v.n;
      struct timespec a, b;
      clock_gettime(CLOCK_MONOTONIC, &a);
      rc = fan_back(c, env, v.tick, v.n, v.n_rpcs, v.decisions, v.rpcs);
      (void)rgb_release(c->ctx, v.slot);
      clock_gettime(CLOCK_MONOTONIC, &b);
      atomic_fetch_add(&c->fb_ns, (unsigned long long)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)));
      atomic_fetch_add(&c->fb_decisions, (unsigned long long)n);
      atomic_fetch_add(&c->fb_batches, 1ull);
    }
    if (rc != RGB_OK) {
      /* a persistent error is reported ONCE to the default owner and ends the thread (the owner falls back to
       * ra_server and may start a new collector): no hot error loop */
      enif_send(NULL, &c->owner, env, enif_make_tuple2(env, enif_make_atom(env, "ra_gpu_batch_error"), mk_error(env, c, rc)));
      enif_clear_env(env);
      break;
    }
  }
  enif_free_env(env);
  /* finished by itself (2): collect/1 works again at once, and the next start_collector/2 -- or stop_collector/1, or
   * the destructor -- joins this thread */
  /* from running (1) -- or from STARTING (3): the thread can end on an error before start_collector/2 has moved 3 to
   * 1, and an exit that only looked for 1 was lost there (start_collector then published 1 for a dead thread: collect/1
   * refused, a new start_collector/2 badarg, although the owner had been told it may start one).  A stopper's 4 stays. */
  /* (a loop: between a failed CAS from 1 and the CAS from 3, start_collector/2 can move 3 to 1 -- tried once each, both
   * failed and 1 stayed for a dead thread) */
  for (;;) {
    int st = atomic_load(&c->collector_on);
    if (st != 1 && st != 3) break;                           /* a stopper's 4: it joins this thread */
    if (atomic_compare_exchange_strong(&c->collector_on, &st, 2)) break;
  }
  enif_release_resource(c);
  return NULL;
}

static ERL_NIF_TERM nif_start_collector(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  ErlNifPid owner;
  if (!get_ctx(env, argv[0], &c) || !enif_get_local_pid(env, argv[1], &owner)) return enif_make_badarg(env);
  /* exactly one caller wins the right to start it: 0 -> 3 (starting), or 2 -> 3 when the previous thread ended by
   * itself after an error (joined here).  A running (1) or starting (3) collector: badarg */
  int was = 0;
  if (!atomic_compare_exchange_strong(&c->collector_on, &was, 3)) {
    was = 2;
    if (!atomic_compare_exchange_strong(&c->collector_on, &was, 3)) return enif_make_badarg(env);
    enif_thread_join(c->tid, NULL);
  }
  c->owner = owner;
  enif_keep_resource(c);
  atomic_store(&c->stop, 0);
  if (enif_thread_create((char *)"rgb_collector", &c->tid, collector_main, c, NULL)) {
    atomic_store(&c->collector_on, 0);
    enif_release_resource(c);
    return mk_error(env, c, RGB_E_NOMEM);
  }
  was = 3;
  atomic_compare_exchange_strong(&c->collector_on, &was, 1);   /* (a thread that already ended moved 3 to 2 itself) */
  return enif_make_atom(env, "ok");
}

/* stop_collector(Ctx) -> ok: ends the collector thread and drops the reference it held on the context (without
 * it the thread would keep the resource alive for ever and the destructor could never run) */
static ERL_NIF_TERM nif_stop_collector(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  if (!get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  /* exactly one caller joins the thread: running (1) or already ended (2) -> stopping (4) */
  int was = 1;
  if (!atomic_compare_exchange_strong(&c->collector_on, &was, 4)) {
    was = 2;
    if (!atomic_compare_exchange_strong(&c->collector_on, &was, 4)) return enif_make_badarg(env);
  }
  atomic_store(&c->stop, 1);
  rgb_wake(c->ctx);                                            /* the thread may be parked in rgb_wait */
  enif_thread_join(c->tid, NULL);
  atomic_store(&c->collector_on, 0);
  atomic_store(&c->stop, 0);
  return enif_make_atom(env, "ok");
}

/* snapshot(Ctx, NGroups) -> {ok, <<rgb_leaderboard_row x NGroups>>} */
static ERL_NIF_TERM nif_snapshot(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned g; ErlNifBinary b;
  (void)argc;
  /* rgb_snapshot writes one row per REGISTERED group: the binary is sized from the registration and a caller
   * whose NGroups disagrees gets badarg (a smaller binary would be overrun) */
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &g) || c->n_groups == 0 || g != c->n_groups)
    return enif_make_badarg(env);
  if (!enif_alloc_binary((size_t)c->n_groups * sizeof(rgb_leaderboard_row), &b)) return mk_error(env, c, RGB_E_NOMEM);
  int rc = rgb_snapshot(c->ctx, (rgb_leaderboard_row *)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, c, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &b));
}

/* route(GroupUId, NContexts) -> 0..NContexts-1: which context (= GPU) of this node owns the Raft group; a pure
 * function (rgb_route: splitmix64(GroupUId) rem NContexts), so every process routes without asking anybody */
static ERL_NIF_TERM nif_route(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  uint64_t uid; unsigned n;
  (void)argc;
  if (!enif_get_uint64(env, argv[0], &uid) || !enif_get_uint(env, argv[1], &n) || n == 0) return enif_make_badarg(env);
  return enif_make_uint(env, rgb_route(uid, n));
}

/* ---- the node's leaderboard all-gather (one context per GPU; include/ra_gpu_batch.h "Multi-GPU") ----
 * comm_unique_id() -> {ok, <<Id:128/binary>>}: one context's owner creates the id and sends it to the owners of the
 * others (plain Erlang messages); comm_init(Ctx, Id, NRanks, Rank) -> ok is collective -- every owner calls it, on a
 * dirty scheduler, before any of them returns; allgather_leaderboard(Ctx, NRows) -> {ok, <<rgb_leaderboard_row x NRanks
 * x NRows>>}: collective as well, NRows = the largest group count of any context (rank r's rows start at r * NRows;
 * the rows beyond a context's own groups are zero) */
static ERL_NIF_TERM nif_comm_unique_id(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  ErlNifBinary b;
  (void)argc; (void)argv;
  if (!enif_alloc_binary(RGB_COMM_ID_BYTES, &b)) return mk_error(env, NULL, RGB_E_NOMEM);
  int rc = rgb_comm_unique_id(b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, NULL, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &b));
}

static ERL_NIF_TERM nif_comm_init(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary id; unsigned n, rank;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &id) || id.size != RGB_COMM_ID_BYTES ||
      !enif_get_uint(env, argv[2], &n) || !enif_get_uint(env, argv[3], &rank) || c->comm != NULL)
    return enif_make_badarg(env);
  int rc = rgb_comm_init_rank(c->ctx, id.data, n, rank, &c->comm);
  if (rc) return mk_error(env, c, rc);
  return enif_make_atom(env, "ok");
}

static ERL_NIF_TERM nif_allgather_leaderboard(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned n_rows; ErlNifBinary b;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &n_rows) || c->comm == NULL || n_rows < c->n_groups)
    return enif_make_badarg(env);
  const size_t rows = (size_t)n_rows * rgb_comm_n_ranks(c->comm);
  if (!enif_alloc_binary(rows * sizeof(rgb_leaderboard_row), &b)) return mk_error(env, c, RGB_E_NOMEM);
  int rc = rgb_leaderboard_allgather_host(c->ctx, c->comm, n_rows, (rgb_leaderboard_row *)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, c, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &b));
}

static int on_load(ErlNifEnv *env, void **priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  CTX_TYPE = enif_open_resource_type(env, NULL, "ra_gpu_batch_ctx", ctx_dtor, ERL_NIF_RT_CREATE, NULL);
  return CTX_TYPE == NULL || rgb_abi_version() != RGB_ABI_VERSION;
}

/* wal_checksums(Ctx, EntriesBin, DataBin) -> {ok, ChecksumsBin}: EntriesBin = n rgb_wal_entry records
 * (index, term, data_offset, data_len), DataBin = the packed payload bytes of the write batch,
 * ChecksumsBin = n little-endian 32-bit Adler-32 values (erlang:adler32([<<Idx:64,Term:64>> | Data]),
 * src/ra_log_wal.erl:528-534) */
static ERL_NIF_TERM nif_wal_checksums(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary e, d, out;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &e) ||
      !enif_inspect_binary(env, argv[2], &d) || e.size % sizeof(rgb_wal_entry) != 0)
    return enif_make_badarg(env);
  const uint32_t n = (uint32_t)(e.size / sizeof(rgb_wal_entry));
  if (!enif_alloc_binary((size_t)n * sizeof(uint32_t), &out)) return mk_error(env, c, RGB_E_NOMEM);
  int rc = rgb_wal_adler32(c->ctx, (const rgb_wal_entry *)e.data, n, d.data, d.size, (uint32_t *)out.data);
  if (rc) { enif_release_binary(&out); return mk_error(env, c, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &out));
}

/* wal_frame(Ctx, RecordsBin, DataBin, Flags) -> {ok, FramedBin}: RecordsBin = n rgb_wal_record
 * descriptors (out_offset is filled here, back to back from 0), DataBin = the HeaderData and payload
 * bytes they point into; FramedBin = the batch's on-disk bytes, what write_data/8 accumulates in
 * #batch.pending and flush_pending/1 hands to file:write/2 (src/ra_log_wal.erl:513-537, 638-650) */
static ERL_NIF_TERM nif_wal_frame(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary r, d, out; unsigned flags;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &r) ||
      !enif_inspect_binary(env, argv[2], &d) || !enif_get_uint(env, argv[3], &flags) ||
      r.size % sizeof(rgb_wal_record) != 0)
    return enif_make_badarg(env);
  const uint32_t n = (uint32_t)(r.size / sizeof(rgb_wal_record));
  rgb_wal_record *recs = (rgb_wal_record *)enif_alloc(r.size ? r.size : 1);
  if (!recs) return mk_error(env, c, RGB_E_NOMEM);
  memcpy(recs, r.data, r.size);                       /* binaries are immutable: lay out a copy */
  const uint64_t total = rgb_wal_layout(recs, n, 0);
  if (!enif_alloc_binary((size_t)total, &out)) { enif_free(recs); return mk_error(env, c, RGB_E_NOMEM); }
  int rc = rgb_wal_frame(c->ctx, recs, n, d.data, d.size, out.data, total, flags);
  enif_free(recs);
  if (rc) { enif_release_binary(&out); return mk_error(env, c, rc); }
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_binary(env, &out));
}

/* wal_recover_check(Ctx, FileBin) -> {ok, ScannedBin, NOk, clean | dropped_last | corrupt}: the
 * record walk of recover_records/5 plus validate_checksum/4 for every record, is_last_record/3 on
 * the first failure (src/ra_log_wal.erl:877-1033).  ScannedBin = n rgb_wal_scanned records; the
 * caller recovers the first NOk of them and throws wal_checksum_validation_failure on `corrupt`. */
static ERL_NIF_TERM nif_wal_recover_check(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary f, out;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &f)) return enif_make_badarg(env);
  uint32_t cap = 0, n = 0, end = 0, n_ok = 0, status = 0; uint64_t consumed = 0;
  int rc = rgb_wal_scan(f.data, f.size, NULL, 0, &cap, &consumed, &end);   /* count, then size the binary */
  if (rc) return mk_error(env, c, rc);
  if (!enif_alloc_binary((size_t)(cap ? cap : 1) * sizeof(rgb_wal_scanned), &out)) return mk_error(env, c, RGB_E_NOMEM);
  rc = rgb_wal_scan(f.data, f.size, (rgb_wal_scanned *)out.data, cap, &n, &consumed, &end);
  if (!rc) rc = rgb_wal_validate(c->ctx, f.data, f.size, (const rgb_wal_scanned *)out.data, n, &n_ok, &status);
  if (rc) { enif_release_binary(&out); return mk_error(env, c, rc); }
  enif_realloc_binary(&out, (size_t)n * sizeof(rgb_wal_scanned));
  const char *st = status == RGB_WAL_CLEAN ? "clean" : status == RGB_WAL_DROPPED_LAST ? "dropped_last" : "corrupt";
  return enif_make_tuple4(env, enif_make_atom(env, "ok"), enif_make_binary(env, &out),
                          enif_make_uint(env, n_ok), enif_make_atom(env, st));
}

static ErlNifFunc nif_funcs[] = {
  {"open", 4, nif_open, 0},
  {"register_groups", 3, nif_register_groups, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"upload_state", 3, nif_upload_state, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"download_state", 3, nif_download_state, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"route", 2, nif_route, 0},
  {"register_owner", 4, nif_register_owner, 0},
  {"unregister_owner", 2, nif_unregister_owner, ERL_NIF_DIRTY_JOB_CPU_BOUND},   /* scans owner_of: O(servers) */
  {"owner_slots", 1, nif_owner_slots, 0},
  {"fan_back_stats", 1, nif_fan_back_stats, 0},
  {"submit", 3, nif_submit, 0},
  {"submit", 4, nif_submit, 0},
  {"collect", 1, nif_collect, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"start_collector", 2, nif_start_collector, 0},
  {"stop_collector", 1, nif_stop_collector, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"snapshot", 2, nif_snapshot, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"comm_unique_id", 0, nif_comm_unique_id, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"comm_init", 4, nif_comm_init, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"allgather_leaderboard", 2, nif_allgather_leaderboard, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_checksums", 3, nif_wal_checksums, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_frame", 4, nif_wal_frame, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_recover_check", 2, nif_wal_recover_check, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(ra_gpu_batch, nif_funcs, on_load, NULL, NULL, NULL)
