/*
 * ra_gpu_batch_nif.c -- Erlang NIF shim over the C ABI of include/ra_gpu_batch.h.
 *
 * No logic lives here: every function unpacks its arguments, calls exactly one rgb_* entry
 * point and packs the result.  Records cross the boundary as binaries with the exact C layout
 * (erlang/ra_gpu_batch.erl builds and parses them), so the hot calls are one memcpy each.
 * Threading follows SURVEY.md section 8b: submit/2 is non-blocking (O(memcpy) into the pinned
 * ring); results come back either through the dirty NIF collect/1 or through the collector
 * thread started by start_collector/2, which loops on rgb_collect and enif_send()s
 * {ra_gpu_batch, Tick, Decisions, Rpcs} to the owning process (fan-back to the gen_statems is
 * done there).  The resource destructor calls rgb_close.
 *
 * Cannot be built against the real erl_nif.h in this image (no OTP): `make nif-check` syntax-checks it
 * against ra_amd/csrc/nif_stub/erl_nif.h, and tests/test_nif_shim_mock_beam.py EXECUTES it against a functional
 * mock of that API (tests/native/mock_beam) on top of the CPU-emulated library.  Build line for a machine with OTP >= 26:
 *   cc -O2 -fPIC -shared -I$ERL_INCLUDE -Iinclude ra_amd/csrc/ra_gpu_batch_nif.c \
 *      -Lra_amd/csrc -lra_gpu_batch -o priv/ra_gpu_batch_nif.so
 */
#include <string.h>

#include "erl_nif.h"
#include "ra_gpu_batch.h"
#include "ra_gpu_wal.h"

typedef struct {
  rgb_ctx *ctx;
  uint32_t ring_capacity, n_members;
  ErlNifTid tid;
  ErlNifPid owner;
  volatile int collector_on, stop;
} nif_ctx;

static ErlNifResourceType *CTX_TYPE;

static ERL_NIF_TERM mk_error(ErlNifEnv *env, nif_ctx *c, int rc) {
  if (rc == RGB_E_HIP)   /* {error, {hip, Code}}: the caller falls back to ra_server */
    return enif_make_tuple2(env, enif_make_atom(env, "error"),
                            enif_make_tuple2(env, enif_make_atom(env, "hip"),
                                             enif_make_int(env, c ? rgb_last_hip_error(c->ctx) : 0)));
  static const char *names[] = {"ok", "invalid", "nomem", "hip", "state", "full", "empty", "unsupported", "nodevice"};
  int k = -rc;
  return enif_make_tuple2(env, enif_make_atom(env, "error"),
                          enif_make_atom(env, (k >= 0 && k <= 8) ? names[k] : "unknown"));
}

static void ctx_dtor(ErlNifEnv *env, void *obj) {
  nif_ctx *c = (nif_ctx *)obj;
  (void)env;
  c->stop = 1;
  if (c->collector_on) enif_thread_join(c->tid, NULL);
  if (c->ctx) rgb_close(c->ctx);
  c->ctx = NULL;
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
  c->n_members = n;
  return enif_make_atom(env, "ok");
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

/* submit(Ctx, <<rgb_msg x N>>, Tick): copies into the pinned ring and returns */
static ERL_NIF_TERM nif_submit(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary b; uint64_t tick;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &b) || b.size % sizeof(rgb_msg) ||
      !enif_get_uint64(env, argv[2], &tick))
    return enif_make_badarg(env);
  int rc = rgb_submit(c->ctx, (const rgb_msg *)b.data, (uint32_t)(b.size / sizeof(rgb_msg)), tick);
  return rc ? mk_error(env, c, rc) : enif_make_atom(env, "ok");
}

static int do_collect(nif_ctx *c, ErlNifBinary *dec, ErlNifBinary *rpc, uint32_t *n, uint32_t *nr, uint64_t *tick) {
  uint32_t rcap = c->ring_capacity * (c->n_members > 1 ? c->n_members - 1 : 1);
  if (!enif_alloc_binary((size_t)c->ring_capacity * sizeof(rgb_decision), dec)) return RGB_E_NOMEM;
  if (!enif_alloc_binary((size_t)rcap * sizeof(rgb_rpc), rpc)) { enif_release_binary(dec); return RGB_E_NOMEM; }
  int rc = rgb_collect(c->ctx, (rgb_decision *)dec->data, c->ring_capacity, n, (rgb_rpc *)rpc->data, rcap, nr, tick);
  if (rc) { enif_release_binary(dec); enif_release_binary(rpc); return rc; }
  /* the binaries carry exactly the records that were produced: byte_size(DecisionsBin) div 64 decisions,
   * byte_size(RpcsBin) div 56 rpc records */
  enif_realloc_binary(dec, (size_t)*n * sizeof(rgb_decision));
  enif_realloc_binary(rpc, (size_t)*nr * sizeof(rgb_rpc));
  return rc;
}

/* collect(Ctx) -> {ok, Tick, NDecisions, DecisionsBin, RpcsBin}   (dirty IO-bound NIF: it waits) */
static ERL_NIF_TERM nif_collect(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; ErlNifBinary dec, rpc; uint32_t n = 0, nr = 0; uint64_t tick = 0;
  (void)argc;
  if (!get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  int rc = do_collect(c, &dec, &rpc, &n, &nr, &tick);
  if (rc) return mk_error(env, c, rc);
  return enif_make_tuple5(env, enif_make_atom(env, "ok"), enif_make_uint64(env, tick), enif_make_uint64(env, n),
                          enif_make_binary(env, &dec), enif_make_binary(env, &rpc));
}

/* the collector thread: owns rgb_collect, fans whole batches back to the owner process */
static void *collector_main(void *arg) {
  nif_ctx *c = (nif_ctx *)arg;
  ErlNifEnv *env = enif_alloc_env();
  while (!c->stop) {
    ErlNifBinary dec, rpc; uint32_t n = 0, nr = 0; uint64_t tick = 0;
    int rc = do_collect(c, &dec, &rpc, &n, &nr, &tick);
    if (rc == RGB_E_EMPTY) { continue; }          /* nothing in flight: a real shim parks on a condvar here */
    ERL_NIF_TERM msg = rc ? enif_make_tuple2(env, enif_make_atom(env, "ra_gpu_batch_error"), mk_error(env, c, rc))
                          : enif_make_tuple5(env, enif_make_atom(env, "ra_gpu_batch"), enif_make_uint64(env, tick),
                                             enif_make_uint64(env, n), enif_make_binary(env, &dec),
                                             enif_make_binary(env, &rpc));
    enif_send(NULL, &c->owner, env, msg);
    enif_clear_env(env);
  }
  enif_free_env(env);
  enif_release_resource(c);
  return NULL;
}

static ERL_NIF_TERM nif_start_collector(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || c->collector_on || !enif_get_local_pid(env, argv[1], &c->owner))
    return enif_make_badarg(env);
  enif_keep_resource(c);
  if (enif_thread_create((char *)"rgb_collector", &c->tid, collector_main, c, NULL)) {
    enif_release_resource(c);
    return mk_error(env, c, RGB_E_NOMEM);
  }
  c->collector_on = 1;
  return enif_make_atom(env, "ok");
}

/* stop_collector(Ctx) -> ok: ends the collector thread and drops the reference it held on the context (without
 * it the thread would keep the resource alive for ever and the destructor could never run) */
static ERL_NIF_TERM nif_stop_collector(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !c->collector_on) return enif_make_badarg(env);
  c->stop = 1;
  enif_thread_join(c->tid, NULL);
  c->collector_on = 0;
  c->stop = 0;
  return enif_make_atom(env, "ok");
}

/* snapshot(Ctx, NGroups) -> {ok, <<rgb_leaderboard_row x NGroups>>} */
static ERL_NIF_TERM nif_snapshot(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  nif_ctx *c; unsigned g; ErlNifBinary b;
  (void)argc;
  if (!get_ctx(env, argv[0], &c) || !enif_get_uint(env, argv[1], &g)) return enif_make_badarg(env);
  if (!enif_alloc_binary((size_t)g * sizeof(rgb_leaderboard_row), &b)) return mk_error(env, c, RGB_E_NOMEM);
  int rc = rgb_snapshot(c->ctx, (rgb_leaderboard_row *)b.data);
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
  {"submit", 3, nif_submit, 0},
  {"collect", 1, nif_collect, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"start_collector", 2, nif_start_collector, 0},
  {"stop_collector", 1, nif_stop_collector, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"snapshot", 2, nif_snapshot, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_checksums", 3, nif_wal_checksums, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_frame", 4, nif_wal_frame, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"wal_recover_check", 2, nif_wal_recover_check, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(ra_gpu_batch, nif_funcs, on_load, NULL, NULL, NULL)
