/*
 * rgb_comm.cpp -- the one collective of the path, behind the C ABI (include/ra_gpu_batch.h, "Multi-GPU"): the
 * all-gather of the per-GPU leaderboard / key-metrics shards (reference: src/ra_leaderboard.erl:18-26 is the node-wide
 * table every member's leader change lands in, src/ra.erl:1242-1270 reads the gauges per server; groups shard by
 * rgb_route() over the GPUs of the node, one rgb_ctx and one process or thread per GPU).  RCCL over xGMI:
 * ncclAllGather on the stream the caller gives -- the train launches' stream, so the gather is ordered behind the
 * snapshot kernel without an event.
 *
 * RCCL is bound at run time (dlopen), not at link time: the decision path has no collective, a single-GPU deployment
 * (and every test process) never loads the library, and inside a process that already carries an RCCL (PyTorch ships
 * its own librccl.so) the symbols of THAT copy are used -- two RCCL runtimes in one process do not share their
 * bootstrap state.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <new>

#include "../../include/ra_gpu_batch.h"

extern "C" void *rgb_ctx_stream(rgb_ctx *ctx);
extern "C" int rgb_ctx_device(rgb_ctx *ctx);
extern "C" int rgb_ctx_set_device(rgb_ctx *ctx);

namespace {

/* the slice of rccl.h this file uses (rccl/rccl.h: ncclUniqueId = 128 opaque bytes, ncclUint8 = 1, ncclSuccess = 0) */
struct nccl_unique_id { char internal[RGB_COMM_ID_BYTES]; };
typedef void *nccl_comm_t;
typedef int (*fn_get_unique_id)(nccl_unique_id *);
typedef int (*fn_comm_init_rank)(nccl_comm_t *, int, nccl_unique_id, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, nccl_comm_t, void *);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_comm_abort)(nccl_comm_t);

struct rccl_api {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
  fn_comm_abort comm_abort = nullptr;     /* optional */
  bool ok = false;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;
thread_local int g_last_rccl = 0;
thread_local const char *g_last_text = nullptr;   /* a failure that is not an RCCL code (timeout, a peer's error) */

void load_rccl() {
  /* a copy already in the process wins; then the loader's search path; then the ROCm install */
  void *h = dlsym(RTLD_DEFAULT, "ncclAllGather") ? RTLD_DEFAULT : nullptr;
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  g_rccl.comm_abort = (fn_comm_abort)dlsym(h, "ncclCommAbort");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather;
}

const rccl_api *rccl() {
  std::call_once(g_rccl_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}

}  // namespace

struct rgb_comm {
  nccl_comm_t comm = nullptr;
  rgb_ctx *ctx = nullptr;
  uint32_t n_ranks = 0, rank = 0;
};

extern "C" {

int rgb_comm_unique_id(void *id_out) {
  if (!id_out) return RGB_E_INVAL;
  const rccl_api *r = rccl();
  if (!r) return RGB_E_UNSUPPORTED;
  nccl_unique_id id;
  memset(&id, 0, sizeof id);
  g_last_rccl = r->get_unique_id(&id);
  if (g_last_rccl) return RGB_E_COMM;
  memcpy(id_out, &id, sizeof id);
  return RGB_OK;
}

int rgb_comm_init_rank(rgb_ctx *ctx, const void *id_in, uint32_t n_ranks, uint32_t rank, rgb_comm **out) {
  if (!ctx || !id_in || !out || n_ranks == 0 || rank >= n_ranks) return RGB_E_INVAL;
  *out = nullptr;
  const rccl_api *r = rccl();
  if (!r) return RGB_E_UNSUPPORTED;
  if (rgb_ctx_set_device(ctx) != RGB_OK) return RGB_E_HIP;        /* the communicator lives on the context's GPU */
  rgb_comm *c = new (std::nothrow) rgb_comm();
  if (!c) return RGB_E_NOMEM;
  nccl_unique_id id;
  memcpy(&id, id_in, sizeof id);
  g_last_rccl = r->comm_init_rank(&c->comm, (int)n_ranks, id, (int)rank);
  if (g_last_rccl) { delete c; return RGB_E_COMM; }
  c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
  *out = c;
  return RGB_OK;
}

void rgb_comm_destroy(rgb_comm *comm) {
  if (!comm) return;
  const rccl_api *r = rccl();
  if (r && comm->comm) (void)r->comm_destroy(comm->comm);
  delete comm;
}

uint32_t rgb_comm_n_ranks(const rgb_comm *comm) { return comm ? comm->n_ranks : 0; }
uint32_t rgb_comm_rank(const rgb_comm *comm) { return comm ? comm->rank : 0; }

int rgb_leaderboard_allgather(rgb_ctx *ctx, rgb_comm *comm, const void *d_rows_local, uint32_t n_rows, void *d_rows_all,
                              void *stream) {
  if (!ctx || !comm || comm->ctx != ctx || !d_rows_local || !d_rows_all) return RGB_E_INVAL;
  const rccl_api *r = rccl();
  if (!r) return RGB_E_UNSUPPORTED;
  if (rgb_ctx_set_device(ctx) != RGB_OK) return RGB_E_HIP;
  void *st = stream ? stream : rgb_ctx_stream(ctx);
  if (!comm->comm) { g_last_text = "the communicator was aborted"; return RGB_E_COMM; }
  g_last_text = nullptr;
  g_last_rccl = r->all_gather(d_rows_local, d_rows_all, (size_t)n_rows * sizeof(rgb_leaderboard_row), 1 /* ncclUint8 */,
                              comm->comm, st);
  return g_last_rccl ? RGB_E_COMM : RGB_OK;
}

/* internal (rgb_api.hip, rgb_leaderboard_allgather_host): `bytes` per rank of anything, same transport */
int rgb_comm_allgather_bytes(rgb_comm *comm, const void *d_local, uint64_t bytes, void *d_all, void *stream) {
  if (!comm || !d_local || !d_all) return RGB_E_INVAL;
  const rccl_api *r = rccl();
  if (!r) return RGB_E_UNSUPPORTED;
  if (!comm->comm) { g_last_text = "the communicator was aborted"; return RGB_E_COMM; }
  g_last_text = nullptr;
  g_last_rccl = r->all_gather(d_local, d_all, (size_t)bytes, 1 /* ncclUint8 */, comm->comm, stream);
  return g_last_rccl ? RGB_E_COMM : RGB_OK;
}
/* internal: give up on a collective that does not complete (a rank never arrived): ncclCommAbort frees the kernels
 * that spin on the missing peer; the communicator is unusable afterwards (rgb_comm_destroy + a new one) */
int rgb_comm_abort(rgb_comm *comm, const char *why) {
  if (!comm) return RGB_E_INVAL;
  const rccl_api *r = rccl();
  g_last_text = why;
  g_last_rccl = 0;
  if (r && r->comm_abort && comm->comm) { (void)r->comm_abort(comm->comm); comm->comm = nullptr; }
  return RGB_E_COMM;
}
void rgb_comm_set_error_text(const char *why) { g_last_text = why; g_last_rccl = 0; }

const char *rgb_comm_last_error(void) {
  if (g_last_text) return g_last_text;
  const rccl_api *r = rccl();
  return (r && r->error_string && g_last_rccl) ? r->error_string(g_last_rccl) : "";
}

}  /* extern "C" */
