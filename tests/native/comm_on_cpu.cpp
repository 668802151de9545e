/*
 * comm_on_cpu.cpp -- TEST INFRASTRUCTURE: the stand-in of ra_amd/csrc/rgb_comm.cpp in the emulated library.  Same
 * entry points and argument checks; where the product calls ncclAllGather on device memory, this hands the local
 * shard and the gathered buffer ("device memory" is host memory here) to a transport the test registers
 * (emu_comm_set_transport: tests/test_shard_gloo.py plugs in torch.distributed's gloo all-gather), so the N > 1 path
 * through the C entry point runs on CPUs.  One rank needs no transport.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ra_gpu_batch.h"

typedef int (*emu_transport_fn)(const void *local, uint64_t bytes, void *all, uint32_t n_ranks, uint32_t rank);
static emu_transport_fn g_transport = nullptr;

struct rgb_comm {
  rgb_ctx *ctx;
  uint32_t n_ranks, rank;
  unsigned char id[RGB_COMM_ID_BYTES];
  int aborted;                       /* rgb_comm_abort: the product's ncclCommAbort leaves the communicator unusable */
};

extern "C" {

void emu_comm_set_transport(emu_transport_fn fn) { g_transport = fn; }

int rgb_comm_unique_id(void *id_out) {
  if (!id_out) return RGB_E_INVAL;
  unsigned char *p = (unsigned char *)id_out;
  for (unsigned i = 0; i < RGB_COMM_ID_BYTES; ++i) p[i] = (unsigned char)(rand() & 0xFF);
  return RGB_OK;
}

int rgb_comm_init_rank(rgb_ctx *ctx, const void *id, uint32_t n_ranks, uint32_t rank, rgb_comm **out) {
  if (!ctx || !id || !out || n_ranks == 0 || rank >= n_ranks) return RGB_E_INVAL;
  rgb_comm *c = (rgb_comm *)calloc(1, sizeof(rgb_comm));
  if (!c) return RGB_E_NOMEM;
  c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
  memcpy(c->id, id, RGB_COMM_ID_BYTES);
  *out = c;
  return RGB_OK;
}

void rgb_comm_destroy(rgb_comm *comm) { free(comm); }
uint32_t rgb_comm_n_ranks(const rgb_comm *comm) { return comm ? comm->n_ranks : 0; }
uint32_t rgb_comm_rank(const rgb_comm *comm) { return comm ? comm->rank : 0; }
static const char *g_text = "";
const char *rgb_comm_last_error(void) { return g_text; }
void rgb_comm_set_error_text(const char *why) { g_text = why ? why : ""; }
int rgb_comm_abort(rgb_comm *comm, const char *why) { if (comm) comm->aborted = 1; g_text = why ? why : ""; return RGB_E_COMM; }
int rgb_comm_allgather_bytes(rgb_comm *comm, const void *d_local, uint64_t bytes, void *d_all, void *stream) {
  (void)stream;
  if (!comm || !d_local || !d_all) return RGB_E_INVAL;
  if (comm->aborted) { g_text = "the communicator was aborted"; return RGB_E_COMM; }
  if (comm->n_ranks == 1) { if (d_all != d_local) memmove(d_all, d_local, bytes); return RGB_OK; }
  if (!g_transport) return RGB_E_UNSUPPORTED;
  return g_transport(d_local, bytes, d_all, comm->n_ranks, comm->rank) ? RGB_E_COMM : RGB_OK;
}

int rgb_leaderboard_allgather(rgb_ctx *ctx, rgb_comm *comm, const void *d_rows_local, uint32_t n_rows, void *d_rows_all,
                              void *stream) {
  (void)stream;
  if (!ctx || !comm || comm->ctx != ctx || !d_rows_local || !d_rows_all) return RGB_E_INVAL;
  const uint64_t bytes = (uint64_t)n_rows * sizeof(rgb_leaderboard_row);
  if (comm->n_ranks == 1) {
    if (d_rows_all != d_rows_local) memmove(d_rows_all, d_rows_local, bytes);
    return RGB_OK;
  }
  if (!g_transport) return RGB_E_UNSUPPORTED;
  return g_transport(d_rows_local, bytes, d_rows_all, comm->n_ranks, comm->rank) ? RGB_E_COMM : RGB_OK;
}

}  /* extern "C" */
