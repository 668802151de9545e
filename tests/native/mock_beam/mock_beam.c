/* mock_beam.c -- a functional stand-in for the erl_nif subset of erl_nif.h in this directory (TEST
 * INFRASTRUCTURE: it exists so the NIF shim ra_amd/csrc/ra_gpu_batch_nif.c can be executed without OTP).
 * Terms are never freed (tests are short); resources are reference-counted for real. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "erl_nif.h"

enum { T_ATOM = 1, T_INT = 2, T_BIN = 3, T_TUPLE = 4, T_RES = 5, T_PID = 6, T_BADARG = 7 };
typedef struct term {
  int tag;
  char atom[40];
  uint64_t u; int64_t i; int is_signed;
  unsigned char *data; size_t size;
  int arity; struct term *elem[5];
  void *res;
} term;
typedef struct { long refc; ErlNifResourceDtor *dtor; ErlNifResourceType *type; } res_hdr;
struct enif_resource_type_t { ErlNifResourceDtor *dtor; };
struct enif_environment_t { int dummy; };

static term *mk(int tag) { term *t = (term *)calloc(1, sizeof *t); t->tag = tag; return t; }
#define TT(x) ((term *)(x))

ERL_NIF_TERM enif_make_atom(ErlNifEnv *e, const char *n) { term *t = mk(T_ATOM); strncpy(t->atom, n, sizeof t->atom - 1); return (ERL_NIF_TERM)t; }
static ERL_NIF_TERM tup(int n, ERL_NIF_TERM *v) { term *t = mk(T_TUPLE); t->arity = n; for (int k = 0; k < n; ++k) t->elem[k] = TT(v[k]); return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_tuple2(ErlNifEnv *e, ERL_NIF_TERM a, ERL_NIF_TERM b) { ERL_NIF_TERM v[] = {a, b}; return tup(2, v); }
ERL_NIF_TERM enif_make_tuple3(ErlNifEnv *e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c) { ERL_NIF_TERM v[] = {a, b, c}; return tup(3, v); }
ERL_NIF_TERM enif_make_tuple4(ErlNifEnv *e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c, ERL_NIF_TERM d) { ERL_NIF_TERM v[] = {a, b, c, d}; return tup(4, v); }
ERL_NIF_TERM enif_make_tuple5(ErlNifEnv *e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c, ERL_NIF_TERM d, ERL_NIF_TERM f) { ERL_NIF_TERM v[] = {a, b, c, d, f}; return tup(5, v); }
ERL_NIF_TERM enif_make_int(ErlNifEnv *e, int x) { term *t = mk(T_INT); t->i = x; t->u = (uint64_t)(int64_t)x; t->is_signed = 1; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_uint(ErlNifEnv *e, unsigned x) { term *t = mk(T_INT); t->u = x; t->i = x; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_uint64(ErlNifEnv *e, uint64_t x) { term *t = mk(T_INT); t->u = x; t->i = (int64_t)x; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *e) { return (ERL_NIF_TERM)mk(T_BADARG); }

/* binaries: enif_alloc_binary gives an owned buffer; enif_make_binary transfers it to the term */
int enif_alloc_binary(size_t n, ErlNifBinary *b) { b->data = (unsigned char *)malloc(n ? n : 1); b->size = n; b->ref_bin = b->data; return b->data != NULL; }
void enif_release_binary(ErlNifBinary *b) { free(b->ref_bin); b->data = NULL; b->ref_bin = NULL; b->size = 0; }
int enif_realloc_binary(ErlNifBinary *b, size_t n) {
  unsigned char *p = (unsigned char *)realloc(b->ref_bin, n ? n : 1);
  if (!p) return 0;
  b->data = p; b->ref_bin = p; b->size = n; return 1;
}
ERL_NIF_TERM enif_make_binary(ErlNifEnv *e, ErlNifBinary *b) { term *t = mk(T_BIN); t->data = b->data; t->size = b->size; b->ref_bin = NULL; return (ERL_NIF_TERM)t; }
/* a view into the parent's bytes (terms are never freed in the mock, so the parent outlives it) */
ERL_NIF_TERM enif_make_sub_binary(ErlNifEnv *e, ERL_NIF_TERM parent, size_t pos, size_t size) {
  term *t = mk(T_BIN); t->data = TT(parent)->data + pos; t->size = size; return (ERL_NIF_TERM)t;
}
int enif_inspect_binary(ErlNifEnv *e, ERL_NIF_TERM x, ErlNifBinary *b) {
  if (TT(x)->tag != T_BIN) return 0;
  b->data = TT(x)->data; b->size = TT(x)->size; b->ref_bin = NULL; return 1;
}
int enif_get_uint(ErlNifEnv *e, ERL_NIF_TERM x, unsigned *o) { term *t = TT(x); if (t->tag != T_INT || t->i < 0 || t->u > 0xFFFFFFFFull) return 0; *o = (unsigned)t->u; return 1; }
int enif_get_uint64(ErlNifEnv *e, ERL_NIF_TERM x, uint64_t *o) { term *t = TT(x); if (t->tag != T_INT || (t->is_signed && t->i < 0)) return 0; *o = t->u; return 1; }
int enif_get_int(ErlNifEnv *e, ERL_NIF_TERM x, int *o) { term *t = TT(x); if (t->tag != T_INT) return 0; *o = (int)t->i; return 1; }
int enif_get_local_pid(ErlNifEnv *e, ERL_NIF_TERM x, ErlNifPid *p) { if (TT(x)->tag != T_PID) return 0; p->pid = x; return 1; }
void *enif_alloc(size_t n) { return malloc(n); }
void enif_free(void *p) { free(p); }

/* resources */
static pthread_mutex_t res_mu = PTHREAD_MUTEX_INITIALIZER;
static long live_resources = 0, dtor_calls = 0;
ErlNifResourceType *enif_open_resource_type(ErlNifEnv *e, const char *m, const char *n, ErlNifResourceDtor *d, ErlNifResourceFlags f, ErlNifResourceFlags *tried) {
  ErlNifResourceType *t = (ErlNifResourceType *)calloc(1, sizeof *t); t->dtor = d; return t;
}
void *enif_alloc_resource(ErlNifResourceType *t, size_t n) {
  res_hdr *h = (res_hdr *)calloc(1, sizeof *h + n); h->refc = 1; h->dtor = t->dtor; h->type = t;
  pthread_mutex_lock(&res_mu); ++live_resources; pthread_mutex_unlock(&res_mu);
  return h + 1;
}
void enif_keep_resource(void *o) { pthread_mutex_lock(&res_mu); ++((res_hdr *)o - 1)->refc; pthread_mutex_unlock(&res_mu); }
void enif_release_resource(void *o) {
  res_hdr *h = (res_hdr *)o - 1;
  pthread_mutex_lock(&res_mu); long r = --h->refc; pthread_mutex_unlock(&res_mu);
  if (r == 0) {
    if (h->dtor) h->dtor(NULL, o);
    pthread_mutex_lock(&res_mu); --live_resources; ++dtor_calls; pthread_mutex_unlock(&res_mu);
    free(h);
  }
}
ERL_NIF_TERM enif_make_resource(ErlNifEnv *e, void *o) { term *t = mk(T_RES); t->res = o; enif_keep_resource(o); return (ERL_NIF_TERM)t; }
int enif_get_resource(ErlNifEnv *e, ERL_NIF_TERM x, ErlNifResourceType *ty, void **o) {
  if (TT(x)->tag != T_RES || TT(x)->res == NULL || ((res_hdr *)TT(x)->res - 1)->type != ty) return 0;
  *o = TT(x)->res; return 1;
}

/* environments, messages, threads */
ErlNifEnv *enif_alloc_env(void) { return (ErlNifEnv *)calloc(1, sizeof(ErlNifEnv)); }
void enif_free_env(ErlNifEnv *e) { free(e); }
void enif_clear_env(ErlNifEnv *e) {}
#define QCAP (1 << 20)   /* one owner per server at 65 536 x 5 sends 327 680 messages per batch */
static term *queue[QCAP]; static uint64_t queue_to[QCAP];
static int q_head = 0, q_tail = 0;
static pthread_mutex_t q_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t q_cv = PTHREAD_COND_INITIALIZER;
int enif_send(ErlNifEnv *caller, const ErlNifPid *to, ErlNifEnv *msg_env, ERL_NIF_TERM msg) {
  pthread_mutex_lock(&q_mu);
  int ok = (q_tail + 1) % QCAP != q_head;
  if (ok) { queue[q_tail] = TT(msg); queue_to[q_tail] = TT(to->pid)->u; q_tail = (q_tail + 1) % QCAP; pthread_cond_signal(&q_cv); }
  pthread_mutex_unlock(&q_mu);
  return ok;
}
struct ErlNifTid_ { pthread_t th; void *(*fn)(void *); void *arg; };
static __thread ErlNifTid tls_self = NULL;               /* threads the "BEAM" did not create have no tid */
static void *thread_tramp(void *p) { ErlNifTid t = (ErlNifTid)p; tls_self = t; return t->fn(t->arg); }
int enif_thread_create(char *name, ErlNifTid *tid, void *(*fn)(void *), void *arg, ErlNifThreadOpts *o) {
  *tid = (ErlNifTid)calloc(1, sizeof(struct ErlNifTid_));
  (*tid)->fn = fn; (*tid)->arg = arg;
  return pthread_create(&(*tid)->th, NULL, thread_tramp, *tid);
}
int enif_thread_join(ErlNifTid tid, void **ret) { int rc = pthread_join(tid->th, ret); free(tid); return rc; }
ErlNifTid enif_thread_self(void) { return tls_self; }
int enif_equal_tids(ErlNifTid a, ErlNifTid b) { return a == b; }
uint64_t enif_hash(ErlNifHash type, ERL_NIF_TERM t, uint64_t salt) {
  uint64_t x = TT(t)->u ^ salt;                          /* (the mock hashes pids only: what the shim asks for) */
  (void)type;
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33;
  return x;
}
int enif_compare_pids(const ErlNifPid *a, const ErlNifPid *b) {
  const uint64_t x = TT(a->pid)->u, y = TT(b->pid)->u;
  return x < y ? -1 : x > y;
}

struct ErlNifMutex_ { pthread_mutex_t mu; };
ErlNifMutex *enif_mutex_create(char *name) { ErlNifMutex *m = (ErlNifMutex *)calloc(1, sizeof *m); pthread_mutex_init(&m->mu, NULL); return m; }
void enif_mutex_destroy(ErlNifMutex *m) { pthread_mutex_destroy(&m->mu); free(m); }
void enif_mutex_lock(ErlNifMutex *m) { pthread_mutex_lock(&m->mu); }
void enif_mutex_unlock(ErlNifMutex *m) { pthread_mutex_unlock(&m->mu); }
/* the BEAM would run fp later on a dirty scheduler thread; the mock runs it at once and counts the reschedules */
static long dirty_reschedules = 0;
ERL_NIF_TERM enif_schedule_nif(ErlNifEnv *e, const char *fun_name, int flags,
                               ERL_NIF_TERM (*fp)(ErlNifEnv *, int, const ERL_NIF_TERM[]), int argc,
                               const ERL_NIF_TERM argv[]) {
  if (flags) __atomic_add_fetch(&dirty_reschedules, 1, __ATOMIC_RELAXED);
  return fp(e, argc, argv);
}
long mock_dirty_reschedules(void) { return __atomic_load_n(&dirty_reschedules, __ATOMIC_RELAXED); }

/* ---- the test driver's side (ctypes) ---- */
const ErlNifFunc *mock_nif_funcs(int *n);
int mock_nif_load(ErlNifEnv *env);
static ErlNifEnv call_env;
int mock_load(void) { return mock_nif_load(&call_env); }
ERL_NIF_TERM mock_call(const char *name, int argc, const ERL_NIF_TERM *argv) {
  int n = 0; const ErlNifFunc *f = mock_nif_funcs(&n);
  for (int k = 0; k < n; ++k)
    if (!strcmp(f[k].name, name) && (int)f[k].arity == argc) return f[k].fptr(&call_env, argc, argv);
  return 0;                                                    /* undef */
}
unsigned mock_func_flags(const char *name, int arity) {
  int n = 0; const ErlNifFunc *f = mock_nif_funcs(&n);
  for (int k = 0; k < n; ++k) if (!strcmp(f[k].name, name) && (int)f[k].arity == arity) return f[k].flags;
  return 0xFFFFFFFFu;
}
ERL_NIF_TERM mock_uint(uint64_t x) { return enif_make_uint64(NULL, x); }
ERL_NIF_TERM mock_int(int x) { return enif_make_int(NULL, x); }
ERL_NIF_TERM mock_atom(const char *n) { return enif_make_atom(NULL, n); }
ERL_NIF_TERM mock_pid(uint64_t id) { term *t = mk(T_PID); t->u = id; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM mock_binary(const void *p, size_t n) { term *t = mk(T_BIN); t->data = (unsigned char *)malloc(n ? n : 1); memcpy(t->data, p, n); t->size = n; return (ERL_NIF_TERM)t; }
int mock_tag(ERL_NIF_TERM x) { return x ? TT(x)->tag : 0; }
const char *mock_atom_name(ERL_NIF_TERM x) { return TT(x)->atom; }
int mock_arity(ERL_NIF_TERM x) { return TT(x)->arity; }
ERL_NIF_TERM mock_elem(ERL_NIF_TERM x, int k) { return (ERL_NIF_TERM)TT(x)->elem[k]; }
uint64_t mock_uint_value(ERL_NIF_TERM x) { return TT(x)->u; }
int64_t mock_int_value(ERL_NIF_TERM x) { return TT(x)->i; }
const void *mock_bin_data(ERL_NIF_TERM x) { return TT(x)->data; }
size_t mock_bin_size(ERL_NIF_TERM x) { return TT(x)->size; }
/* the garbage collector dropping the last reference a term held on a resource */
void mock_gc_resource_term(ERL_NIF_TERM x) { void *o = TT(x)->res; TT(x)->res = NULL; if (o) enif_release_resource(o); }
long mock_live_resources(void) { return live_resources; }
long mock_dtor_calls(void) { return dtor_calls; }
/* receive: the next message sent with enif_send, 0 after timeout_ms; *to = the pid it was sent to */
ERL_NIF_TERM mock_recv(int timeout_ms, uint64_t *to) {
  struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
  ts.tv_sec += timeout_ms / 1000; ts.tv_nsec += (long)(timeout_ms % 1000) * 1000000L;
  if (ts.tv_nsec >= 1000000000L) { ts.tv_sec += 1; ts.tv_nsec -= 1000000000L; }
  pthread_mutex_lock(&q_mu);
  while (q_head == q_tail)
    if (pthread_cond_timedwait(&q_cv, &q_mu, &ts)) break;
  term *m = NULL;
  if (q_head != q_tail) { m = queue[q_head]; if (to) *to = queue_to[q_head]; q_head = (q_head + 1) % QCAP; }
  pthread_mutex_unlock(&q_mu);
  return (ERL_NIF_TERM)m;
}
