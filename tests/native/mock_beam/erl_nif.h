/*
 * erl_nif.h -- FUNCTIONAL MOCK of the erl_nif subset ra_gpu_batch_nif.c uses (test infrastructure; this image
 * has no Erlang/OTP).  Same declarations as ra_amd/csrc/nif_stub/erl_nif.h, implemented by mock_beam.c: terms are
 * heap objects, binaries own their bytes, resources are reference-counted and run their destructor, enif_send
 * queues a deep copy of the message for mock_recv(), threads are pthreads.  It lets tests/test_nif_shim_mock_beam.py
 * EXECUTE the shim (argument unpacking, result packing, ownership, the collector thread) -- not the BEAM itself.
 */
#ifndef RGB_ERL_NIF_STUB_H
#define RGB_ERL_NIF_STUB_H
#include <stddef.h>
#include <stdint.h>

typedef uintptr_t ERL_NIF_TERM;
typedef struct enif_environment_t ErlNifEnv;
typedef struct { size_t size; unsigned char *data; void *ref_bin; void *spare[2]; } ErlNifBinary;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef void ErlNifResourceDtor(ErlNifEnv *, void *);
typedef struct { ERL_NIF_TERM pid; } ErlNifPid;
typedef struct ErlNifTid_ *ErlNifTid;
typedef struct ErlNifThreadOpts_ ErlNifThreadOpts;
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef struct {
  const char *name; unsigned arity;
  ERL_NIF_TERM (*fptr)(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]);
  unsigned flags;
} ErlNifFunc;
#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2

ERL_NIF_TERM enif_make_atom(ErlNifEnv *, const char *);
ERL_NIF_TERM enif_make_tuple2(ErlNifEnv *, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple3(ErlNifEnv *, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple4(ErlNifEnv *, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple5(ErlNifEnv *, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_int(ErlNifEnv *, int);
ERL_NIF_TERM enif_make_uint(ErlNifEnv *, unsigned);
ERL_NIF_TERM enif_make_uint64(ErlNifEnv *, uint64_t);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *);
ERL_NIF_TERM enif_make_binary(ErlNifEnv *, ErlNifBinary *);
ERL_NIF_TERM enif_make_resource(ErlNifEnv *, void *);
int enif_get_uint(ErlNifEnv *, ERL_NIF_TERM, unsigned *);
int enif_get_uint64(ErlNifEnv *, ERL_NIF_TERM, uint64_t *);
int enif_get_int(ErlNifEnv *, ERL_NIF_TERM, int *);
int enif_get_resource(ErlNifEnv *, ERL_NIF_TERM, ErlNifResourceType *, void **);
int enif_get_local_pid(ErlNifEnv *, ERL_NIF_TERM, ErlNifPid *);
int enif_inspect_binary(ErlNifEnv *, ERL_NIF_TERM, ErlNifBinary *);
int enif_alloc_binary(size_t, ErlNifBinary *);
void enif_release_binary(ErlNifBinary *);
int enif_realloc_binary(ErlNifBinary *, size_t);
void *enif_alloc(size_t);
void enif_free(void *);
void *enif_alloc_resource(ErlNifResourceType *, size_t);
void enif_release_resource(void *);
void enif_keep_resource(void *);
ErlNifResourceType *enif_open_resource_type(ErlNifEnv *, const char *, const char *, ErlNifResourceDtor *,
                                            ErlNifResourceFlags, ErlNifResourceFlags *);
ErlNifEnv *enif_alloc_env(void);
void enif_free_env(ErlNifEnv *);
void enif_clear_env(ErlNifEnv *);
int enif_send(ErlNifEnv *, const ErlNifPid *, ErlNifEnv *, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_sub_binary(ErlNifEnv *, ERL_NIF_TERM, size_t, size_t);
int enif_thread_create(char *, ErlNifTid *, void *(*)(void *), void *, ErlNifThreadOpts *);
int enif_thread_join(ErlNifTid, void **);
ErlNifTid enif_thread_self(void);
int enif_equal_tids(ErlNifTid, ErlNifTid);
int enif_compare_pids(const ErlNifPid *, const ErlNifPid *);
typedef enum { ERL_NIF_INTERNAL_HASH = 1, ERL_NIF_PHASH2 = 2 } ErlNifHash;
uint64_t enif_hash(ErlNifHash, ERL_NIF_TERM, uint64_t salt);
typedef struct ErlNifMutex_ ErlNifMutex;
ErlNifMutex *enif_mutex_create(char *name);
void enif_mutex_destroy(ErlNifMutex *);
void enif_mutex_lock(ErlNifMutex *);
void enif_mutex_unlock(ErlNifMutex *);
ERL_NIF_TERM enif_schedule_nif(ErlNifEnv *, const char *fun_name, int flags,
                               ERL_NIF_TERM (*fp)(ErlNifEnv *, int, const ERL_NIF_TERM[]), int argc,
                               const ERL_NIF_TERM argv[]);

#define ERL_NIF_INIT(MODULE, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                                   \
  const ErlNifFunc *mock_nif_funcs(int *n) { *n = (int)(sizeof(FUNCS) / sizeof(FUNCS[0])); return FUNCS; } \
  int mock_nif_load(ErlNifEnv *env) { void *priv = 0; return LOAD(env, &priv, 0); }
#endif
