/*
 * erl_nif.h -- COMPILE-CHECK STUB ONLY.  This image has no Erlang/OTP, so ra_gpu_batch_nif.c
 * cannot be built against the real erl_nif.h here.  This stub declares just the subset of the
 * public erl_nif API (OTP 26 signatures) the shim uses so that `make nif-check` can run
 * gcc -fsyntax-only on it.  It is never linked or shipped; on a machine with OTP, build the NIF
 * with -I"$(erl -noshell -eval 'io:format("~s",[code:root_dir()])' -s init stop)/usr/include".
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

#define ERL_NIF_INIT(MODULE, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD) \
  const ErlNifFunc *rgb_stub_nif_init_##MODULE(void) { (void)LOAD; return FUNCS; }
#endif
