/*
 * ra_oracle.h -- TEST INFRASTRUCTURE (see ra_oracle.c).  The CPU checker shares only the
 * boundary TYPES of include/ra_gpu_batch.h (messages, decisions, host state); none of the
 * product's code.
 */
#ifndef RA_ORACLE_H
#define RA_ORACLE_H
#include "../include/ra_gpu_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_ctx ora_ctx;

ora_ctx *ora_new(uint32_t n_groups, uint32_t n_members, uint32_t max_pipeline_count,
                 uint32_t max_aer_batch);
void     ora_free(ora_ctx *c);
uint32_t ora_n_servers(const ora_ctx *c);
int      ora_set_state(ora_ctx *c, uint32_t first, uint32_t n, const rgb_server_state *in);
int      ora_get_state(const ora_ctx *c, uint32_t first, uint32_t n, rgb_server_state *out);
/* sequential, in submission order (the reference's mailbox order) */
int      ora_step(ora_ctx *c, const rgb_msg *msgs, uint32_t n, rgb_decision *out,
                  rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs);
/* one tick with at most one message per server, OpenMP over messages (CPU baseline) */
int      ora_step_parallel(ora_ctx *c, const rgb_msg *msgs, uint32_t n, rgb_decision *out,
                           int n_threads, uint64_t *n_rpcs_total);
int      ora_max_threads(void);
/* bound the term-run table like the engine does (0 = unbounded, the default) */
void     ora_set_max_runs(ora_ctx *c, uint32_t max_runs);
/* the range list the RGB_MF_SEQX written events of the next steps name ((first, last) pairs; NULL / 0 = none) */
void     ora_set_seq_ranges(ora_ctx *c, const uint64_t *ranges, uint32_t n_ranges);
uint64_t ora_agreed_commit(const uint64_t *idxs, uint32_t n);
uint64_t ora_server_checksum(const rgb_server_state *h);
size_t   ora_struct_size(int which);

#ifdef __cplusplus
}
#endif
#endif
