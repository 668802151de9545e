/*
 * rgb_internal.h -- device-side layout shared by the kernels and the C-ABI implementation.
 *
 * Data layout in HBM ("line-granular SoA"): one array per access class, each row one or two
 * 64/128-byte lines so that a sparse gather of servers never fetches a line it does not use.
 *
 *   hot  [S][16] u64   128 B  every message reads it, most write it back
 *        0 current_term   1 commit_index   2 last_applied     3 last_index
 *        4 last_term      5 last_written_index               6 last_written_term
 *        7 packed (role, condition, slots, masks -- see PK_*)
 *        8 snapshot_index 9 snapshot_term  10 first_index
 *        11 last-run start index           12 last-run term
 *        13 pre_vote_token   14 machine_version | effective_machine_version << 32   15 spare
 *   peers[S][PS] u64   PS = roundup(3*N, 8): match_index[N] | next_index[N] | commit_index_sent[N]
 *        only leader-side messages touch it
 *   runs [S][K][2] u64 (start, term) of each term run of the ra_log range; only probed when an
 *        index older than the last run is looked up (log-matching repair)
 *   cond [S][4] u64    stored reply of await_condition (cold)
 */
#ifndef RGB_INTERNAL_H
#define RGB_INTERNAL_H

#include <stdint.h>
#include "../../include/ra_gpu_batch.h"

typedef unsigned long long u64;
typedef uint32_t u32;

#define RGB_HOT_WORDS 16

#define HOT_CT    0
#define HOT_CI    1
#define HOT_LA    2
#define HOT_LI    3
#define HOT_LT    4
#define HOT_LWI   5
#define HOT_LWT   6
#define HOT_PK    7
#define HOT_SI    8
#define HOT_ST    9
#define HOT_FIRST 10
#define HOT_LRS   11
#define HOT_LRT   12
#define HOT_TOKEN 13
#define HOT_MACVER 14

/* packed word: bit offset / width */
#define PK_ROLE_SH      0   /* 3 */
#define PK_COND_SH      3   /* 2 */
#define PK_SELF_SH      5   /* 4 */
#define PK_VOTES_SH     9   /* 4 */
#define PK_NRUNS_SH     13  /* 5 */
#define PK_NONVOTER_SH  18  /* 1 */
#define PK_VOTED_SH     20  /* 4, 0xF = undefined */
#define PK_LEADER_SH    24  /* 4, 0xF = undefined */
#define PK_CONDLDR_SH   28  /* 4, 0xF = undefined */
#define PK_PRESENT_SH   32  /* 8 */
#define PK_VOTER_SH     40  /* 8 */
#define PK_STATUS_SH    48  /* 8 */

static inline __host__ __device__ unsigned rgb_peer_stride(unsigned n_members) {
  return (3u * n_members + 7u) & ~7u;
}

struct rgb_dev {
  u64 *hot;
  u64 *peers;
  u64 *runs;
  u64 *cond;
  u32 n_servers;
  u32 n_members;
  u32 max_runs;
  u32 peer_stride;
  u32 max_pipeline_count;
  u32 max_aer_batch;
};

/* kernel launchers (rgb_kernels.hip) */
/* d_rpcs: n * max(N-1,1) fixed slots (message i owns slots [i*(N-1), (i+1)*(N-1))), or NULL */
/* d_n: optional device-resident message count (min(n, *d_n) messages are processed) */
int rgb_launch_tick(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, const u32 *d_n, rgb_decision *d_dec,
                    rgb_rpc *d_rpcs, u32 msg_index_base, void *stream);
/* d_scratch: 2*2*(RGB_MSG_KIND_MAX+1) u32 of device scratch */
int rgb_launch_synth(const rgb_dev &dev, u64 seed, u64 tick, rgb_msg *d_msgs, u32 *d_scratch,
                     u32 *d_kind_counts, u32 *d_n, void *stream);
int rgb_launch_pack(const rgb_dev &dev, const rgb_server_state *d_in, u32 first, u32 n, void *stream);
int rgb_launch_unpack(const rgb_dev &dev, rgb_server_state *d_out, u32 first, u32 n, void *stream);
int rgb_launch_leaderboard(const rgb_dev &dev, rgb_leaderboard_row *d_rows, void *stream);
int rgb_launch_checksum(const rgb_dev &dev, u32 first, u32 n, u64 *d_out, void *stream);

#endif
