/*
 * rgb_internal.h -- device-side layout shared by the kernels and the C-ABI implementation.
 *
 * Data layout in HBM ("line-granular SoA"): one array per access class, each row one or two
 * 64/128-byte lines so that a sparse gather of servers never fetches a line it does not use.
 *
 *   hot  [S][16] u64   128 B  every message reads it, most write it back
 *        0 current_term   1 packed (role, condition, slots, masks -- see PK_*)
 *        2 last-run term  3 start of the run before the last (mirrors of the run table: a term lookup touches
 *          memory only below the newest two runs)
 *        4 commit_index   5 last_applied     6 last_index       7 last_term
 *        8 snapshot_index 9 snapshot_term  10 first_index      11 last-run start index
 *        12 last_written_index             13 last_written_term
 *        14 term of the run before the last 15 first pending index (ra_log `pending` = [this .. last_index])
 *   peers[S][PS] u64   PS = roundup(3*N, 8): (match_index, next_index) x N | commit_index_sent[N]
 *        only leader-side messages touch it
 *   runs [S][K][2] u64 (start, term) of each term run of the ra_log range; only probed when an
 *        index older than the last run is looked up (log-matching repair)
 *   cond [S][4] u64    stored reply of await_condition (cold)
 *   qry  [S][16] u64   consistent-query heartbeats (cold): 0 query_index, 1+i query_index of peer slot i,
 *        9 snapshot_backoff mask, 10 pre_vote_token, 11 machine versions (election kinds only),
 *        12..15 the two older ranges of a sparse `pending` (valid while packed-word bit 59 is set)
 */
#ifndef RGB_INTERNAL_H
#define RGB_INTERNAL_H

#include <stdint.h>
#include "../../include/ra_gpu_batch.h"

typedef unsigned long long u64;
typedef uint32_t u32;

#define RGB_HOT_WORDS 16
#define RGB_QRY_WORDS 16

/* 16-byte pieces hold what changes together, so a message dirties as few pieces as possible:
 * (term, packed) votes/roles | (commit, applied) every commit advance | (last index, last term)
 * appends | (last written index, term) written events | snapshot | range start, last-run start |
 * last-run term, pre-vote token | machine versions, first pending index */
/* Word order chosen by what a steady-state message WRITES: a follower appending dirties (commit_index, last_applied)
 * and (last_index, last_term) -- pieces 2 and 3, ONE aligned 32-byte sector; a written event dirties (last written
 * index, term) and the first pending index -- pieces 6 and 7, one sector; the write-back of a dirty line goes out in
 * 32-byte sectors and a tick's time follows its written bytes (round 4: the earlier order -- commit / last index in
 * pieces 1 and 2, last written in piece 3 -- made each of them two sectors). */
#define HOT_CT    0
#define HOT_PK    1
#define HOT_LRT   2    /* last-run term | (start of run n_runs-2: a mirror of the run table, like HOT_LRS of the last run) */
#define HOT_PRS   3
#define HOT_CI    4
#define HOT_LA    5
#define HOT_LI    6
#define HOT_LT    7
#define HOT_SI    8
#define HOT_ST    9
#define HOT_FIRST 10
#define HOT_LRS   11
#define HOT_LWI   12
#define HOT_LWT   13
#define HOT_PRT   14   /* term of run n_runs-2 */
#define HOT_PEND  15
/* the 16-byte piece (pair of words) that holds ..: the explicit piece code of the kernels goes through these */
#define HOT_P_TERM  0  /* (current_term, packed)          */
#define HOT_P_LRT   1  /* (last-run term, prev-run start) */
#define HOT_P_CI    2  /* (commit_index, last_applied)    */
#define HOT_P_LI    3  /* (last_index, last_term)         */
#define HOT_P_SI    4  /* (snapshot index, term)          */
#define HOT_P_FIRST 5  /* (first_index, last-run start)   */
#define HOT_P_LW    6  /* (last written index, term)      */
#define HOT_P_PEND  7  /* (prev-run term, first pending)  */
/* peers row: (match_index, next_index) of member i side by side -- a counted reply dirties one 16-byte piece -- then
 * the commit_index_sent words */
#define PEER_MI(i, N) (2u * (unsigned)(i))
#define PEER_NI(i, N) (2u * (unsigned)(i) + 1u)
#define PEER_CS(i, N) (2u * (unsigned)(N) + (unsigned)(i))

/* packed word: bit offset / width */
#define PK_ROLE_SH      0   /* 3 */
#define PK_COND_SH      3   /* 2 */
#define PK_SELF_SH      5   /* 4 */
#define PK_VOTES_SH     9   /* 4 */
#define PK_NRUNS_SH     13  /* 5 */
#define PK_NONVOTER_SH  18  /* 1 */
#define PK_CONDTO_SH    19  /* 1: the stored condition's transition_to is leader (RGB_COND_WAL_DOWN_LEADER = COND 3 + this bit) */
#define PK_VOTED_SH     20  /* 4, 0xF = undefined */
#define PK_LEADER_SH    24  /* 4, 0xF = undefined */
#define PK_CONDLDR_SH   28  /* 4, 0xF = undefined */
#define PK_PRESENT_SH   32  /* 8 */
#define PK_VOTER_SH     40  /* 8 */
#define PK_STATUS_SH    48  /* 8 */
#define PK_QSELF_SH     56  /* 1: query_index > 0 (qry row word 0 is worth reading)      */
#define PK_QPEER_SH     57  /* 1: some peer query_index > 0 (reset_query_index has work) */
#define PK_BACKOFF_SH   58  /* 1: some peer is in {snapshot_backoff,_}: qry row word QRY_BACKOFF holds the mask */
#define PK_PENDX_SH     59  /* 1: `pending` has ranges below its newest one: qry row words QRY_PEND_LO.. hold them */
#define QRY_PEND_LO     12  /* (first, last) of the lower old range, (1, 0) when there is only one */
#define QRY_PEND_HI     14  /* (first, last) of the old range next below the newest range [HOT_PEND .. last_index] */
#define QRY_BACKOFF     9   /* qry row: word 0 query_index, 1..8 peer query_index, 9 backoff mask */
#define QRY_TOKEN       10  /* pre_vote_token (election kinds only)                                   */
#define QRY_MACVER      11  /* machine_version | effective_machine_version << 32                      */

/* Device order of a tick: clause family = (class rank of the message kind, success flag).  Every
 * kind is its own kernel class: a wavefront of the class-dispatch kernel runs the code path
 * specialised (compile-time kind) for its 64-message slice. */
#define RGB_N_CLASSES 15
static inline __host__ __device__ unsigned rgb_kind_rank(unsigned kind) {
  switch (kind) {
    case RGB_MSG_AER: return 0;
    case RGB_MSG_AER_REPLY: return 1;
    case RGB_MSG_WRITTEN: return 2;
    case RGB_MSG_APPEND: return 3;
    case RGB_MSG_PIPELINE_RPCS: return 4;
    case RGB_MSG_REQUEST_VOTE: return 5;
    case RGB_MSG_VOTE_RESULT: return 6;
    case RGB_MSG_AWAIT_TIMEOUT: return 7;
    case RGB_MSG_ELECTION_TIMEOUT: return 8;
    case RGB_MSG_PRE_VOTE_RPC: return 9;
    case RGB_MSG_PRE_VOTE_RESULT: return 10;
    case RGB_MSG_SNAPSHOT_WRITTEN: return 11;
    case RGB_MSG_HEARTBEAT_RPC: return 12;
    case RGB_MSG_HEARTBEAT_REPLY: return 13;
    case RGB_MSG_CONSISTENT_QUERY: return 14;
    default: return 15;   /* NOP */
  }
}
#define RGB_N_FAMILIES 32
static inline __host__ __device__ unsigned rgb_family(unsigned kind, unsigned flags) {
  return 2u * rgb_kind_rank(kind) + ((flags & RGB_MF_SUCCESS) ? 1u : 0u);
}
static inline unsigned rgb_class_of_kind(unsigned kind) { return rgb_kind_rank(kind); }   /* NOP has no class */

/* ---- train launches: several ticks in one launch (rgb_train_kernel) ----
 * Servers are sharded by group: shard = group mod RGB_TRAIN_SHARDS (= the XCDs of the device); a train tick is
 * ordered by (class, shard, success flag): see rgb_bucket. */
#define RGB_TRAIN_SHARDS 8u
#define RGB_N_BUCKETS ((RGB_N_CLASSES + 1u) * RGB_TRAIN_SHARDS * 2u)   /* 256 */
/* RGB_TRAIN_ERR_*: include/ra_gpu_batch.h */
/* control words of a train launch: 0 sticky error flags | 1..7 calibration scratch | 8..15 blocks arrived per XCC |
 * 32 (1 + x): the ticket counter of shard x, one 128-byte line each (rgb_train_kernel) */
#define RGB_TRAIN_CTL_WORDS (320u + 32u * 64u)   /* .. | 320 + 32 j, j < 64: the rotation marks of a dealt launch */
static inline __host__ __device__ unsigned rgb_shard_of_server(unsigned server, unsigned n_members) {
  return (server / n_members) & (RGB_TRAIN_SHARDS - 1u);
}
/* bucket = (class rank (15 = NOP), shard, success flag): class-major, so a family-ordered consumer still finds every
 * class contiguous, and (class, shard) is one contiguous range */
static inline __host__ __device__ unsigned rgb_bucket(unsigned kind, unsigned flags, unsigned server, unsigned n_members) {
  return (rgb_kind_rank(kind) * RGB_TRAIN_SHARDS + rgb_shard_of_server(server, n_members)) * 2u +
         ((flags & RGB_MF_SUCCESS) ? 1u : 0u);
}
/* The same with the producer's HINT for the kinds that carry no success flag: off_steady = "the owning gen_statem is
 * not in the state this kind's steady-state outcome needs" (a written event of a server that is leader, an
 * append_entries_rpc for a server that is not follower -- ra_server_proc knows its own state name, the device-side
 * generator reads the role).  The hint only ORDERS the tick: hinted messages sit in sub-bucket 1 of their (class,
 * shard), so the wavefronts of sub-bucket 0 are steady-state lanes only and skip the general clause code.  A wrong or
 * absent hint costs time, never a result: nothing in the kernels reads the sub-bucket. */
static inline __host__ __device__ unsigned rgb_bucket_hinted(unsigned kind, unsigned flags, unsigned server,
                                                             unsigned n_members, bool off_steady) {
  const bool hinted_kind = kind == RGB_MSG_WRITTEN || kind == RGB_MSG_AER;
  return rgb_bucket(kind, flags, server, n_members) | ((hinted_kind && off_steady) ? 1u : 0u);
}
/* position of a server's sequence byte: the bytes of one shard are contiguous, so an XCD's L2 never holds a line of
 * the array that another XCD writes */
/* RGB_SEQ_SPREAD: bytes between two servers' sequence bytes (1 = packed, the product).  A tick's wavefronts publish with
 * 214 k one-byte stores onto the array's lines, 84 per line and tick when packed; spreading them (A/B builds of round 5:
 * 4 -> +3 % per tick, 16 -> +12 %) is worse -- a wavefront's polls and stores then touch more lines, and it is the lines
 * a wavefront touches that cost (profiles/EXPERIMENTS.md, round 5) */
#ifndef RGB_SEQ_SPREAD
#define RGB_SEQ_SPREAD 1u
#endif
static inline __host__ __device__ unsigned rgb_seq_index(unsigned server, unsigned n_members, unsigned seq_stride) {
  const unsigned g = server / n_members, m = server - g * n_members;
  return (g & (RGB_TRAIN_SHARDS - 1u)) * seq_stride + ((g / RGB_TRAIN_SHARDS) * n_members + m) * RGB_SEQ_SPREAD;
}
#define RGB_TRAIN_MAX_TICKS 255u   /* ticks per launch: the values a sequence byte takes within one launch are distinct */
/* one tick of a train: rows of RGB_TRAIN_SHARDS blocks; row r of class c serves slice r of every shard (the row table
 * of rgb_train_make_tick says which (class, row) a block row of the tick is) */
/* A class's two sub-buckets (bucket bit 0: the success flag of a reply, the producer's steady-state hint for the other
 * kinds -- rgb_bucket_hinted) are separate row sets, "plan classes" pc = 2 x class + sub: a slice never straddles them,
 * and each is spread over the tick by its own group order (a sub-bucket that simply sat behind the other one inside
 * the class's rows would meet its servers' next messages less than a tick later). */
#define RGB_N_PCLASSES (2u * RGB_N_CLASSES)
/* Plan class RGB_PC_SNAP: the rows of a LEADERBOARD SNAPSHOT taken in front of the tick, inside the launch (row j =
 * groups 64 j .. 64 j + 63 of every shard) -- rgb_train_snap_slice.  To the sequence bytes a snapshot is one more
 * message to every server: its rows wait until every member of their groups has applied what came before the
 * boundary, read the rows, and advance the bytes; the tick's own messages carry stamps one higher. */
#define RGB_PC_SNAP RGB_N_PCLASSES
#define RGB_SNAP_LEAD 0.5f          /* ticks: in front of every message class's lead, so that a snapshot row is
                                       dispatched before the tick's messages to its groups (waits only point back) */
struct rgb_train_tick {
  u32 n_rows;
  u32 msg_base;                                  /* first message of the tick when the launch has no tick stride (rgb_submit) */
  u32 snap;                                      /* 1 + ordinal of the snapshot in front of this tick, 0 = none */
  u32 pad[13];
  u32 off[RGB_N_PCLASSES][RGB_TRAIN_SHARDS];    /* first message of (plan class, shard)  */
  u32 cnt[RGB_N_PCLASSES][RGB_TRAIN_SHARDS];    /* its message count                     */
};
/* the load generator's scratch: RGB_SYNTH_FIXED_WORDS (family totals -- what rgb_tick_classes_kernel reads -- |
 * bucket totals | bucket bases) + RGB_N_BUCKETS words per generator block (64 groups): rgb_synth_scratch_words() */
#define RGB_SYNTH_FIXED_WORDS (RGB_N_FAMILIES + 2u * RGB_N_BUCKETS)
u32 rgb_synth_scratch_words(u32 n_groups);

static inline __host__ __device__ unsigned rgb_peer_stride(unsigned n_members) {
  return (3u * n_members + 7u) & ~7u;
}

struct rgb_dev {
  u64 *hot;
  u64 *peers;
  u64 *runs;
  u64 *cond;
  u64 *qry;
  u32 n_servers;
  u32 n_members;
  u32 max_runs;
  u32 peer_stride;
  u32 max_pipeline_count;
  u32 max_aer_batch;
  unsigned char *seq;   /* train launches: per-server sequence byte (messages applied, mod 256), shard-major:
                           rgb_seq_index().  Only rgb_train_kernel reads or writes it; never reset */
  u32 seq_stride;       /* bytes per shard of seq */
  const u64 *seq_ranges;  /* RGB_MF_SEQX: (first, last) pairs of the launch's range list (rgb_submit_seq: the slot's;
                             rgb_set_seq_ranges_device: the caller's), or null */
  u32 n_seq_ranges;
  u32 fuse_pipeline;  /* RGB_CFG_FUSE_PIPELINE: a leader's success reply / written event emits its pipeline_rpcs event's rpcs */
  u32 synth_hint;  /* the load generator's bucketing hint (rgb_synth_set_hint): 0 none, 1 the owner's state name,
                      2 (default) + the O(1) header compare an owner can make against the fields it holds */
  u32 dbg;   /* always 0 in the product library.  The -DRGB_PROFILE build (libra_gpu_batch_prof.so, tools/ only)
                reads RGB_DEBUG: 1 = no state write-back, 2 = no decision store, 8 = no hot-line load (zero
                state), 16 = per-wave timestamps into dbg_buf; all but 16 break parity */
  u64 *dbg_buf;
};

/* kernel launchers (rgb_kernels.hip) */
/* d_rpcs: n * max(N-1,1) fixed slots (message i owns slots [i*(N-1), (i+1)*(N-1))), or NULL */
/* d_n: optional device-resident message count (min(n, *d_n) messages are processed).
 * cls: -1 = generic kernel (any kinds), 0..3 = the kernel specialised for that class's kind (every
 * message of the slice must have it), RGB_TICK_CLS_WRITTEN_SEQX = written events that may carry a range list,
 * RGB_TICK_CLS_NOP = NOP slots only.  rpc_slot_base: fixed-slot index of the slice's message 0. */
#define RGB_TICK_CLS_WRITTEN_SEQX 32
#define RGB_TICK_CLS_NOP 33
int rgb_launch_tick(const rgb_dev &dev, int cls, const rgb_msg *d_msgs, u32 n, const u32 *d_n,
                    rgb_decision *d_dec, rgb_rpc *d_rpcs, u32 rpc_slot_base, u32 msg_index_base, void *stream);
/* family-ordered tick, ONE launch: counts[] (host) or d_family_totals (device, RGB_N_FAMILIES u32,
 * with max_msgs bounding the grid) give the class sizes */
int rgb_launch_tick_classes(const rgb_dev &dev, const rgb_msg *d_msgs, const u32 counts[RGB_N_CLASSES],
                            const u32 *d_family_totals, u32 max_msgs, rgb_decision *d_dec, rgb_rpc *d_rpcs,
                            u32 rpc_slot_base, u32 msg_index_base, void *stream);
/* d_scratch: 2*RGB_N_FAMILIES u32 of device scratch */
/* d_scratch: rgb_synth_scratch_words(groups) u32; d_bucket_counts (may be NULL): RGB_N_BUCKETS u32 of this tick */
/* d_stamps (may be NULL): one byte per message slot = the generator's count d_sent[] (one byte per server, laid out
 * like dev.seq, advanced here) of the messages it addressed to that server before: a train's sequence stamps */
int rgb_launch_synth(const rgb_dev &dev, u64 seed, u64 tick, rgb_msg *d_msgs, u32 *d_scratch,
                     u32 *d_kind_counts, u32 *d_n, u32 *d_bucket_counts, unsigned char *d_stamps, unsigned char *d_sent,
                     void *stream);
/* ticks [0, n_ticks) of d_plan in one launch (n_ticks <= RGB_TRAIN_MAX_TICKS) of n_blocks persistent blocks
 * (rgb_train_resident_blocks) on a device of n_xcc XCCs (1, 2, 4 or 8: a block serves the shard of the XCC it runs
 * on); bpt = RGB_TRAIN_SHARDS x the rows per tick of d_row_tab; tick_stride = 0: tick t starts at d_plan[t].msg_base
 * and a message's rpc slots follow its index in the whole buffer (the sub-tick rounds of one rgb_submit); d_stamps:
 * one byte per message, laid out like d_msgs; d_rpcs (may be NULL): rpc_ring tick-sized regions, tick t uses region
 * t mod rpc_ring; rgb_rpc.msg_index = index_base + t * tick_stride + i; d_ctl: RGB_TRAIN_CTL_WORDS words, word 0 =
 * sticky error flags, the rest per-launch counters (zeroed here) */
int rgb_launch_train(const rgb_dev &dev, const rgb_msg *d_msgs, const unsigned char *d_stamps, u32 tick_stride,
                     const rgb_train_tick *d_plan, const u32 *d_row_tab, u32 n_ticks, u32 bpt, rgb_decision *d_dec,
                     rgb_rpc *d_rpcs, u32 rpc_ring, u32 index_base, u32 *d_ctl, u32 n_xcc, u32 n_blocks, void *stream,
                     const unsigned char *d_snap_stamps = nullptr, rgb_leaderboard_row *d_snap_rows = nullptr,
                     u32 tab_rpt = 0 /* rows per tick of d_row_tab when it is more than bpt / RGB_TRAIN_SHARDS */);
u32 rgb_train_resident_blocks(unsigned n_members);
/* the placement marks of the last dealt launch on d_ctl -> RGB_TRAIN_ERR_PLACEMENT in d_ctl[0] (the next launch does
 * this by itself; the host calls it before it reads the error word) */
int rgb_launch_train_verify(u32 *d_ctl, void *stream);
/* stamps of the n messages of one tick from the running counters d_seq_cnt (ticks in train order) */
int rgb_launch_train_seq(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_seq_cnt,
                         unsigned char *d_stamps, void *stream);
/* d_out: one u32, zeroed by the caller: bit k = a block of the launch ran on XCC k */
int rgb_launch_train_calibrate(u32 *d_out, void *stream);
/* host: the plan of one tick from its bucket counts (uint32[RGB_N_BUCKETS]); returns the tick's rows; row_tab (may be
 * NULL: count only) receives them when they fit row_cap */
u32 rgb_train_make_tick(const u32 *bucket_counts, unsigned n_members, rgb_train_tick *out, u32 *row_tab, u32 row_cap,
                        u32 snap_rows = 0);
/* the same on the device, one block per tick: ticks [first_tick, first_tick + n_ticks) of d_ticks / d_rows (rpt rows of
 * table per tick) from d_bucket_counts[n_ticks][RGB_N_BUCKETS]; d_err: the sticky error word (RGB_TRAIN_ERR_PLAN) */
int rgb_launch_train_plan(const u32 *d_bucket_counts, rgb_train_tick *d_ticks, u32 *d_rows, u32 rpt, u32 first_tick,
                          u32 n_ticks, u32 snapshot_every, u32 n_groups, u32 n_members, u32 *d_err, void *stream);
/* rows per tick a table must hold for ANY tick of n_servers servers (at most one message per server) */
u32 rgb_train_rows_bound(u32 n_servers, u32 n_members, bool with_snapshot);
/* rows of a snapshot: 64 groups of every shard per row */
static inline u32 rgb_train_snap_rows(u32 n_groups) {
  return ((n_groups + RGB_TRAIN_SHARDS - 1u) / RGB_TRAIN_SHARDS + 63u) / 64u;
}
/* every sequence byte + 1 (a snapshot outside a launch: rgb_snapshot_train_device); the generator's marks
 * (rgb_synth_snapshot_mark_device): out (may be NULL) = sent, sent += 1 */
int rgb_launch_seq_bump(unsigned char *d_seq, unsigned char *d_out, u32 n_bytes, void *stream);
int rgb_launch_pack(const rgb_dev &dev, const rgb_server_state *d_in, u32 first, u32 n, void *stream);
int rgb_launch_unpack(const rgb_dev &dev, rgb_server_state *d_out, u32 first, u32 n, void *stream);
int rgb_launch_leaderboard(const rgb_dev &dev, rgb_leaderboard_row *d_rows, void *stream);
int rgb_launch_checksum(const rgb_dev &dev, u32 first, u32 n, u64 *d_out, void *stream);
/* rgb_submit: stamps = sequence byte before the launch + the round the host wrote into d_stamps */
int rgb_launch_stamp_rounds(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_stamps, void *stream);
/* a batch's results written by the device into the slot's pinned host buffers: decisions expanded and in submission
 * order (d_pos[i] = device position of submitted message i), rpc records compacted in (message, slot) order with msg_index = the submission index, header = {records,
 * train error word, over-count flag}; d_scratch = rgb_results_blocks(cap) + 1 words, ZERO when allocated */
u32 rgb_results_blocks(u32 n);
int rgb_launch_results(const rgb_decision *d_dec, const u32 *d_pos, u32 n, u32 cap, const rgb_rpc *d_rpcs, u32 rpc_stride, u32 *d_scratch,
                       const u32 *d_ctl, rgb_decision *out_dec, rgb_rpc *out_rpcs, u32 *out_hdr, void *stream);
/* undo log: rgb_undo_pieces(dev) 16-byte pieces per server (every row + the sequence byte) of the n servers d_ids
 * name, saved to (restore = 0) or written back from (restore = 1) d_undo */
u32 rgb_undo_pieces(const rgb_dev &dev);
int rgb_launch_undo(const rgb_dev &dev, const u32 *d_ids, u32 n, void *d_undo, u32 restore, void *stream);

#endif
