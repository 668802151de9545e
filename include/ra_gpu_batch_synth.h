/*
 * ra_gpu_batch_synth.h -- device-side synthetic load generator for libra_gpu_batch: what
 * ra_bench (reference src/ra_bench.erl) is to Ra.  Benchmark tooling, not part of the drop-in
 * boundary: it reads the CURRENT device state of every group and writes one dense tick of
 * messages (at most one per server) straight into HBM, so long message streams can be produced
 * without a host round trip per tick.
 *
 * The tick is written COMPACTED (no empty slots) and ordered by (class of the message kind, group mod 8, success
 * flag) = rgb_train_bucket: every kernel class is contiguous (what the per-tick class kernel needs, like the
 * family order rgb_submit gives host batches) and so is every (class, shard) pair (what a train launch needs); its
 * message count goes to *d_n.
 *
 * Per group and tick (a fully loaded node):
 *   leader      append_entries_reply ok 75 % / failed 5 % / {commands,_} append of 1..4 entries
 *               20 %; a {written,..} event instead with probability 1/4 while it has unwritten
 *               entries; a noop append (Force) right after winning an election
 *   follower    append_entries_rpc with probability 1/2 (80 % append at the tail, 5 % empty
 *               heartbeat, 5 % gap -> missing, 5 % wrong prev_log_term -> term_mismatch,
 *               5 % overlapping resend), else {written,..} for its unwritten entries (80 %)
 *   queries     2 % of the leader's turns are a consistent_query (query_index + 1, heartbeats
 *               requested); while a follower has not confirmed the leader's query_index it
 *               receives the #heartbeat_rpc{} with probability 1/8, and the leader the
 *               #heartbeat_reply{} of the first such follower with probability 1/3
 *   term churn  5 % of the groups: one member gets a request_vote_rpc with term+1
 *   elections   a leader that is behind a member's term receives that member's failed reply
 *               (steps down); a leaderless group runs election_timeout -> pre_vote_result x
 *               quorum -> request_vote_result x quorum on its most advanced member
 */
#ifndef RA_GPU_BATCH_SYNTH_H
#define RA_GPU_BATCH_SYNTH_H
#include "ra_gpu_batch.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Write tick `tick` (at most n_groups*n_members rgb_msg) to d_msgs from the current device state.
 * d_kind_counts (may be NULL): uint32[RGB_MSG_KIND_MAX+1], incremented per generated message
 * kind; d_n (may be NULL): uint32 message count of the tick.  Enqueued on `stream` (NULL = the
 * context's stream); no synchronisation. */
int rgb_synth_tick_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs,
                          void *d_kind_counts, void *d_n, void *stream);
/* the same; d_bucket_counts (may be NULL): uint32[RGB_TRAIN_BUCKETS] = this tick's messages per train bucket
 * (rgb_train_bucket) -- what rgb_train_plan_create wants.  Ticks are written in bucket order. */
int rgb_synth_tick_buckets_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs,
                                  void *d_kind_counts, void *d_n, void *d_bucket_counts, void *stream);

/* the same, plus the TRAIN STAMPS of the tick: d_stamps (uint8 per message slot, laid out like d_msgs) = the value of
 * the server's sequence byte every message must find in a train launch (ra_gpu_batch.h, "Train launches").  A producer
 * knows how many messages it has addressed to a server: the generator keeps that count per server (mod 256, device
 * memory of the context) and stamps as it writes -- no pass over the stream afterwards (rgb_train_stamp_device is
 * that pass, for streams whose producer does not stamp).  The count starts from what the servers' sequence bytes
 * hold when the first stamped tick is generated; rgb_synth_stamps_resync_device re-reads them (after trains of a
 * stream the generator did not produce, or none of its own: the stamps of a stream are valid for a replay that
 * starts from the same sequence bytes -- per-tick launches and rgb_upload_state do not move them). */
int rgb_synth_tick_stamped_device(rgb_ctx *ctx, uint64_t seed, uint64_t tick, void *d_msgs,
                                  void *d_kind_counts, void *d_n, void *d_bucket_counts, void *d_stamps,
                                  void *stream);
int rgb_synth_stamps_resync_device(rgb_ctx *ctx, void *stream);
/* a leaderboard snapshot boundary in a stamped stream (ra_gpu_batch.h, "Leaderboard snapshots inside a train"): the
 * producer's counts go to d_snap_stamps (uint8[rgb_train_seq_bytes()], may be NULL) -- what every server's sequence
 * byte must show at the boundary -- and advance by one; call it between the ticks the boundary separates */
int rgb_synth_snapshot_mark_device(rgb_ctx *ctx, void *d_snap_stamps, void *stream);

/* The generator's ORDERING HINT for the kinds that carry no success flag (rgb_bucket_hinted, ra_gpu_batch.h "Train
 * launches"): which messages it sorts into sub-bucket 1 of their (class, shard).  0: none.  1: the owner's gen_statem
 * state name (an append_entries_rpc / written event whose owner is not in state follower; a written event whose owner
 * is leader).  2 (default): + what the owner sees by comparing the rpc's HEADER with fields it holds anyway -- an
 * append_entries_rpc whose term is not the owner's current_term, whose sender is not its leader_id, or whose
 * (prev_log_index, prev_log_term) is not ra_log's last (index, term), or whose entries open a new term; a written event
 * of a server that knows no leader.  All O(1) on the ra_server_state() map ra_server_proc holds at the interception
 * point; the generator reads the same fields from the device rows.  The hint only ORDERS a tick: the
 * kernels re-check every condition, a wrong or absent hint costs time, never a result. */
int rgb_synth_set_hint(rgb_ctx *ctx, uint32_t level);

/* Apply the tick that rgb_synth_tick_device just wrote (same stream): one launch of the
 * class-dispatch kernel sized from the family totals the generator left in device memory, so no
 * host round trip is needed between generating a tick and applying it. */
int rgb_synth_apply_tick_device(rgb_ctx *ctx, const void *d_msgs, uint32_t max_msgs, void *d_decisions,
                                void *d_rpcs, void *stream);

#ifdef __cplusplus
}
#endif
#endif
