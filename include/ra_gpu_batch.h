/*
 * ra_gpu_batch.h -- C ABI of libra_gpu_batch: a batched, MI355X-resident
 * evaluator of the per-server Raft state transition that rabbitmq/ra runs
 * one-process-per-group in ra_server:handle_leader/2, handle_follower/2,
 * handle_candidate/2, handle_pre_vote/2 and handle_await_condition/2 for
 * the message classes append_entries_rpc / append_entries_reply /
 * request_vote_rpc / request_vote_result / {ra_log_event,{written,..}} /
 * pipeline_rpcs / {commands,..}.
 *
 * This header is the drop-in boundary.  The call site
 * it replaces is ra_server_proc:handle_raft_state/3 and handle_leader/2
 *   (reference src/ra_server_proc.erl:1356-1397), i.e. the call
 *   ra_server:RaftState(Msg, ServerState) -> {NextState, ServerState, Effects}
 *   (reference src/ra_server.erl:530-531, 1281-1282).
 * An Erlang NIF (ra_amd/csrc/ra_gpu_batch_nif.c) binds exactly these entry
 * points; INTEGRATION.md shows the Erlang-side stub.
 *
 * Plain C: pointers, sizes, fixed-width integers.  No C++/torch types.
 * All indexes and terms are uint64_t (ra_index()/ra_term() are
 * non_neg_integer(), reference src/ra.hrl:18-23); Erlang 'undefined' is
 * RGB_UNDEF; an undefined server id is RGB_NONE.
 */
#ifndef RA_GPU_BATCH_H
#define RA_GPU_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGB_ABI_VERSION   9u
#define RGB_UNDEF         UINT64_MAX   /* Erlang 'undefined' (index or term)            */
#define RGB_NONE          0xFFu        /* undefined ra_server_id() (member slot)        */
#define RGB_MAX_MEMBERS   8u           /* members per Raft group held on the device     */
#define RGB_MAX_PENDING_RANGES 3u     /* ranges of the ra_log `pending` ra_seq held per server */
#define RGB_MAX_RUNS      16u          /* (first_index, term) runs kept per server log  */

/* reference src/ra_server.hrl:7-8 (cfg defaults) */
#define RGB_AER_CHUNK_SIZE              128u
#define RGB_DEFAULT_MAX_PIPELINE_COUNT  4096u

/* ra_state(), reference src/ra.hrl:92-100 (subset handled on the device) */
enum {
  RGB_ROLE_FOLLOWER        = 0,
  RGB_ROLE_CANDIDATE       = 1,
  RGB_ROLE_LEADER          = 2,
  RGB_ROLE_PRE_VOTE        = 3,
  RGB_ROLE_AWAIT_CONDITION = 4
};

/* original reason of follower_catchup_cond_fun/1, reference src/ra_server.erl:2196-2216 */
enum {
  RGB_COND_NONE          = 0,
  RGB_COND_MISSING       = 1,
  RGB_COND_TERM_MISMATCH = 2,
  RGB_COND_WAL_DOWN      = 3,  /* follower whose ra_log:write/2 returned {error, wal_down} (src/ra_server.erl:1377-1385,
                                  wal_down_condition/2 :2232-2233).  The write is host I/O: the host re-uploads the server in
                                  role await_condition with this reason and the log as it was BEFORE the write; from then on
                                  it sets RGB_MF_CAN_WRITE on the server's messages once ra_log:can_write/1 is true again --
                                  the predicate of the reference.  No stored reply: a timeout just returns to follower. */
  RGB_COND_WAL_DOWN_LEADER = 4 /* leader whose ra_log:append/2 raised wal_down on a {command, _} (src/ra_server.erl:655-672):
                                  the same predicate (RGB_MF_CAN_WRITE), but the condition carries transition_to => leader
                                  and a timeout that also returns to leader with [{next_event, cast, {transfer_leadership,
                                  Peer}}] (RGB_F_TRANSFER_LEADERSHIP) when the log still cannot be written.  The append is
                                  host I/O: the host re-uploads the server as it was BEFORE the command, in role
                                  await_condition with this reason (test/ra_server_SUITE.erl:1035-1073, vector W2).       */
};

/* message kinds (one inbound ra_msg() for one server) */
enum {
  RGB_MSG_NOP           = 0,  /* empty slot in a dense tick                                        */
  RGB_MSG_AER           = 1,  /* #append_entries_rpc{}       src/ra.hrl:123-129                    */
  RGB_MSG_AER_REPLY     = 2,  /* {Peer,#append_entries_reply{}} src/ra.hrl:131-142                 */
  RGB_MSG_REQUEST_VOTE  = 3,  /* #request_vote_rpc{}         src/ra.hrl:145-149                    */
  RGB_MSG_VOTE_RESULT   = 4,  /* #request_vote_result{}      src/ra.hrl:152-154                    */
  RGB_MSG_WRITTEN       = 5,  /* {ra_log_event,{written,Term,Seq}} (contiguous Seq = [from..to])    */
  RGB_MSG_PIPELINE_RPCS = 6,  /* info event pipeline_rpcs    src/ra_server.erl:793-801             */
  RGB_MSG_APPEND        = 7,  /* {command,_} / {commands,_}: leader appended n entries in its term */
  RGB_MSG_AWAIT_TIMEOUT = 8,  /* await_condition_timeout     src/ra_server.erl:1932-1945           */
  RGB_MSG_ELECTION_TIMEOUT = 9,  /* election_timeout -> call_for_election/2 src/ra_server.erl:2877-2924 */
  RGB_MSG_PRE_VOTE_RPC     = 10, /* #pre_vote_rpc{}          src/ra.hrl:157-166                    */
  RGB_MSG_PRE_VOTE_RESULT  = 11, /* #pre_vote_result{}       src/ra.hrl:168-171                    */
  RGB_MSG_SNAPSHOT_WRITTEN = 12, /* {ra_log_event,{snapshot_written,{Idx,Term},_,snapshot,_,_}}
                                    src/ra_log.erl:1054-1150: the log prefix up to Idx is released */
  RGB_MSG_HEARTBEAT_RPC    = 13, /* #heartbeat_rpc{}         src/ra.hrl:193-196                    */
  RGB_MSG_HEARTBEAT_REPLY  = 14, /* {Peer,#heartbeat_reply{}} src/ra.hrl:198-200                   */
  RGB_MSG_CONSISTENT_QUERY = 15  /* {consistent_query,_,_} / {consistent_aux,_,_} with
                                    cluster_change_permitted = true  src/ra_server.erl:855-860, 868-873:
                                    the query itself (a fun) stays queued on the host under the
                                    query_index the decision returns                               */
};
#define RGB_MSG_KIND_MAX RGB_MSG_CONSISTENT_QUERY
#define RGB_PROTO_VERSION 1u    /* ?RA_PROTO_VERSION src/ra.hrl:107 */

/* rgb_msg.flags */
#define RGB_MF_SUCCESS  0x01u  /* AER_REPLY: success=true; VOTE_RESULT: vote_granted=true */
#define RGB_MF_FORCE    0x02u  /* APPEND: noop command => Force pipelining (src/ra_server.erl:682-689) */
#define RGB_MF_SEQ2     0x08u  /* WRITTEN: the written ra_seq has TWO ranges: [run0_term .. run1_term] (the lower one, reusing
                                  those two fields as indexes) and [a .. b] above it, run1_term + 1 < a */
#define RGB_MF_SEQX     0x20u  /* WRITTEN (with RGB_MF_SEQ2): the written ra_seq has MORE than two ranges.  The two highest ride
                                  in the record as for RGB_MF_SEQ2; the others -- ascending, non-adjacent, all below
                                  run0_term -- are entries c .. c + n_entries - 1 of the batch's RANGE LIST ((first, last)
                                  pairs of uint64: rgb_submit_seq, rgb_set_seq_ranges_device).  A record that names entries
                                  the list does not hold commits nothing (RGB_INV_WRITTEN_SEQ_LIST).  Served by the
                                  kind-GENERIC kernel: rgb_submit_seq routes a batch that holds such a record there; a
                                  device-resident producer runs such a tick with rgb_run_ticks_device WITHOUT kind counts
                                  (the class-specialised and train kernels report the record RGB_F_UNHANDLED, state
                                  unchanged) */
#define RGB_MF_CAN_WRITE 0x10u /* any kind, for a server awaiting RGB_COND_WAL_DOWN: ra_log:can_write(Log) is true */
#define RGB_MF_TICK     0x04u  /* PIPELINE_RPCS: the leader's tick_timeout -- ra_server:make_rpcs/1: heartbeats for
                                  waiting queries plus one batch-of-1 rpc per stale peer, next_index not advanced
                                  (src/ra_server.erl:2348-2351, 2369-2377, 3012-3030; src/ra_server_proc.erl:613-616) */

/*
 * One inbound message, 64 bytes.  Field use by kind:
 *   AER           term, from=leader_id, a=prev_log_index, b=prev_log_term, c=leader_commit,
 *                 entries = n_entries contiguous indexes starting at a+1+gap; the first n_run0
 *                 carry term run0_term, the remaining carry run1_term (payloads stay on the host)
 *   AER_REPLY     term, from=peer, flags&SUCCESS, a=next_index, b=last_index, c=last_term
 *   REQUEST_VOTE  term, from=candidate_id, a=last_log_index, b=last_log_term
 *   VOTE_RESULT   term, from=voter, flags&SUCCESS=vote_granted
 *   WRITTEN       term, a=first index of the written range, b=last index of the written range
 *   APPEND        n_entries = number of commands appended by the leader (flags&FORCE for noop)
 *   PIPELINE_RPCS (no fields); flags&TICK = tick_timeout on a leader (make_rpcs/1 over stale_peers/1)
 *   ELECTION_TIMEOUT  c = fresh pre-vote token (the host's make_ref())
 *   PRE_VOTE_RPC  term, from=candidate_id, a=last_log_index, b=last_log_term, c=token,
 *                 n_entries=candidate machine_version, gap=protocol version
 *   PRE_VOTE_RESULT term, from=voter, flags&SUCCESS=vote_granted, c=token
 *   SNAPSHOT_WRITTEN a=snapshot index, b=snapshot term (kind `snapshot`, not `checkpoint`)
 *   HEARTBEAT_RPC term, from=leader_id, a=query_index
 *   HEARTBEAT_REPLY term, from=peer, a=query_index
 *   CONSISTENT_QUERY (no fields)
 */
typedef struct rgb_msg {
  uint32_t server;      /* target server id = group * n_members + member slot */
  uint8_t  kind;        /* RGB_MSG_*                                          */
  uint8_t  from;        /* sender member slot or RGB_NONE                     */
  uint8_t  flags;       /* RGB_MF_*                                           */
  uint8_t  gap;         /* AER: first entry index = a + 1 + gap (0 normally)  */
  uint64_t term;
  uint64_t a;
  uint64_t b;
  uint64_t c;
  uint32_t n_entries;
  uint32_t n_run0;
  uint64_t run0_term;
  uint64_t run1_term;
} rgb_msg;

/* rgb_decision.flags: the effects()/state facts the owning gen_statem must act on */
#define RGB_F_REPLY          (1u << 0)  /* {cast,To,{Id,#append_entries_reply{}}} or {reply,#request_vote_result{}} */
#define RGB_F_REPLY_SUCCESS  (1u << 1)  /* reply.success / reply.vote_granted                                        */
#define RGB_F_REPLY_VOTE     (1u << 2)  /* the reply is a #request_vote_result{}                                     */
#define RGB_F_PERSIST        (1u << 3)  /* update_term_and_voted_for stored term/voted_for (src/ra_server.erl:3041-3058) */
#define RGB_F_LEADER_MSG     (1u << 4)  /* {record_leader_msg, LeaderId}                                              */
#define RGB_F_LEADER_CHANGED (1u << 5)  /* leader_id differs from before                                              */
#define RGB_F_APPLIED        (1u << 6)  /* last_applied advanced: apply (old+1 .. last_applied)                       */
#define RGB_F_AUX_EVAL       (1u << 7)  /* {aux, eval}                                                                */
#define RGB_F_WROTE          (1u << 8)  /* follower ra_log:write of entries reply_next_index..reply_last_index        */
#define RGB_F_TRUNCATED      (1u << 9)  /* ra_log:set_last_index(prev_log_index) was applied                          */
#define RGB_F_PIPELINE       (1u << 10) /* {next_event, info, pipeline_rpcs}                                          */
#define RGB_F_REPROCESSED    (1u << 11) /* {next_event, Msg}: role changed and Msg was re-processed in the new role   */
#define RGB_F_ROLE_CHANGED   (1u << 12)
#define RGB_F_BECAME_LEADER  (1u << 13) /* candidate won: post_election_effects are the host's                        */
#define RGB_F_UNHANDLED      (1u << 14) /* catch-all clause: state unchanged                                          */
#define RGB_F_INVARIANT      (1u << 15) /* the reference would exit/assert; code in .invariant; state unchanged       */
#define RGB_F_RUNS_OVERFLOW  (1u << 16) /* term-run table overflowed: oldest run dropped, first_index raised          */
#define RGB_F_SEND_SNAPSHOT  (1u << 17) /* a pipelined peer needs {send_snapshot,..} (rgb_rpc kind RGB_RPC_SNAPSHOT)  */
#define RGB_F_REPLY_PRE_VOTE (1u << 18) /* the reply is a #pre_vote_result{}: reply_term, token in reply_next_index   */
#define RGB_F_START_ELECTION_TIMEOUT (1u << 19) /* effect start_election_timeout                                      */
#define RGB_F_SEND_VOTE_REQUESTS (1u << 20) /* {send_vote_requests,Reqs} to every peer: term=reply_term,
                                             last_log_index=reply_last_index, last_log_term=reply_last_term;
                                             #request_vote_rpc{} unless RGB_F_PRE_VOTE_REQS                           */
#define RGB_F_PRE_VOTE_REQS  (1u << 21) /* the requests are #pre_vote_rpc{} with token=reply_next_index               */
#define RGB_F_REPLY_HEARTBEAT (1u << 23) /* the reply is a #heartbeat_reply{term=reply_term, query_index=reply_next_index} */
#define RGB_F_SEND_HEARTBEATS (1u << 24) /* {send_rpc,Peer,#heartbeat_rpc{}} to every member slot set in
                                            rgb_decision.heartbeat_to: term=reply_term, query_index=reply_last_term
                                            (heartbeat_rpc_effects/4 src/ra_server.erl:3775-3795)                  */
#define RGB_F_QUERY_QUORUM   (1u << 25) /* a same-term #heartbeat_reply{} was counted: reply_next_index is the
                                           consensus query index (get_current_query_quorum/1 :3831-3832); the host
                                           releases its queued queries up to it (heartbeat_rpc_quorum/3 :3797-3814) */
#define RGB_F_QUERY_APPLY    (1u << 26) /* no peers: the consistent query (or every waiting one) applies now
                                           (src/ra_server.erl:3738-3743, 3757-3760)                                */
#define RGB_F_CANCEL_SNAPSHOT_RETRY (1u << 27) /* {cancel_snapshot_retry_timer, Peer} for every member slot set in
                                                   rgb_decision.cancel_backoff: make_all_rpcs/1 also contacts peers in
                                                   {snapshot_backoff, _} (src/ra_server.erl:2353-2367)                  */
#define RGB_F_RESEND_PENDING (1u << 22) /* the written event is not a prefix of `pending` (a WAL gap): the host runs
                                           ra_log:resend_pending/2 (src/ra_log.erl:917-919, 1663-1700); the log
                                           cursors are unchanged                                                 */

#define RGB_F_TRANSFER_LEADERSHIP (1u << 29) /* await_condition_timeout of a leader's wal_down condition with the log still
                                               not writable: [{next_event, cast, {transfer_leadership, PeerId}}] where PeerId is
                                               the host's hd(maps:to_list(maps:remove(Self, Cluster))) -- raised only when the
                                               cluster has another member (src/ra_server.erl:660-668, 1932-1945)           */
#define RGB_F_COMPACT        (1u << 28) /* DEVICE-RESIDENT decision streams only (rgb_run_ticks_device, rgb_train_run_device,
                                           the generator's apply): only the first 32 bytes of this 64-byte slot were written
                                           -- the compact form below; rgb_decision_expand() gives the record back.
                                           rgb_collect always hands out full records (expanded on the device) */

/* rgb_decision.invariant: exit reasons / failed assertions of the reference */
enum {
  RGB_INV_NONE                      = 0,
  RGB_INV_LEADER_SAW_AER_SAME_TERM  = 1, /* exit(leader_saw_append_entries_rpc_in_same_term) src/ra_server.erl:845-849 */
  RGB_INV_TRUNCATE_BELOW_APPLIED    = 2, /* ?assertNot(PLIdx < LastApplied)  src/ra_server.erl:1317 */
  RGB_INV_WRITE_BELOW_APPLIED       = 3, /* ?assertNot(FstIdx < LastApplied) src/ra_server.erl:1370 */
  RGB_INV_MISMATCH_TERM_UNDEFINED   = 4, /* ?assert(LATerm =/= undefined)    src/ra_server.erl:3616-3617 */
  RGB_INV_WRITE_INTEGRITY           = 5, /* {error,{integrity_error,_}} -> exit(Err) src/ra_log.erl:592-599 */
  RGB_INV_SET_LAST_INDEX_NOT_FOUND  = 6, /* {ok,L} = ra_log:set_last_index badmatch  src/ra_log.erl:857-859 */
  RGB_INV_LAST_WRITTEN_TERM         = 7, /* true = Term =/= undefined        src/ra_log.erl:576, 878 */
  RGB_INV_NEXT_INDEX_REGRESSED      = 8, /* ?assert(NewNextIdx >= NextIdx)   src/ra_server.erl:2333 */
  RGB_INV_PIPELINE_PREV_UNDEFINED   = 9, /* make_rpc_effect: no term for NextIdx-1 and no snapshot above it
                                            (case_clause / ?assert(PrevIdx < SnapIdx)) src/ra_server.erl:2392-2408 */
  RGB_INV_WRITTEN_NOT_PREFIX        = 10, /* {ok, Pend} = ra_seq:remove_prefix(..) badmatch  src/ra_log.erl:929 */
  RGB_INV_LEADER_SAW_HEARTBEAT_SAME_TERM = 11, /* exit(leader_saw_heartbeat_rpc_in_same_term) src/ra_server.erl:898-903 */
  RGB_INV_WRITTEN_SEQ_LIST          = 12  /* (no reference clause) a RGB_MF_SEQX record names entries its range list does not hold */
};

/*
 * One decision per message, 64 bytes, same order as the submitted messages.
 * commit_index / last_applied are the server's values AFTER the transition.
 * When RGB_F_WROTE is set (non-empty append: the reference sends no reply now,
 * src/ra_server.erl:1373-1376) reply_next_index/reply_last_index hold the first/last
 * entry index actually written after drop_existing/3.
 */
typedef struct rgb_decision {
  uint32_t server;
  uint8_t  role;       /* next ra_state()                          */
  uint8_t  reply_to;   /* member slot the reply goes to / RGB_NONE */
  uint8_t  n_rpcs;     /* rgb_rpc records emitted for this message */
  uint8_t  kind;       /* echo of rgb_msg.kind                     */
  uint32_t flags;      /* RGB_F_*                                  */
  uint16_t invariant;  /* RGB_INV_*                                */
  uint8_t  heartbeat_to; /* RGB_F_SEND_HEARTBEATS: bit i = member slot i gets a #heartbeat_rpc{} */
  uint8_t  cancel_backoff; /* RGB_F_CANCEL_SNAPSHOT_RETRY: bit i = cancel member slot i's snapshot retry timer */
  uint64_t reply_term;
  uint64_t reply_next_index;
  uint64_t reply_last_index;
  uint64_t reply_last_term;
  uint64_t commit_index;
  uint64_t last_applied;
} rgb_decision;

/* Compact decisions.  The decision stream is a sixth of the path's memory traffic and a tick's time follows its bytes
 * (32-byte decisions instead of 64: -8.6 % per tick), while the steady-state outcomes -- a follower
 * that appended or confirmed, a leader that counted a reply -- carry a handful of values that lie close together.  Such
 * a decision is written as 32 bytes into its 64-byte slot (the other half keeps whatever the buffer held):
 *   bytes 0..7   as in rgb_decision (server, role, reply_to, n_rpcs, kind)
 *   bytes 8..11  flags with RGB_F_COMPACT        bytes 12..15  aux (where invariant / heartbeat_to / cancel_backoff sit;
 *                                                              all three are 0 in every decision that is compacted)
 *   bytes 16..23 A    bytes 24..31 B
 * Three forms, told apart by kind and flags (every other field of the expanded record is 0):
 *   counted     kind AER_REPLY, or WRITTEN without RGB_F_REPLY: commit_index = A, last_applied = B
 *   wrote       kind AER with RGB_F_WROTE (no reply): reply_last_index = A, reply_next_index = A - aux[0:16],
 *               commit_index = B, last_applied = A - aux[16:32]
 *   confirmed   kind AER or WRITTEN with RGB_F_REPLY | RGB_F_REPLY_SUCCESS: reply_next_index = A + 1, reply_term = B,
 *               reply_last_index = A - aux[0:8], reply_last_term = B - aux[8:12],
 *               commit_index = A + aux[12:22] - 512, last_applied = A + 1 - aux[22:32]
 * A decision whose values do not fit (or that is of any other shape) is written in full, without the flag: the
 * encoding loses nothing.  The kernels compact at the store; rgb_decision_expand is the only decoder a consumer needs. */
static inline void rgb_decision_expand(rgb_decision *d) {
  if (!(d->flags & RGB_F_COMPACT)) return;
  uint64_t w[4];
  const unsigned char *raw = (const unsigned char *)d;
  for (int k = 0; k < 4; ++k) {
    uint64_t v = 0;
    for (int b = 7; b >= 0; --b) v = (v << 8) | raw[8 * k + b];
    w[k] = v;
  }
  const uint32_t aux = (uint32_t)(w[1] >> 32);
  const uint64_t A = w[2], B = w[3];
  d->flags &= ~RGB_F_COMPACT;
  d->invariant = 0; d->heartbeat_to = 0; d->cancel_backoff = 0;
  d->reply_term = d->reply_next_index = d->reply_last_index = d->reply_last_term = 0;
  if (d->flags & RGB_F_REPLY) {
    d->reply_next_index = A + 1; d->reply_term = B;
    d->reply_last_index = A - (aux & 0xFFu); d->reply_last_term = B - ((aux >> 8) & 0xFu);
    d->commit_index = A + ((aux >> 12) & 0x3FFu) - 512u; d->last_applied = A + 1 - ((aux >> 22) & 0x3FFu);
  } else if (d->flags & RGB_F_WROTE) {
    d->reply_last_index = A; d->reply_next_index = A - (aux & 0xFFFFu);
    d->commit_index = B; d->last_applied = A - (aux >> 16);
  } else {
    d->commit_index = A; d->last_applied = B;
  }
}

enum { RGB_RPC_AER = 1, RGB_RPC_SNAPSHOT = 2 };

/*
 * One outbound {send_rpc, Peer, #append_entries_rpc{}} shaped by
 * make_pipelined_rpc_effects/3 (src/ra_server.erl:2285-2346): entries are
 * prev_log_index+1 .. prev_log_index+n_entries, read from ra_log by the host.
 * RGB_RPC_SNAPSHOT: {send_snapshot, Peer, _}; prev_log_index = snapshot index.
 * rgb_collect returns them ordered by (msg_index, peer).
 */
typedef struct rgb_rpc {
  uint32_t msg_index;   /* index of the triggering message in the submitted batch */
  uint32_t server;
  uint8_t  peer;
  uint8_t  kind;        /* RGB_RPC_* */
  uint16_t n_entries;
  uint32_t _pad;
  uint64_t term;
  uint64_t prev_log_index;
  uint64_t prev_log_term;
  uint64_t leader_commit;
  uint64_t next_index;  /* the peer's next_index after this rpc */
} rgb_rpc;

/*
 * Host-visible state of one ra_server (one member of one group): the integer
 * part of ra_server_state() (src/ra_server.erl:73-112), ra_peer_state()
 * (src/ra.hrl:61-73) and the ra_log cursors (src/ra_log.erl:830-839, 1166-1200).
 * The log's index->term map is a run-length table: run i covers indexes
 * run_start[i] .. run_start[i+1]-1 (the last run ends at last_index) with
 * term run_term[i]; run_start[0] == first_index when the range is not empty.
 * Peer status is two masks: status_mask bit i = peer i is `normal`; backoff_mask bit i = peer i is
 * {snapshot_backoff, _} (the only non-normal status the path tells apart, make_all_rpcs/1
 * src/ra_server.erl:2356-2363); every other status is "neither bit".  become(follower,_,_) and
 * initialise_peers/1 make every peer normal.
 * The ra_log range is {first_index, last_index}; it is empty (undefined)
 * iff first_index > last_index, in which case (last_index,last_term) equal
 * the snapshot's (src/ra_log.erl:831-835).
 */
typedef struct rgb_server_state {
  uint64_t current_term;
  uint64_t commit_index;
  uint64_t last_applied;
  uint64_t last_index;
  uint64_t last_term;
  uint64_t last_written_index;
  uint64_t last_written_term;
  uint64_t snapshot_index;      /* RGB_UNDEF: no snapshot */
  uint64_t snapshot_term;
  uint64_t first_index;
  uint64_t cond_reply[4];       /* stored timeout reply of await_condition: term,next_index,last_index,last_term */
  uint64_t match_index[RGB_MAX_MEMBERS];
  uint64_t next_index[RGB_MAX_MEMBERS];
  uint64_t commit_index_sent[RGB_MAX_MEMBERS];
  uint64_t run_start[RGB_MAX_RUNS];
  uint64_t run_term[RGB_MAX_RUNS];
  uint8_t  role;                /* RGB_ROLE_*                                             */
  uint8_t  cond_reason;         /* RGB_COND_* (role == AWAIT_CONDITION)                   */
  uint8_t  self;                /* own member slot                                        */
  uint8_t  n_members;
  uint8_t  voted_for;           /* member slot / RGB_NONE                                 */
  uint8_t  leader_id;           /* member slot / RGB_NONE                                 */
  uint8_t  votes;
  uint8_t  n_runs;
  uint8_t  present_mask;        /* bit i: member i is a key of the cluster map            */
  uint8_t  voter_mask;          /* bit i: member i's voter_status is voter (or absent)    */
  uint8_t  status_mask;         /* bit i: peer i status == normal                         */
  uint8_t  self_nonvoter;       /* own `membership` =/= voter                             */
  uint8_t  cond_leader;         /* await_condition: who the stored reply is cast to       */
  uint8_t  backoff_mask;  /* bit i = peer i's status is {snapshot_backoff, _} (its status_mask bit is 0) */
  uint8_t  n_pending_old; /* 0..2: ranges of `pending` BELOW its newest range, see pending_old */
  uint8_t  _pad[1];
  uint64_t pre_vote_token;      /* pre_vote_token (an Erlang reference, opaque 64 bits)   */
  uint64_t query_index;         /* query_index (src/ra_server.erl:96): consistent-query heartbeat counter */
  uint64_t peer_query_index[RGB_MAX_MEMBERS]; /* #{query_index} of every peer (src/ra.hrl:61-73); own slot unused (0) */
  uint64_t pending_first;       /* ra_log `pending` (src/ra_log.erl:126): the indexes handed to the WAL
                                   and not yet confirmed are [pending_first .. last_index]; empty is
                                   stored as last_index + 1 (set it so on upload)            */
  uint32_t machine_version;     /* cfg.machine_version                                    */
  uint32_t effective_machine_version; /* cfg.effective_machine_version                    */
  uint64_t pending_old[2][2];   /* `pending` is a ra_seq (src/ra_seq.erl:8-12): after ra_log:write_sparse/3 (snapshot
                                   installation with live indexes, src/ra_log.erl:601-635) it has gaps.  Up to
                                   RGB_MAX_PENDING_RANGES = 3 ranges are held: the newest is [pending_first .. last_index]
                                   (above), the n_pending_old older ones are pending_old[k] = {first, last}, ascending,
                                   non-adjacent (pending_old[k][1] + 1 < the next range's first), all below the newest
                                   range and below last_index + 1.  A server whose pending has more ranges stays on the
                                   host until written events have shortened it.                                      */
} rgb_server_state;

/* ra_leaderboard row + key_metrics gauges per group (src/ra_leaderboard.erl:18-26, src/ra.erl:1242-1250) */
typedef struct rgb_leaderboard_row {
  uint32_t leader;        /* member slot of the leader with the highest term, RGB_NONE if none */
  uint32_t n_leaders;     /* members currently in role leader (split-brain diagnostics)        */
  uint64_t term;          /* highest current_term among members                                */
  uint64_t commit_index;  /* the leader's commit_index (max over members when no leader)       */
  uint64_t last_applied;  /* the leader's last_applied (max over members when no leader)       */
} rgb_leaderboard_row;

#define RGB_CFG_ROUNDS_PER_LAUNCH 1u   /* rgb_submit: one kernel launch per sub-tick round, never a train -- the DEFAULT since
                                          round 5 (the flag is accepted and changes nothing; it wins over RGB_CFG_SUBMIT_TRAINS) */
#define RGB_CFG_SUBMIT_TRAINS    16u  /* OPT-IN (round 5; the default of rounds 3-4): rgb_submit runs the sub-tick rounds of a
                                         batch (2..16 rounds, >= 4096 messages) as ONE train launch (see "Train launches",
                                         "Fail-safe").  Bit-identical results; measured no faster than one launch per round on
                                         this path -- 54.7 against 55.8 M decisions/s on 856 k-message batches of four rounds,
                                         318 against 312 us for the round trip of a 13.5 k-message batch of four rounds
                                         (bench.py, host_path.rounds4 / rounds4_small): the host path is bound by the host's
                                         passes over the batch and the two copies, not by launches -- so the simpler form,
                                         whose progress does not rest on how the dispatcher places blocks, is the default */
#define RGB_CFG_TRAIN_PERSISTENT 2u   /* trains always in the persistent form (placement by construction), also on a device
                                         whose dispatcher deals blocks round robin (see "Train launches") */
#define RGB_CFG_FUSE_PIPELINE    4u   /* OPT-IN (ABI v8; default off: the decision stream is the reference's event by
                                         event).  A leader's same-term success reply and its own written event end with
                                         {next_event, info, pipeline_rpcs} (src/ra_server.erl:552, 744), which the
                                         gen_statem handles before anything else in its mailbox (:793-801): with this
                                         flag the SAME decision carries the rpc records of that pipeline_rpcs event
                                         (make_pipelined_rpc_effects :2285-2346) -- n_rpcs and its rpc slots filled,
                                         next_index / commit_index_sent advanced, RGB_F_PIPELINE left only when the
                                         event would re-arm itself (More = true) -- instead of RGB_F_PIPELINE and a second
                                         message (RGB_MSG_PIPELINE_RPCS) from the host.  State and records equal the two
                                         steps bit for bit (tests/test_gpu_parity.py); if the event would fail a
                                         reference assertion the decision is NOT fused (RGB_F_PIPELINE stays: the host's
                                         message then reports the invariant as always) */
typedef struct rgb_config {
  uint32_t abi_version;          /* RGB_ABI_VERSION                                             */
  int32_t  device;               /* HIP device ordinal                                          */
  uint32_t max_runs;             /* term runs kept per server on the device, 2..RGB_MAX_RUNS    */
  uint32_t ring_slots;           /* pinned staging ring: batches in flight (>=1)                */
  uint32_t ring_capacity;        /* messages per ring slot                                      */
  uint32_t max_pipeline_count;   /* cfg.max_pipeline_count                                      */
  uint32_t max_aer_batch;        /* cfg.max_append_entries_rpc_batch_size                       */
  uint32_t flags;                /* RGB_CFG_*                                                   */
} rgb_config;

typedef struct rgb_ctx rgb_ctx;

/* error codes: 0 ok, negative failure; never aborts */
enum {
  RGB_OK            =  0,
  RGB_E_INVAL       = -1,
  RGB_E_NOMEM       = -2,
  RGB_E_HIP         = -3,  /* rgb_last_hip_error() has the hipError_t */
  RGB_E_STATE       = -4,
  RGB_E_FULL        = -5,  /* staging ring full: collect first        */
  RGB_E_EMPTY       = -6,  /* nothing submitted                       */
  RGB_E_UNSUPPORTED = -7,
  RGB_E_NODEVICE    = -8,  /* no HIP device: there is no CPU fallback */
  RGB_E_COMM        = -9   /* RCCL failed: rgb_comm_last_error()      */
};

uint32_t    rgb_abi_version(void);
/* sizeof() of the ABI structs as compiled: 0 rgb_msg, 1 rgb_decision, 2 rgb_rpc,
 * 3 rgb_server_state, 4 rgb_leaderboard_row, 5 rgb_config (bindings verify their mirrors) */
size_t      rgb_struct_size(int which);
const char *rgb_strerror(int code);
void        rgb_default_config(rgb_config *cfg);

int  rgb_open(const rgb_config *cfg, rgb_ctx **out);
void rgb_close(rgb_ctx *ctx);
int  rgb_last_hip_error(const rgb_ctx *ctx);

/* allocate device state for n_groups x n_members servers, every server = ra_server:init/1 of an
 * empty log (empty_state: term 0, log [0:0], peers next_index 1 / match_index 0). */
int  rgb_register_groups(rgb_ctx *ctx, uint32_t n_groups, uint32_t n_members);
uint32_t rgb_n_servers(const rgb_ctx *ctx);

/* host <-> device state transfer for servers [first, first+n) */
int  rgb_upload_state(rgb_ctx *ctx, uint32_t first, uint32_t n, const rgb_server_state *in);
int  rgb_download_state(rgb_ctx *ctx, uint32_t first, uint32_t n, rgb_server_state *out);

/* Asynchronous host path: copy n messages into the pinned staging ring, enqueue
 * H2D + transition kernel(s) + the results kernels (which write decisions and rpc records into the slot's pinned host
 * buffers: no D2H copy command) on the context's stream and return.  (The H2D copy of a batch of 2 MB and more runs
 * on a second stream the decision stream waits for: PCIe carries batch k + 1 in while batch k's results go out.)  Messages for the same
 * server are applied in submission order (serialised over sub-ticks); messages for different
 * servers are applied in parallel.  rgb_collect waits for the OLDEST submitted batch. */
int  rgb_submit(rgb_ctx *ctx, const rgb_msg *msgs, uint32_t n, uint64_t tick);
/* rgb_submit with the batch's RANGE LIST (ABI v8): written events whose ra_seq has more than two ranges
 * (RGB_MF_SEQX; src/ra_log.erl:897-944, src/ra_seq.erl:17-66 -- what a WAL that has been rolling over under load
 * confirms) name their lower ranges in it: ranges = n_ranges x (first, last), copied with the batch.  A record whose
 * entries are not in the list, not ascending or not below its inline ranges: RGB_E_INVAL, nothing is enqueued. */
int  rgb_submit_seq(rgb_ctx *ctx, const rgb_msg *msgs, uint32_t n, uint64_t tick, const uint64_t *ranges, uint32_t n_ranges);
/* the same list for the device-resident entry points (rgb_run_ticks_device, the trains): the launches that follow
 * read d_ranges (device memory of the caller's, n_ranges pairs; NULL / 0 = none) until the next call */
int  rgb_set_seq_ranges_device(rgb_ctx *ctx, const void *d_ranges, uint32_t n_ranges);
int  rgb_collect(rgb_ctx *ctx, rgb_decision *out, uint32_t cap, uint32_t *n_out,
                 rgb_rpc *rpc_out, uint32_t rpc_cap, uint32_t *n_rpc_out, uint64_t *tick_out);
/* rgb_collect without the copy (ABI v9).  A batch's results are WRITTEN BY THE DEVICE into the slot's pinned host
 * buffers in exactly the form rgb_collect hands out -- full decisions in submission order, the rpc records compacted
 * and ordered by (msg_index, peer) -- so a consumer that can read them where they lie (a NIF that wraps them in
 * resource binaries, a C caller that walks them once) takes the oldest batch as a VIEW: pointers into the slot, valid
 * until rgb_release(ctx, view.slot).  A held slot is not free: rgb_submit returns RGB_E_FULL when the ring comes round
 * to it.  Same errors and the same threading contract as rgb_collect (each batch is handed out exactly once). */
typedef struct rgb_view {
  const rgb_decision *decisions;   /* n records, submission order */
  const rgb_rpc *rpcs;             /* n_rpcs records, ordered by (msg_index, peer) */
  uint64_t tick;
  uint32_t n;
  uint32_t n_rpcs;
  uint32_t slot;                   /* for rgb_release */
  uint32_t _pad;
} rgb_view;
int  rgb_collect_view(rgb_ctx *ctx, rgb_view *view);
int  rgb_release(rgb_ctx *ctx, uint32_t slot);
/* Threading (the interception point is per gen_statem, reference src/ra_server_proc.erl:1356-1397, so many
 * scheduler threads reach the boundary at once).
 * rgb_submit may be called from any number of threads.  Callers prepare their batches IN PARALLEL (validation, the
 * sub-tick rounds, the bucket sort into the slot's pinned buffer); they meet in two short critical sections: one
 * hands out the next ring slot and a ticket, the other lets the tickets through in order for the stream's work.
 * Batches reach the device in the order their submits took their slots; every batch is enqueued as a whole (all its
 * rounds).  RGB_E_FULL when the next ring slot is not free (nothing was enqueued).
 * rgb_collect may be called from any number of threads, concurrently with submits: waiting for the oldest batch,
 * the size check and taking the slot are serialised, the copy back to submission order is not, so consumers copy
 * different batches at once; each batch is handed out exactly once, oldest first.  A buffer that is too small
 * leaves the batch in the ring: RGB_E_INVAL (cap) or RGB_E_FULL (rpc_cap) with the needed counts in *n_out /
 * *n_rpc_out, the caller retries with larger buffers (or asks rgb_peek first).  RGB_E_STATE from rgb_collect (a
 * batch whose records are inconsistent, or whose train launch failed) consumes the batch.
 * rgb_wait parks the calling thread until a batch is in flight (RGB_OK), timeout_ms have passed or rgb_wake
 * was called (both RGB_E_EMPTY): a collector thread blocks here instead of polling rgb_collect.
 * rgb_upload_state / rgb_download_state / rgb_snapshot / rgb_state_checksum may run beside submit / collect from
 * any thread.  They serialise among themselves (shared staging buffers) and against the enqueue of batches: their
 * work is ordered against WHOLE batches -- never between the rounds of one -- and after every rgb_submit that
 * returned before the call.  A submit that is still preparing its batch in another thread may be enqueued before
 * or after; a caller that needs "after batch X" lets that rgb_submit return (or collects X) first.  The *_device
 * entry points take the CALLER's stream and are ordered by it alone. */
/* Sub-tick rounds: a batch that holds several messages for one server is applied in rounds (round r = every server's
 * r-th message), one launch per round.  With RGB_CFG_SUBMIT_TRAINS (opt-in since round 5) a batch of at least 4096
 * messages with 2..16 rounds and no NOP padding runs its rounds as ONE train launch instead (see "Train launches" below:
 * the per-server sequence bytes order a server's messages); rgb_submit_trains counts the batches that did.  Results
 * are identical either way. */
uint32_t rgb_submit_trains(const rgb_ctx *ctx);
/* sizes of the oldest batch in flight (waits for it, consumes nothing): decisions and rpc records rgb_collect will
 * hand out next -- a caller that allocates per batch (the NIF's binaries) sizes them from this */
int      rgb_peek(rgb_ctx *ctx, uint32_t *n_out, uint32_t *n_rpc_out);
int      rgb_wait(rgb_ctx *ctx, uint32_t timeout_ms);
void     rgb_wake(rgb_ctx *ctx);
uint32_t rgb_in_flight(const rgb_ctx *ctx);

/* Multi-GPU routing below any host language (the group -> node-local shard map that
 * ra_leaderboard / the ra_directory lookup give the reference, src/ra_leaderboard.erl:18-26): the context
 * (one per GPU) that owns a Raft group = splitmix64(group_uid) mod n_contexts.  Pure function. */
uint32_t rgb_route(uint64_t group_uid, uint32_t n_contexts);

/* ---- Multi-GPU: the one collective of the path ----
 * Groups are independent, so the decision path has no exchange step; what a node-wide view needs is the gathered
 * leaderboard / key-metrics snapshot (ra_leaderboard is a node-wide ETS table, src/ra_leaderboard.erl:18-26; the
 * gauges of src/ra.erl:1242-1270 are read per server): every context (one per GPU) produces one rgb_leaderboard_row per
 * LOCAL group (rgb_snapshot_device) and rgb_leaderboard_allgather gathers the shards of all ranks with RCCL
 * (ncclAllGather over xGMI) on the stream the caller gives -- the train launches' stream: ordered behind the snapshot
 * kernel, no event.  One rank creates the id (rgb_comm_unique_id) and hands its RGB_COMM_ID_BYTES to the others by
 * whatever the host has (Erlang distribution for the NIF: INTEGRATION.md); rgb_comm_init_rank is collective.  Every
 * rank contributes the SAME n_rows (hash sharding leaves the shards unequal: pad to the largest); rank r's rows land at
 * d_rows_all + r * n_rows.  RCCL is bound at run time (dlopen; a copy already in the process is used):
 * RGB_E_UNSUPPORTED when there is none, RGB_E_COMM with rgb_comm_last_error() for an RCCL failure. */
#define RGB_COMM_ID_BYTES 128u
typedef struct rgb_comm rgb_comm;
int  rgb_comm_unique_id(void *id_out /* RGB_COMM_ID_BYTES */);
int  rgb_comm_init_rank(rgb_ctx *ctx, const void *id, uint32_t n_ranks, uint32_t rank, rgb_comm **out);
void rgb_comm_destroy(rgb_comm *comm);
uint32_t rgb_comm_n_ranks(const rgb_comm *comm);
uint32_t rgb_comm_rank(const rgb_comm *comm);
int  rgb_leaderboard_allgather(rgb_ctx *ctx, rgb_comm *comm, const void *d_rows_local, uint32_t n_rows,
                               void *d_rows_all /* n_ranks * n_rows rows */, void *stream);
/* the same with host buffers (the NIF's form): this context's rows are produced, padded to n_rows >= its group
 * count, gathered and copied to rows_all (n_ranks * n_rows rows); synchronous for the CALLER only -- the locks of
 * rgb_submit / rgb_collect are held while the snapshot is enqueued, not while the ranks meet (the collective runs on a
 * side stream).  Collective-safe exits: every rank first takes part in an 8-byte status exchange; if any rank failed
 * locally (RGB_E_INVAL for n_rows below its group count, RGB_E_NOMEM, ..) no rank starts the gather, the failing rank
 * returns its own error and the others RGB_E_COMM.  A collective that does not complete within RGB_COMM_TIMEOUT_MS
 * (environment, default 30 000) is abandoned (ncclCommAbort): RGB_E_COMM, rgb_comm_last_error() says which; destroy
 * the communicator and create a new one. */
int  rgb_leaderboard_allgather_host(rgb_ctx *ctx, rgb_comm *comm, uint32_t n_rows, rgb_leaderboard_row *rows_all);
const char *rgb_comm_last_error(void);

/* Device-resident path (benchmarks, device-side producers): d_msgs holds n_ticks ticks laid out
 * tick_stride messages apart; tick t carries tick_counts[t] messages (host array; NULL = every
 * tick carries tick_stride messages) further clamped by d_tick_counts[t] when that device array
 * (uint32, written by a device-side producer) is not NULL; at most ONE message per server per
 * tick (caller's guarantee).  d_decisions has the same layout.  d_rpcs, when not NULL, receives
 * the pipelined rpcs of the CURRENT tick in fixed slots: message i owns records
 * [i*(n_members-1), (i+1)*(n_members-1)) of which the first rgb_decision.n_rpcs are valid; the
 * buffer (tick_stride*(n_members-1) records) is rewritten every tick.
 * kind_counts (host, may be NULL): uint32[n_ticks][RGB_MSG_KIND_MAX+1] message counts per kind for
 * ticks whose messages are ordered by clause family (the order rgb_submit and the load generator
 * produce: append_entries_rpc, append_entries_reply, written, append, then the rest); the library
 * then launches the class-dispatch kernel (each wavefront runs the code path specialised for its
 * slice's message kind) instead of the kind-generic kernel.
 * Enqueued on `stream` (a hipStream_t, NULL = the context's stream); returns without
 * synchronising. */
int  rgb_run_ticks_device(rgb_ctx *ctx, const void *d_msgs, uint32_t tick_stride,
                          const uint32_t *tick_counts, const void *d_tick_counts,
                          const uint32_t *kind_counts, uint32_t n_ticks, void *d_decisions,
                          void *d_rpcs, void *stream);

/* ---- Train launches: several device-resident ticks in ONE launch ----
 * rgb_run_ticks_device pays a kernel boundary per tick (launch gap, dispatch ramp, the write-back of every dirty
 * L2 line) and all wavefronts of a tick move through their memory and compute phases in lock step.  A TRAIN runs
 * consecutive ticks in one launch: what orders two messages of one server is not the kernel boundary but a
 * per-server sequence byte on the device, which a message waits for and its commit advances; wavefronts of
 * neighbouring ticks overlap.  Results are bit-identical to rgb_run_ticks_device on the same ticks.
 *
 * Requirements on the ticks: at most one message per server per tick; every tick ordered by BUCKET =
 * rgb_train_bucket(kind, flags, server, n_members) ascending -- (class of the kind, group mod 8, success flag); a
 * tick in that order is also in the clause-family order rgb_run_ticks_device wants.  The kernel checks the order
 * (RGB_TRAIN_ERR_ORDER).  Every message carries a STAMP = the value its server's sequence byte must hold when the
 * message is applied = the number of messages applied to that server before it, mod 256: a producer that counts what
 * it sends stamps as it writes (rgb_submit does, from its host mirror; rgb_synth_tick_stamped_device does, on the
 * device); rgb_train_stamp_device is the pass for streams whose producer did not.
 *
 * Coherence and the two forms of a launch.  The L2 caches of the XCDs are not coherent with each other, so all
 * messages of a server must be applied on ONE XCD during a launch: servers are sharded by group mod 8 and a shard's
 * messages only ever run on one XCD.  PERSISTENT form: the grid is the device's wavefront slots, a block serves the
 * shard of the XCD it runs on (HW_REG_XCC_ID; shards xcc, xcc + n, .. on a device of n < 8 XCCs) and takes that
 * shard's rows from a ticket counter -- placement by construction, no assumption about the dispatcher.  DEALT form:
 * one block per row, block b serves shard b mod 8 -- correct when the dispatcher deals the blocks of a launch round
 * robin over eight XCDs, 7 % faster where that holds (no ticket in front of every slice).  A context uses the dealt
 * form only if its calibration launch was dealt that way, EVERY block of every dealt launch verifies its placement
 * (RGB_TRAIN_ERR_PLACEMENT), and after one failure the context stays with the persistent form;
 * RGB_CFG_TRAIN_PERSISTENT asks for it from the start; rgb_train_form tells.
 *
 * Fail-safe.  A launch that fails (RGB_TRAIN_ERR_*) has applied SOME of its messages.  On the host path the engine
 * repairs that itself: every batch enqueued while a train is in flight saves the rows of the servers it touches
 * first (an undo log on the device); when rgb_collect (or any state call) finds a failed launch, the logs of the
 * batches in flight go back newest first, the batches run again oldest first with one launch per round, and
 * rgb_collect hands out what a faultless run would have -- every message applied in order, exactly once
 * (reference: src/ra_server_proc.erl:1356-1397); rgb_train_recoveries counts.  On the device-resident path the
 * caller owns the streams: rgb_train_status reports the error, the caller restores the state (rgb_upload_state) and
 * falls back to rgb_run_ticks_device.
 *
 *   rgb_train_plan_create   bucket_counts = uint32[n_ticks][RGB_TRAIN_BUCKETS] (host): builds and uploads the plan.
 *                           The first call of a context runs the calibration launch (which XCCs, dealt round robin
 *                           or not): RGB_E_UNSUPPORTED if the XCC ids are not 0 .. n-1 with n = 1, 2, 4 or 8
 *   rgb_train_stamp_device  d_stamps = uint8[n_ticks * tick_stride], laid out like d_msgs: the stamps of a stream
 *                           whose producer did not write them, counted tick by tick (tick_counts[t] messages in tick
 *                           t) from what the servers hold at this point of the stream.  Call it after the trains
 *                           enqueued before (same stream) and before the ones that use the stamps; the messages
 *                           themselves are not touched.
 *   rgb_train_run_device    ticks [first_tick, first_tick + n_ticks) of the plan; d_msgs / d_stamps / d_decisions
 *                           point at tick 0 of the plan; d_rpcs (may be NULL) holds rpc_ring regions of tick_stride *
 *                           (n_members-1) records, tick k of a launch uses region k mod rpc_ring (ticks of one launch
 *                           overlap: a region must not be reused within the overlap depth -- 4 is plenty).  More
 *                           than 255 ticks are split into several launches.
 *   rgb_train_status        after the caller synchronised the stream it used: 0 / RGB_E_STATE with the error flags
 *                           (RGB_TRAIN_ERR_*) and, optionally, the XCD every shard runs on.
 * The per-tick entry points never touch the sequence bytes, so trains and per-tick launches can alternate freely. */
#define RGB_TRAIN_BUCKETS 256u
#define RGB_TRAIN_ERR_PLACEMENT 1u   /* dealt form: a block ran on another XCD than its shard's (the L2s are not coherent) */
#define RGB_TRAIN_ERR_SPIN      2u   /* a wavefront's dependencies did not commit within the spin bound        */
#define RGB_TRAIN_ERR_ORDER     4u   /* a message sits in another shard's bucket: the tick is not in bucket order */
#define RGB_TRAIN_ERR_PLAN      8u   /* rgb_train_plan_build_device: a tick needs more rows than the plan's table holds */
#define RGB_TRAIN_FORM_NONE       0u /* no train has been set up yet (or the device cannot run them)            */
#define RGB_TRAIN_FORM_DEALT      1u
#define RGB_TRAIN_FORM_PERSISTENT 2u
typedef struct rgb_train_plan rgb_train_plan;
uint32_t rgb_train_bucket(uint32_t kind, uint32_t flags, uint32_t server, uint32_t n_members);
int  rgb_train_plan_create(rgb_ctx *ctx, const uint32_t *bucket_counts, uint32_t n_ticks, rgb_train_plan **out);
void rgb_train_plan_destroy(rgb_train_plan *plan);
uint32_t rgb_train_plan_blocks_per_tick(const rgb_train_plan *plan);
int  rgb_train_stamp_device(rgb_ctx *ctx, const void *d_msgs, void *d_stamps, uint32_t tick_stride,
                            const uint32_t *tick_counts, uint32_t n_ticks, void *stream);
int  rgb_train_run_device(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                          const void *d_msgs, const void *d_stamps, uint32_t tick_stride, void *d_decisions,
                          void *d_rpcs, uint32_t rpc_ring, void *stream);
/* Leaderboard snapshots INSIDE a train (ABI v7).  A train that covers several leaderboard periods (ra_leaderboard is
 * refreshed every few ticks, src/ra_leaderboard.erl:18-26) would pay a kernel boundary per period for the snapshot;
 * instead the snapshot's rows run as rows of the launch.  To the sequence bytes a snapshot is ONE MORE MESSAGE TO
 * EVERY SERVER: group by group it waits until every member has applied what came before the boundary, reads the
 * members' rows (the row of rgb_snapshot_device, bit for bit), and advances their bytes; the messages behind the
 * boundary carry stamps one higher.
 *   rgb_train_plan_create_snap   as rgb_train_plan_create; ticks k * snapshot_every (k >= 1) of the plan carry the
 *                                snapshot "in front of tick k * snapshot_every", ordinal k - 1
 *   rgb_train_run_snap_device    as rgb_train_run_device (one launch: at most 255 ticks); d_snap_stamps = ordinal-major
 *                                uint8[rgb_train_seq_bytes()] per snapshot (the value every server's byte must show at
 *                                that boundary -- the producer's count, rgb_synth_snapshot_mark_device), d_snap_rows =
 *                                rgb_leaderboard_row[n_groups] per ordinal.  The snapshot in front of the launch's
 *                                FIRST tick is not part of the launch:
 *   rgb_snapshot_train_device    rgb_snapshot_device + every sequence byte advanced: a boundary between two launches
 *                                (stream order does the waiting).  Every boundary of a stamped stream must be taken
 *                                exactly once, one way or the other. */
int  rgb_train_plan_create_snap(rgb_ctx *ctx, const uint32_t *bucket_counts, uint32_t n_ticks, uint32_t snapshot_every,
                                rgb_train_plan **out);
int  rgb_train_run_snap_device(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                               const void *d_msgs, const void *d_stamps, uint32_t tick_stride, void *d_decisions,
                               void *d_rpcs, uint32_t rpc_ring, const void *d_snap_stamps, void *d_snap_rows,
                               void *stream);
int  rgb_snapshot_train_device(rgb_ctx *ctx, void *d_rows, void *stream);
/* The plan built ON THE DEVICE (ABI v8): a device-resident producer leaves its bucket counts in device memory
 * (rgb_synth_tick_*_device: d_bucket_counts; rgb_submit's own device-side bucketing) and nothing of the plan passes
 * through the host -- the reference has no stop between a mailbox and its handler either
 * (src/ra_server_proc.erl:1382-1397).
 *   rgb_train_plan_create_device  an EMPTY plan of n_ticks ticks (snapshot_every as in rgb_train_plan_create_snap; 0 =
 *                                 none): tables sized for any tick of the registered groups (at most one message
 *                                 per server) whose message classes are spread evenly over the eight shards -- what
 *                                 hashing groups over the shards gives.  A tick skewed so that different classes peak
 *                                 in different shards can need up to eight times the rows: it is REFUSED (built as an
 *                                 empty tick, the launch reports RGB_TRAIN_ERR_PLAN through rgb_train_status and the
 *                                 ticks behind it do not run), never mis-run
 *   rgb_train_plan_build_device   ticks [first_tick, first_tick + n_ticks) of the plan from d_bucket_counts =
 *                                 uint32[n_ticks][RGB_TRAIN_BUCKETS] of exactly those ticks, one kernel on `stream`
 *                                 (offsets, rows per class, the row table: bit for bit what rgb_train_plan_create
 *                                 computes on the host).  Enqueue it behind the producer and in front of
 *                                 rgb_train_run*_device, same stream.
 * Launches of a device-built plan take the form the context's other trains take.  DEALT: the grid is the rows BOUND
 * of a tick (any tick the registered groups can produce), not the tick's rows, which only the device knows -- the
 * blocks behind a tick's real rows find an empty table entry and exit (65 536 x 5 closed loop: 11.5 k blocks per tick
 * instead of 5.2 k, +1 % per tick; the persistent form, rows from per-shard ticket counters: +4 %). */
int  rgb_train_plan_create_device(rgb_ctx *ctx, uint32_t n_ticks, uint32_t snapshot_every, rgb_train_plan **out);
int  rgb_train_plan_build_device(rgb_ctx *ctx, rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks,
                                 const void *d_bucket_counts, void *stream);
/*   rgb_train_plan_fit            optional: the HOST learns the rows of the built ticks [first_tick, first_tick + n_ticks)
 *                                 -- four bytes per tick come back, nothing else of the plan; synchronises `stream` -- and
 *                                 launches over ticks it knows take their rows as the grid instead of the rows bound
 *                                 (about half as many blocks: -2.5 % per tick in long launches, -4..6 % in a 20-tick one).
 *                                 Building a tick again forgets what was known of it. */
int  rgb_train_plan_fit(rgb_ctx *ctx, rgb_train_plan *plan, uint32_t first_tick, uint32_t n_ticks, void *stream);
/* inspection (tests, tools): tick `tick` of a plan as it stands on the device -- out_tick = the 16 header words (rows,
 * message base, snapshot ordinal + 1, padding) followed by off[30][8] and cnt[30][8] (1984 bytes), out_rows = its row
 * table (plan class << 24 | row of the class), at most rows_cap entries; returns the tick's rows or a negative error.  A plan built on a
 * stream of the caller's: synchronise that stream first */
int  rgb_train_plan_download(rgb_ctx *ctx, const rgb_train_plan *plan, uint32_t tick, void *out_tick, uint32_t *out_rows,
                             uint32_t rows_cap);
uint32_t rgb_train_seq_bytes(const rgb_ctx *ctx);     /* bytes of the per-server sequence array (and of one d_snap_stamps) */
int  rgb_train_status(rgb_ctx *ctx, uint32_t *flags_out, uint32_t *xcc_of_shard /* [8] or NULL */);
uint32_t rgb_train_form(const rgb_ctx *ctx);          /* RGB_TRAIN_FORM_*: how the next train launch will run */
uint32_t rgb_train_recoveries(const rgb_ctx *ctx);    /* failed train launches of rgb_submit the engine repaired */

/* leaderboard / metrics snapshot: one row per group */
int  rgb_snapshot(rgb_ctx *ctx, rgb_leaderboard_row *out);
int  rgb_snapshot_device(rgb_ctx *ctx, void *d_rows, void *stream);

/* 64-bit FNV-style checksum over the canonical state of servers [first, first+n): computed on
 * the device, used by size-independent parity checks. */
int  rgb_state_checksum(rgb_ctx *ctx, uint32_t first, uint32_t n, uint64_t *out);

int  rgb_synchronize(rgb_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RA_GPU_BATCH_H */
