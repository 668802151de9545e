%% ra_gpu_batch -- Erlang side of the NIF (ra_amd/csrc/ra_gpu_batch_nif.c) plus the helpers the
%% gen_statem shell needs to route ra_server transitions through the GPU engine.
%%
%% NOT COMPILED OR RUN IN THIS REPOSITORY'S ENVIRONMENT (no Erlang/OTP here); it documents the
%% reference-side binding a maintainer would add.  INTEGRATION.md walks through it.
%%
%% Records cross the NIF boundary as binaries with the C layout of include/ra_gpu_batch.h:
%%   rgb_msg      64 bytes, little endian:
%%     server:32, kind:8, from:8, flags:8, gap:8, term:64, a:64, b:64, c:64,
%%     n_entries:32, n_run0:32, run0_term:64, run1_term:64
%%   rgb_decision 64 bytes:
%%     server:32, role:8, reply_to:8, n_rpcs:8, kind:8, flags:32, invariant:16, heartbeat_to:8, cancel_backoff:8,
%%     reply_term:64, reply_next_index:64, reply_last_index:64, reply_last_term:64,
%%     commit_index:64, last_applied:64
-module(ra_gpu_batch).

-export([init/0, open/4, register_groups/3, upload_state/3, download_state/3, register_owner/4, unregister_owner/2, owner_slots/1, fan_back_stats/1, route/2,
         submit/3, submit/4, collect/1, start_collector/2, stop_collector/1, snapshot/2, wal_checksums/3,
         comm_unique_id/0, comm_init/4, allgather_leaderboard/2, node_leaderboard/3]).
-export([wal_batch_checksums/2, wal_frame/4, wal_recover_check/2, wal_frame_batch/3, wal_recover/2]).
-export([encode_msg/3, encode_msgs/1, decode_decision/1, decision_to_effects/3]).

-include_lib("ra/src/ra.hrl").

-on_load(init/0).

-define(MSG_AER, 1).
-define(MSG_AER_REPLY, 2).
-define(MSG_REQUEST_VOTE, 3).
-define(MSG_VOTE_RESULT, 4).
-define(MSG_WRITTEN, 5).
-define(MSG_PIPELINE_RPCS, 6).
-define(MSG_APPEND, 7).
-define(MSG_AWAIT_TIMEOUT, 8).
-define(MSG_ELECTION_TIMEOUT, 9).
-define(MSG_PRE_VOTE_RPC, 10).
-define(MSG_PRE_VOTE_RESULT, 11).
-define(MSG_SNAPSHOT_WRITTEN, 12).
-define(MSG_HEARTBEAT_RPC, 13).
-define(MSG_HEARTBEAT_REPLY, 14).
-define(MSG_CONSISTENT_QUERY, 15).
-define(NONE, 255).
-define(UNDEF, 16#FFFFFFFFFFFFFFFF).

-define(F_REPLY, 1).
-define(F_REPLY_SUCCESS, 2).
-define(F_REPLY_VOTE, 4).
-define(F_PERSIST, 8).
-define(F_LEADER_MSG, 16).
-define(F_APPLIED, 64).
-define(F_AUX_EVAL, 128).
-define(F_PIPELINE, 1024).
-define(F_INVARIANT, 32768).
-define(F_REPLY_PRE_VOTE, 262144).
-define(F_RESEND_PENDING, 4194304).
-define(F_REPLY_HEARTBEAT, 8388608).
-define(F_SEND_HEARTBEATS, 16777216).
-define(F_QUERY_QUORUM, 33554432).
-define(F_QUERY_APPLY, 67108864).
-define(F_CANCEL_SNAPSHOT_RETRY, 134217728).
-define(F_TRANSFER_LEADERSHIP, 536870912).

init() ->
    erlang:load_nif(filename:join(code:priv_dir(ra), "ra_gpu_batch_nif"), 0).

open(_Device, _MaxRuns, _RingSlots, _RingCapacity) -> erlang:nif_error(not_loaded).
register_groups(_Ctx, _NGroups, _NMembers) -> erlang:nif_error(not_loaded).
upload_state(_Ctx, _First, _Bin) -> erlang:nif_error(not_loaded).
download_state(_Ctx, _First, _N) -> erlang:nif_error(not_loaded).
%% route(GroupUId, NContexts) -> 0..NContexts-1: the context (one per GPU, one open/4 each) that owns a Raft group:
%% splitmix64(GroupUId) rem NContexts, the partition of SURVEY.md section 8(e).  GroupUId = any stable 64-bit id of
%% the cluster (e.g. erlang:phash2 is NOT stable across nodes; use the integer kept beside the ra_directory entry).
route(_GroupUId, _NContexts) -> erlang:nif_error(not_loaded).
%% register_owner(Ctx, FirstServer, N, Pid): the collector thread sends every decision of servers
%% [FirstServer, FirstServer+N) to Pid as {ra_gpu_batch, Tick, NDecisions, DecisionsBin, RpcsBin} (only that
%% process's decisions, submission order; rpc msg_index = position inside DecisionsBin).  A ra_server_proc
%% registers itself for its one server in init/1; unregistered servers go to the start_collector/2 pid.
register_owner(_Ctx, _FirstServer, _N, _Pid) -> erlang:nif_error(not_loaded).

%% unregister_owner(Ctx, Pid): Pid owns nothing any more (its servers go back to the default owner of
%% start_collector/2) and its slot of the owner table is free for the next register_owner/4 -- call it from
%% ra_server_proc:terminate/3, so that restarts do not grow the table (src/ra_server_proc.erl terminate/3).
unregister_owner(_Ctx, _Pid) -> erlang:nif_error(not_loaded).

%% owner_slots(Ctx) -> {Slots, Live}: diagnostics of the owner table.
owner_slots(_Ctx) -> erlang:nif_error(not_loaded).

%% fan_back_stats(Ctx) -> {Batches, Decisions, Nanoseconds} spent by the collector thread fanning batches back.
fan_back_stats(_Ctx) -> erlang:nif_error(not_loaded).
%% submit/3 may be called from any process; batches above 2048 messages run on a dirty CPU scheduler.
submit(_Ctx, _MsgsBin, _Tick) -> erlang:nif_error(not_loaded).
%% submit/4: + the batch's range list, <<First:64/little, Last:64/little>> per entry -- the lower ranges of written
%% events whose ra_seq has more than two ranges (encode_msgs/2 builds both binaries)
submit(_Ctx, _MsgsBin, _Tick, _RangesBin) -> erlang:nif_error(not_loaded).
collect(_Ctx) -> erlang:nif_error(not_loaded).
start_collector(_Ctx, _Pid) -> erlang:nif_error(not_loaded).
stop_collector(_Ctx) -> erlang:nif_error(not_loaded).
snapshot(_Ctx, _NGroups) -> erlang:nif_error(not_loaded).
%% The node-wide leaderboard over the GPUs of one node (one context per GPU, groups routed by route/2; the reference's
%% ra_leaderboard is one ETS table per node, src/ra_leaderboard.erl:18-26): RCCL all-gather behind the C ABI.
%% comm_unique_id() -> {ok, <<Id:128/binary>>} on ONE context's owner, Id sent to the others as a plain message;
%% comm_init(Ctx, Id, NRanks, Rank) -> ok and allgather_leaderboard(Ctx, NRows) -> {ok, RowsBin} are collective: the
%% owner process of every context calls them (dirty scheduler), NRows = the largest group count of any context.
comm_unique_id() -> erlang:nif_error(not_loaded).
comm_init(_Ctx, _Id, _NRanks, _Rank) -> erlang:nif_error(not_loaded).
allgather_leaderboard(_Ctx, _NRows) -> erlang:nif_error(not_loaded).

%% node_leaderboard(Ctx, NRows, GroupsOfRank) -> [{GroupUId, LeaderSlot | undefined, Term, CommitIndex, LastApplied}]
%% for EVERY group of the node, from this context's side of the collective: GroupsOfRank(Rank) -> [GroupUId] ascending
%% (the groups with route(GroupUId, NRanks) =:= Rank, in local-id order).  What ra_leaderboard:lookup_leader/1 and the
%% commit_index / last_applied gauges of ra:key_metrics/1 (src/ra.erl:1242-1270) read, for all GPUs at once.
node_leaderboard(Ctx, NRows, GroupsOfRank) ->
    {ok, Bin} = allgather_leaderboard(Ctx, NRows),
    RowBytes = 32,
    NRanks = byte_size(Bin) div (NRows * RowBytes),
    lists:append(
      [begin
           Shard = binary:part(Bin, Rank * NRows * RowBytes, NRows * RowBytes),
           [begin
                <<Leader:32/little, _NLeaders:32/little, Term:64/little, CI:64/little, LA:64/little>> =
                    binary:part(Shard, K * RowBytes, RowBytes),
                {UId, case Leader of 255 -> undefined; _ -> Leader end, Term, CI, LA}
            end || {K, UId} <- lists:zip(lists:seq(0, length(GroupsOfRank(Rank)) - 1), GroupsOfRank(Rank))]
       end || Rank <- lists:seq(0, NRanks - 1)]).
wal_checksums(_Ctx, _EntriesBin, _DataBin) -> erlang:nif_error(not_loaded).

wal_frame(_Ctx, _RecordsBin, _DataBin, _Flags) -> erlang:nif_error(not_loaded).
wal_recover_check(_Ctx, _FileBin) -> erlang:nif_error(not_loaded).

%% ra_log_wal:write_data/8 computes erlang:adler32([<<Idx:64, Term:64>> | EntryData]) per record
%% (src/ra_log_wal.erl:528-534); for a whole write_batch: Entries = [{Idx, Term, Bin}].
wal_batch_checksums(Ctx, Entries) ->
    {EntriesBin, Data, _} =
        lists:foldl(fun({Idx, Term, Bin}, {E, D, Off}) ->
                            Len = byte_size(Bin),
                            {<<E/binary, Idx:64/little, Term:64/little, Off:64/little,
                               Len:32/little, 0:32>>, [D, Bin], Off + Len}
                    end, {<<>>, [], 0}, Entries),
    {ok, Sums} = wal_checksums(Ctx, EntriesBin, iolist_to_binary(Data)),
    [C || <<C:32/little>> <= Sums].

%% The record bytes of a whole WAL batch.  Records = [{HeaderData, Idx, Term, EntryData}] where
%% HeaderData is what ra_log_wal:serialize_header/3 returned for the writer (the writer-name cache
%% stays in #wal{}); the result is the iodata write_data/8 would have accumulated in
%% #batch.pending (src/ra_log_wal.erl:513-537), ready for file:write/2 in flush_pending/1.
wal_frame_batch(Ctx, Records, ComputeChecksums) ->
    {RecsBin, Data, _} =
        lists:foldl(fun({Hdr, Idx, Term, Bin}, {R, D, Off}) ->
                            HLen = byte_size(Hdr),
                            Len = byte_size(Bin),
                            {<<R/binary, Idx:64/little, Term:64/little, (Off + HLen):64/little,
                               Len:32/little, HLen:32/little, Off:64/little, 0:64>>,
                             [D, Hdr, Bin], Off + HLen + Len}
                    end, {<<>>, [], 0}, Records),
    Flags = case ComputeChecksums of true -> 0; false -> 1 end,
    {ok, Framed} = wal_frame(Ctx, RecsBin, iolist_to_binary(Data), Flags),
    Framed.

%% Recovery of one WAL file (src/ra_log_wal.erl:877-1033): the walk and every checksum in one
%% call; returns the records to hand to recover_entry/5, in file order, or throws like the
%% reference.  IsRegistered = fun(UId) -> boolean() (ra_directory:is_registered_uid/2).
wal_recover(Ctx, FileBin) ->
    {ok, Scanned, NOk, Status} = wal_recover_check(Ctx, FileBin),
    Status =:= corrupt andalso throw(wal_checksum_validation_failure),
    Recs = [R || <<R:56/binary>> <= Scanned],
    {Good, _} = lists:split(NOk, Recs),
    {_, Out} =
        lists:foldl(
          fun(<<Idx:64/little, Term:64/little, DOff:64/little, DLen:32/little, _Crc:32/little,
                UOff:64/little, IdRef:32/little, ULen:16/little, Trunc:8, Flags:8, _Next:64/little>>,
              {Names0, Acc}) ->
                  Names = case Flags band 1 of
                              1 -> Names0#{IdRef => binary:part(FileBin, UOff, ULen)};
                              0 -> Names0
                          end,
                  case Flags band 4 of
                      4 -> {Names, Acc};            %% IdRef of a deleted writer: skipped
                      0 -> {Names, [{maps:get(IdRef, Names), Trunc =:= 1, Idx, Term,
                                     binary:part(FileBin, DOff, DLen)} | Acc]}
                  end
          end, {#{}, []}, Good),
    lists:reverse(Out).

%% encode_msgs([{Server, Msg, Slot}]) -> {MsgsBin, RangesBin} | {fallback, Server}: a whole batch for submit/4 -- the
%% records in order, and the range list that written events of more than two ranges refer to (empty for most batches:
%% submit/3 will do).
encode_msgs(Items) ->
    encode_msgs(Items, <<>>, <<>>).
encode_msgs([], Msgs, Ranges) -> {Msgs, Ranges};
encode_msgs([{Server, Msg, Slot} | Rest], Msgs, Ranges) ->
    case encode_msg(Server, Msg, Slot) of
        fallback -> {fallback, Server};
        {seqx, Record, Lower} ->
            encode_msgs(Rest, <<Msgs/binary, (Record(byte_size(Ranges) div 16))/binary>>, <<Ranges/binary, Lower/binary>>);
        Bin when is_binary(Bin) -> encode_msgs(Rest, <<Msgs/binary, Bin/binary>>, Ranges)
    end.

%% Slot = fun(ra_server_id()) -> 0..7 | 255, the member slot of a server id inside its group.
%% Returns the 64-byte record, or `fallback` for an event the batched path does not take (the caller runs
%% ra_server:handle_* itself and re-uploads the server's integers with upload_state/3).  Every message kind
%% of include/ra_gpu_batch.h has a clause; anything else (install_snapshot, cluster changes, ..) is not a
%% message of this path and must not be passed in.
encode_msg(Server, #append_entries_rpc{term = T, leader_id = L, leader_commit = LC,
                                       prev_log_index = PI, prev_log_term = PT,
                                       entries = Entries}, Slot) ->
    %% entries are contiguous; at most two term runs ride inline (payloads stay on the host)
    {N, N0, T0, T1, Gap} = entry_runs(PI, Entries),
    <<Server:32/little, ?MSG_AER:8, (Slot(L)):8, 0:8, Gap:8, T:64/little, PI:64/little,
      PT:64/little, LC:64/little, N:32/little, N0:32/little, T0:64/little, T1:64/little>>;
encode_msg(Server, {Peer, #append_entries_reply{term = T, success = S, next_index = NI,
                                                 last_index = LI, last_term = LT}}, Slot) ->
    Flags = case S of true -> 1; false -> 0 end,
    <<Server:32/little, ?MSG_AER_REPLY:8, (Slot(Peer)):8, Flags:8, 0:8, T:64/little,
      NI:64/little, LI:64/little, (term_or_undef(LT)):64/little, 0:64, 0:64, 0:64>>;
encode_msg(Server, #request_vote_rpc{term = T, candidate_id = C, last_log_index = LLI,
                                     last_log_term = LLT}, Slot) ->
    <<Server:32/little, ?MSG_REQUEST_VOTE:8, (Slot(C)):8, 0:8, 0:8, T:64/little, LLI:64/little,
      LLT:64/little, 0:64, 0:64, 0:64, 0:64>>;
encode_msg(Server, {ra_log_event, {written, T, Seq}}, _Slot) ->
    %% the written event carries a ra_seq:state() -- a list of indexes and ranges, newest first
    %% (src/ra_log_wal.erl:807, src/ra_log.erl:74,1641; ra_seq.erl:8-12) -- e.g. [{From,To}], [Idx],
    %% {written,0,[0]} or, after ra_log:write_sparse/3, something like [14, {2,9}].  One range, or two
    %% (RGB_MF_SEQ2 = 8: the lower one rides in the run0_term/run1_term fields), fit the record.  A longer
    %% sequence (ABI v8, RGB_MF_SEQX = 32) keeps its two highest ranges in the record and names the others in
    %% the batch's range list: {seqx, Record, LowerRanges} -- the record's `c` (first list entry) is filled in
    %% by encode_msgs/2, which knows where the batch's list stands.  No written event falls back any more.
    case seq_ranges(Seq) of
        [{From, To}] ->
            <<Server:32/little, ?MSG_WRITTEN:8, ?NONE:8, 0:8, 0:8, T:64/little, From:64/little,
              To:64/little, 0:64, 0:64, 0:64, 0:64>>;
        [{From2, To2}, {From, To}] ->
            <<Server:32/little, ?MSG_WRITTEN:8, ?NONE:8, 8:8, 0:8, T:64/little, From:64/little,
              To:64/little, 0:64, 0:64, From2:64/little, To2:64/little>>;
        Ranges ->
            {Lower, [{From2, To2}, {From, To}]} = lists:split(length(Ranges) - 2, Ranges),
            {seqx,
             fun(C) ->
                     <<Server:32/little, ?MSG_WRITTEN:8, ?NONE:8, (8 bor 32):8, 0:8, T:64/little, From:64/little,
                       To:64/little, C:64/little, (length(Lower)):32/little, 0:32, From2:64/little, To2:64/little>>
             end,
             << <<F:64/little, L:64/little>> || {F, L} <- Lower >>}
    end;
encode_msg(Server, {Peer, #request_vote_result{term = T, vote_granted = G}}, Slot) ->
    Flags = case G of true -> 1; false -> 0 end,
    <<Server:32/little, ?MSG_VOTE_RESULT:8, (Slot(Peer)):8, Flags:8, 0:8, T:64/little, 0:(6 * 64)>>;
encode_msg(Server, #pre_vote_rpc{version = V, machine_version = MV, term = T, token = Tok, candidate_id = C,
                                 last_log_index = LLI, last_log_term = LLT}, Slot) ->
    %% the token is a reference in the reference; the caller maps it to a 64-bit integer (erlang:phash2 of the
    %% ref is NOT unique enough: keep a per-server counter and a map Ref -> integer beside the gen_statem)
    <<Server:32/little, ?MSG_PRE_VOTE_RPC:8, (Slot(C)):8, 0:8, V:8, T:64/little, LLI:64/little,
      LLT:64/little, Tok:64/little, MV:32/little, 0:32, 0:64, 0:64>>;
encode_msg(Server, {Peer, #pre_vote_result{term = T, token = Tok, vote_granted = G}}, Slot) ->
    Flags = case G of true -> 1; false -> 0 end,
    <<Server:32/little, ?MSG_PRE_VOTE_RESULT:8, (Slot(Peer)):8, Flags:8, 0:8, T:64/little, 0:64, 0:64,
      Tok:64/little, 0:64, 0:64, 0:64>>;
encode_msg(Server, {election_timeout, Tok}, _Slot) ->
    %% election_timeout in follower / pre_vote / candidate; Tok = the integer standing for make_ref() of
    %% call_for_election(pre_vote, ..) (src/ra_server.erl:2900-2924)
    <<Server:32/little, ?MSG_ELECTION_TIMEOUT:8, ?NONE:8, 0:8, 0:8, 0:64, 0:64, 0:64, Tok:64/little,
      0:64, 0:64, 0:64>>;
encode_msg(Server, {commands, NEntries, Force}, _Slot) ->
    %% {command, _} / {commands, _} on the leader: only the COUNT crosses (payloads go to ra_log:append on the
    %% host, src/ra_server.erl:653-738); Force = true for the noop of a new term (RGB_MF_FORCE = 2)
    Flags = case Force of true -> 2; false -> 0 end,
    <<Server:32/little, ?MSG_APPEND:8, ?NONE:8, Flags:8, 0:8, 0:64, 0:64, 0:64, 0:64,
      NEntries:32/little, 0:32, 0:64, 0:64>>;
encode_msg(Server, await_condition_timeout, _Slot) ->
    <<Server:32/little, ?MSG_AWAIT_TIMEOUT:8, ?NONE:8, 0:8, 0:8, 0:(7 * 64)>>;
encode_msg(Server, {ra_log_event, {snapshot_written, {Idx, T}, _, snapshot, _, _}}, _Slot) ->
    <<Server:32/little, ?MSG_SNAPSHOT_WRITTEN:8, ?NONE:8, 0:8, 0:8, 0:64, Idx:64/little,
      T:64/little, 0:64, 0:64, 0:64, 0:64>>;
encode_msg(Server, #heartbeat_rpc{term = T, leader_id = L, query_index = QI}, Slot) ->
    <<Server:32/little, ?MSG_HEARTBEAT_RPC:8, (Slot(L)):8, 0:8, 0:8, T:64/little, QI:64/little,
      0:(5 * 64)>>;
encode_msg(Server, {Peer, #heartbeat_reply{term = T, query_index = QI}}, Slot) ->
    <<Server:32/little, ?MSG_HEARTBEAT_REPLY:8, (Slot(Peer)):8, 0:8, 0:8, T:64/little,
      QI:64/little, 0:(5 * 64)>>;
encode_msg(Server, {consistent_query, _From, _Fun}, _Slot) ->
    %% only while cluster_change_permitted; the QueryRef is queued by the caller under the
    %% query index the decision returns (reply_last_term)
    <<Server:32/little, ?MSG_CONSISTENT_QUERY:8, ?NONE:8, 0:8, 0:8, 0:(7 * 64)>>;
encode_msg(Server, pipeline_rpcs, _Slot) ->
    <<Server:32/little, ?MSG_PIPELINE_RPCS:8, ?NONE:8, 0:8, 0:8, 0:(7 * 64)>>;
encode_msg(Server, tick_timeout, _Slot) ->
    %% leader(_, tick_timeout, _) -> make_rpcs/1 (src/ra_server_proc.erl:613-616): RGB_MF_TICK = 4
    <<Server:32/little, ?MSG_PIPELINE_RPCS:8, ?NONE:8, 4:8, 0:8, 0:(7 * 64)>>.

entry_runs(_PI, []) -> {0, 0, 0, 0, 0};
entry_runs(PI, [{I0, T0, _} | _] = Es) ->
    {Run0, Rest} = lists:splitwith(fun({_, T, _}) -> T =:= T0 end, Es),
    T1 = case Rest of [] -> 0; [{_, T, _} | _] -> T end,
    true = lists:all(fun({_, T, _}) -> T =:= T1 end, Rest), %% else: fall back to ra_server
    {length(Es), length(Run0), T0, T1, I0 - (PI + 1)}.

%% a ra_seq as ascending, merged {Lo, Hi} ranges
seq_ranges(Seq) ->
    lists:reverse(ra_seq:fold(fun (I, [{Lo, Hi} | R]) when I =:= Hi + 1 -> [{Lo, I} | R];
                                  (I, Acc) -> [{I, I} | Acc]
                              end, [], Seq)).

term_or_undef(undefined) -> ?UNDEF;
term_or_undef(T) -> T.

decode_decision(<<Server:32/little, Role:8, ReplyTo:8, NRpcs:8, Kind:8, Flags:32/little,
                  Inv:16/little, HbTo:8, Cancel:8, RT:64/little, RNI:64/little, RLI:64/little, RLT:64/little,
                  CI:64/little, LA:64/little>>) ->
    #{server => Server, role => role(Role), reply_to => ReplyTo, n_rpcs => NRpcs, kind => Kind,
      flags => Flags, invariant => Inv, heartbeat_to => HbTo, cancel_backoff => Cancel, reply_term => RT, reply_next_index => RNI,
      reply_last_index => RLI, reply_last_term => RLT, commit_index => CI, last_applied => LA}.

role(0) -> follower; role(1) -> candidate; role(2) -> leader; role(3) -> pre_vote;
role(4) -> await_condition.

%% Reconstitute the effects() list ra_server_proc:handle_effects/4 expects from a decision
%% (reference src/ra_server.erl:178-206 for the vocabulary).  Id = this server's id,
%% Member = fun(Slot) -> ra_server_id(); Member(first_peer) = the first key of the cluster map without Id
%% (only asked for with F_TRANSFER_LEADERSHIP).
decision_to_effects(Id, Member, #{flags := F} = D) when F band ?F_INVARIANT =/= 0 ->
    %% the reference would have exited: do exactly that (reason by code, see the header)
    exit({ra_gpu_batch_invariant, Id, maps:get(invariant, D), Member});
decision_to_effects(Id, Member, #{flags := F, reply_to := To} = D) ->
    Reply =
        if F band ?F_REPLY =:= 0 -> [];
           F band ?F_REPLY_VOTE =/= 0 ->
               [{reply, #request_vote_result{term = maps:get(reply_term, D),
                                             vote_granted = F band ?F_REPLY_SUCCESS =/= 0}}];
           F band ?F_REPLY_HEARTBEAT =/= 0 ->
               [{cast, Member(To),
                 {Id, #heartbeat_reply{term = maps:get(reply_term, D),
                                       query_index = maps:get(reply_next_index, D)}}}];
           F band ?F_REPLY_PRE_VOTE =/= 0 ->
               [{reply, {pre_vote_result, maps:get(reply_term, D), maps:get(reply_next_index, D),
                         F band ?F_REPLY_SUCCESS =/= 0}}];
           true ->
               [{cast, Member(To),
                 {Id, #append_entries_reply{term = maps:get(reply_term, D),
                                            success = F band ?F_REPLY_SUCCESS =/= 0,
                                            next_index = maps:get(reply_next_index, D),
                                            last_index = maps:get(reply_last_index, D),
                                            last_term = maps:get(reply_last_term, D)}}}]
        end,
    Heartbeats =
        [{send_rpc, Member(Slot),
          #heartbeat_rpc{term = maps:get(reply_term, D), leader_id = Id,
                         query_index = maps:get(reply_last_term, D)}}
         || F band ?F_SEND_HEARTBEATS =/= 0, Slot <- lists:seq(0, 7),
            maps:get(heartbeat_to, D) band (1 bsl Slot) =/= 0],
    %% F_QUERY_QUORUM: release queued queries with index =< reply_next_index (the caller owns
    %% queries_waiting_heartbeats); F_QUERY_APPLY: no peers, apply now; F_RESEND_PENDING:
    %% ra_log:resend_pending/2 on the host log
    Cancels =
        [{cancel_snapshot_retry_timer, Member(Slot)}
         || F band ?F_CANCEL_SNAPSHOT_RETRY =/= 0, Slot <- lists:seq(0, 7),
            maps:get(cancel_backoff, D) band (1 bsl Slot) =/= 0],
    Cancels ++ Reply ++ Heartbeats
    ++ [{record_leader_msg, Member(To)} || F band ?F_LEADER_MSG =/= 0]
    ++ [{next_event, info, pipeline_rpcs} || F band ?F_PIPELINE =/= 0]
    ++ [{aux, eval} || F band ?F_AUX_EVAL =/= 0]
    %% the leader's wal_down condition timed out with the WAL still down (ra_server.erl:660-668): Member(first_peer)
    %% must resolve to hd(maps:to_list(maps:remove(Id, Cluster))) -- the caller's Member fun knows the cluster map
    ++ [{next_event, cast, {transfer_leadership, Member(first_peer)}} || F band ?F_TRANSFER_LEADERSHIP =/= 0].
