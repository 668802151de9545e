#!/usr/bin/env escript
%%! -noshell
%% ra_server_oracle.escript -- EXECUTES the transcribed reference vectors against the reference itself.
%%
%% SURVEY.md section 8(c), last row ("bench/ra_server_oracle.escript").  tests/golden/ra_server_suite_vectors.json
%% holds what the reference's own tests assert, transcribed by hand; the CPU checker (oracle/ra_oracle.c) is pinned
%% to those transcriptions.  This script turns "pinned by transcription" into "pinned by execution" on any machine
%% that has Erlang/OTP (>= 27 for the json module) and a rabbitmq/ra checkout built with its test profile:
%%
%%   cd ra && make test-build          # or: rebar3 as test compile   (needs meck, compiled test/ra_log_memory.erl)
%%   ERL_LIBS=_build/test/lib escript /path/to/erlang/ra_server_oracle.escript \
%%       /path/to/tests/golden/ra_server_suite_vectors.json observed.jsonl  [path/to/ra/test]
%%   python tools/check_reference_run.py observed.jsonl        # diffs every observation against the expectations
%%
%% What it does, per vector: mocks ra_log with test/ra_log_memory.erl the way the suite's setup_log/0 does
%% (test/ra_server_SUITE.erl:172-248: every ra_log call the server makes is delegated to the in-memory log),
%% builds the starting state like the suite's fixtures (empty_state/2 :4139-4149, base_state/2 :4151-4192), applies
%% the vector's "tweak", feeds each step's message to ra_server:handle_<as>/2 and writes ONE JSON line per step with
%% what the reference returned (next state name, the integers of the state, the reply found in the effects, the
%% names of the other effects).  Nothing is asserted here: tools/check_reference_run.py compares.
%%
%% NOT RUN IN THIS REPOSITORY'S ENVIRONMENT: the image has no Erlang/OTP, so this file has never been compiled.
%% Steps it cannot drive are reported with "status":"skipped" and a reason (never silently passed): states that need
%% the internal await_condition predicate fun, snapshot_written log events (the in-memory log has no such event) and
%% the leader's tick (ra_server_proc's clause, not ra_server's).
-mode(compile).

-include_lib("ra/src/ra.hrl").
-include_lib("ra/src/ra_server.hrl").

main([VectorsPath, OutPath | Rest]) ->
    case Rest of
        [TestDir] -> code:add_patha(TestDir);
        [] -> ok
    end,
    {ok, Bin} = file:read_file(VectorsPath),
    #{<<"vectors">> := Vectors} = json:decode(Bin),
    {ok, Out} = file:open(OutPath, [write]),
    lists:foreach(fun (V) -> run_vector(V, Out) end, Vectors),
    ok = file:close(Out),
    io:format("~b vectors -> ~ts~n", [length(Vectors), OutPath]);
main(_) ->
    io:format("usage: ra_server_oracle.escript VECTORS.json OUT.jsonl [RA_TEST_EBIN_DIR]~n"),
    halt(2).

%% ---------------------------------------------------------------------------------------------------- mocks
%% every ra_log function ra_server calls is served by the in-memory log of the reference's own tests
setup_log() ->
    _ = (catch meck:unload()),
    ok = meck:new(ra_log, []),
    ok = meck:new(ra_snapshot, [passthrough]),
    ok = meck:new(ra_machine, [passthrough]),
    ok = meck:new(ra_log_meta, [passthrough]),
    Delegated = [{init, 1}, {recover_snapshot, 1}, {snapshot_state, 1}, {set_snapshot_state, 2},
                 {install_snapshot, 4}, {snapshot_index_term, 1}, {fold, 5}, {release_resources, 3},
                 {overview, 1}, {write_config, 2}, {next_index, 1}, {append, 2}, {write, 2}, {write_sparse, 3},
                 {handle_event, 2}, {last_written, 1}, {last_index_term, 1}, {set_last_index, 2},
                 {fetch_term, 2}, {update_release_cursor, 5}],
    lists:foreach(fun ({F, A}) -> meck:expect(ra_log, F, fun_of(ra_log_memory, F, A)) end, Delegated),
    meck:expect(ra_log, fold, fun (A, B, C, D, E, _) -> ra_log_memory:fold(A, B, C, D, E) end),
    meck:expect(ra_log, has_pending, fun (_) -> false end),
    meck:expect(ra_log, append_sync,
                fun ({Idx, Term, _} = E, L0) ->
                        L1 = ra_log_memory:append(E, L0),
                        {L, _} = ra_log_memory:handle_event({written, Term, [Idx]}, L1),
                        L
                end),
    meck:expect(ra_log, exists,
                fun ({Idx, Term}, L) ->
                        case ra_log_memory:fetch_term(Idx, L) of
                            {Term, Log} -> {true, Log};
                            {_, Log} -> {false, Log}
                        end
                end),
    meck:expect(ra_log_meta, store, fun (_, U, K, V) -> put({U, K}, V), ok end),
    meck:expect(ra_log_meta, store_sync, fun (_, U, K, V) -> put({U, K}, V), ok end),
    meck:expect(ra_log_meta, fetch, fun (_, U, K) -> get({U, K}) end),
    meck:expect(ra_log_meta, fetch, fun (_, U, K, D) -> ra_lib:default(get({U, K}), D) end),
    meck:expect(ra_snapshot, recovery_checkpoint, fun (_) -> undefined end),
    ok.

fun_of(M, F, 1) -> fun (A) -> M:F(A) end;
fun_of(M, F, 2) -> fun (A, B) -> M:F(A, B) end;
fun_of(M, F, 3) -> fun (A, B, C) -> M:F(A, B, C) end;
fun_of(M, F, 4) -> fun (A, B, C, D) -> M:F(A, B, C, D) end;
fun_of(M, F, 5) -> fun (A, B, C, D, E) -> M:F(A, B, C, D, E) end.

mock_machine(Mod) ->
    _ = (catch meck:unload(Mod)),
    meck:new(Mod, [non_strict]),
    meck:expect(Mod, init, fun (_) -> init_state end),
    meck:expect(Mod, apply, fun (_, Cmd, _) -> {Cmd, ok} end),
    ok.

%% ---------------------------------------------------------------------------------------------------- fixtures
sid(<<"n", N/binary>>) -> {list_to_atom("n" ++ binary_to_list(N)), node()};
sid(null) -> undefined.

name_of(undefined) -> null;
name_of({Name, _Node}) -> atom_to_binary(Name, utf8).

usr(Data) -> {'$usr', #{ts => 0}, Data, after_log_append}.

members(N) -> [sid(iolist_to_binary(["n", integer_to_list(I)])) || I <- lists:seq(1, N)].

%% empty_state/2 of the suite (:4139-4149)
empty_state(N, Self) ->
    {Name, _} = Self,
    ra_server:recover(
      ra_server:init(#{cluster_name => someid,
                       id => Self,
                       uid => atom_to_binary(Name, utf8),
                       initial_members => members(N),
                       log_init_args => #{uid => <<>>},
                       machine => {simple, fun (E, _) -> E end, <<>>}})).

%% base_state/2 of the suite (:4151-4192): log [1:1, 2:3, 3:5] all written, term 5, commit = applied = 3,
%% every peer next_index 4 / match_index 3, leader_id n1
base_state(N, Self) ->
    Log0 = lists:foldl(fun (E, L) -> ra_log:append(E, L) end,
                       ra_log:init(#{system_config => ra_system:default_config(), uid => <<>>}),
                       [{1, 1, usr(<<"hi1">>)}, {2, 3, usr(<<"hi2">>)}, {3, 5, usr(<<"hi3">>)}]),
    {Log, _} = ra_log:handle_event({written, 5, [{1, 3}]}, Log0),
    Peer = #{next_index => 4, match_index => 3, query_index => 0, status => normal,
             commit_index_sent => 0, voter_status => #{membership => voter}},
    Cluster = maps:from_list([{Id, Peer} || Id <- members(N)]),
    MacMod = ra_gpu_batch_vectors_machine,
    mock_machine(MacMod),
    {SelfName, _} = Self,
    Cfg = #cfg{id = Self,
               uid = atom_to_binary(SelfName, utf8),
               log_id = atom_to_binary(SelfName, utf8),
               metrics_key = SelfName,
               metrics_labels = #{},
               machine = {machine, MacMod, #{}},
               machine_version = 0,
               machine_versions = [{0, 0}],
               effective_machine_version = 0,
               effective_machine_module = MacMod,
               system_config = ra_system:default_config()},
    #{cfg => Cfg,
      leader_id => sid(<<"n1">>),
      cluster => Cluster,
      cluster_index_term => {0, 0},
      cluster_change_permitted => true,
      machine_state => <<"hi3">>,
      current_term => 5,
      commit_index => 3,
      last_applied => 3,
      log => Log,
      query_index => 0,
      queries_waiting_heartbeats => queue:new(),
      pending_consistent_queries => []}.

%% the vector's "tweak": the same overrides tests/vector_runner.py applies to the engine's state
apply_tweaks(State0, Tw, Tokens) ->
    maps:fold(fun (K, V, {S, Skip}) -> tweak(K, V, S, Skip, Tokens) end, {State0, []}, Tw).

tweak(<<"commit_index">>, V, S, Sk, _) -> {S#{commit_index => V}, Sk};
tweak(<<"last_applied">>, V, S, Sk, _) -> {S#{last_applied => V}, Sk};
tweak(<<"current_term">>, V, S, Sk, _) -> {S#{current_term => V}, Sk};
tweak(<<"votes">>, V, S, Sk, _) -> {S#{votes => V}, Sk};
tweak(<<"query_index">>, V, S, Sk, _) -> {S#{query_index => V}, Sk};
tweak(<<"voted_for">>, V, S, Sk, _) -> {S#{voted_for => sid(V)}, Sk};
tweak(<<"leader_id">>, V, S, Sk, _) -> {S#{leader_id => sid(V)}, Sk};
tweak(<<"pre_vote_token">>, V, S, Sk, Tokens) -> {S#{pre_vote_token => token(V, Tokens)}, Sk};
tweak(<<"role">>, _V, S, Sk, _) -> {S, Sk};       %% the step's "as" picks the handler
tweak(<<"machine_version">>, V, #{cfg := C} = S, Sk, _) -> {S#{cfg := C#cfg{machine_version = V}}, Sk};
tweak(<<"effective_machine_version">>, V, #{cfg := C} = S, Sk, _) ->
    {S#{cfg := C#cfg{effective_machine_version = V}}, Sk};
tweak(<<"self_nonvoter">>, true, #{cfg := #cfg{id = Id}, cluster := Cl} = S, Sk, _) ->
    P = maps:get(Id, Cl),
    {S#{cluster := Cl#{Id := P#{voter_status => #{membership => promotable, target => 0}}},
        membership => promotable}, Sk};
tweak(<<"self_nonvoter">>, _, S, Sk, _) -> {S, Sk};
tweak(<<"members_present">>, Names, #{cluster := Cl} = S, Sk, _) ->
    Keep = [sid(N) || N <- Names],
    {S#{cluster := maps:with(Keep, Cl)}, Sk};
tweak(<<"nonvoters">>, Names, #{cluster := Cl} = S, Sk, _) ->
    Cl1 = lists:foldl(fun (N, C) ->
                              P = maps:get(sid(N), C),
                              C#{sid(N) := P#{voter_status => #{membership => promotable, target => 0}}}
                      end, Cl, Names),
    {S#{cluster := Cl1}, Sk};
tweak(<<"peers_not_normal">>, Names, #{cluster := Cl} = S, Sk, _) ->
    Cl1 = lists:foldl(fun (N, C) -> P = maps:get(sid(N), C), C#{sid(N) := P#{status => disconnected}} end,
                      Cl, Names),
    {S#{cluster := Cl1}, Sk};
tweak(<<"peers_backoff">>, Names, #{cluster := Cl} = S, Sk, _) ->
    Cl1 = lists:foldl(fun (N, C) ->
                              P = maps:get(sid(N), C),
                              C#{sid(N) := P#{status => {snapshot_backoff, 1}}}
                      end, Cl, Names),
    {S#{cluster := Cl1}, Sk};
tweak(<<"peers">>, Peers, #{cluster := Cl} = S, Sk, _) ->
    Cl1 = maps:fold(fun (N, Fields, C) ->
                            P0 = maps:get(sid(N), C),
                            P = maps:fold(fun (F, Val, Acc) -> Acc#{binary_to_atom(F, utf8) => Val} end,
                                          P0, Fields),
                            C#{sid(N) := P}
                    end, Cl, Peers),
    {S#{cluster := Cl1}, Sk};
tweak(<<"log">>, Entries, S, Sk, _) ->
    %% [[Idx, Term], ..] including [0,0]: rebuild the in-memory log; "last_written" is applied below
    L0 = ra_log:init(#{system_config => ra_system:default_config(), uid => <<>>}),
    L = lists:foldl(fun ([0, _], Acc) -> Acc;
                        ([I, T], Acc) -> ra_log:append({I, T, usr(<<"x">>)}, Acc)
                    end, L0, Entries),
    {S#{log := L}, Sk};
tweak(<<"last_written">>, [I, T], #{log := L0} = S, Sk, _) ->
    {L, _} = ra_log:handle_event({written, T, [{0, I}]}, L0),
    {S#{log := L}, Sk};
tweak(<<"install_snapshot">>, [I, T], #{log := L0, cfg := #cfg{machine = _}} = S, Sk, _) ->
    Meta = #{index => I, term => T, cluster => [], machine_version => 0},
    case catch ra_log_memory:install_snapshot({I, T}, undefined, [], ra_log_memory:set_snapshot_state({Meta, []}, L0)) of
        {ok, L, _} -> {S#{log := L}, Sk};
        {ok, L} -> {S#{log := L}, Sk};
        Other -> {S, [{install_snapshot, Other} | Sk]}
    end;
tweak(K, _V, S, Sk, _) when K =:= <<"cond_reason">>; K =:= <<"cond_reply">>; K =:= <<"cond_leader">> ->
    %% the await_condition predicate is an internal fun of ra_server (follower_catchup_cond/3): such states
    %% can only be reached by replaying the rpc that produced them
    {S, [{needs_internal_condition_fun, K} | Sk]};
tweak(K, _V, S, Sk, _) -> {S, [{unknown_tweak, K} | Sk]}.

token(V, Tokens) ->
    case get({token, V}) of
        undefined -> R = make_ref(), put({token, V}, R), put({token_of, R}, V), _ = Tokens, R;
        R -> R
    end.
token_int(R) when is_reference(R) ->
    case get({token_of, R}) of undefined -> 0; V -> V end;
token_int(_) -> 0.

%% ---------------------------------------------------------------------------------------------------- messages
make_msg(#{<<"kind">> := <<"aer">>} = M) ->
    [PI, PT] = maps:get(<<"prev">>, M),
    {ok, #append_entries_rpc{term = maps:get(<<"term">>, M), leader_id = sid(maps:get(<<"from">>, M)),
                             leader_commit = maps:get(<<"commit">>, M), prev_log_index = PI, prev_log_term = PT,
                             entries = [{I, T, usr(<<"e">>)} || [I, T] <- maps:get(<<"entries">>, M)]}};
make_msg(#{<<"kind">> := <<"aer_reply">>} = M) ->
    {ok, {sid(maps:get(<<"from">>, M)),
          #append_entries_reply{term = maps:get(<<"term">>, M), success = maps:get(<<"success">>, M),
                                next_index = maps:get(<<"next_index">>, M),
                                last_index = maps:get(<<"last_index">>, M),
                                last_term = maps:get(<<"last_term">>, M)}}};
make_msg(#{<<"kind">> := <<"request_vote">>} = M) ->
    [LI, LT] = maps:get(<<"last">>, M),
    {ok, #request_vote_rpc{term = maps:get(<<"term">>, M), candidate_id = sid(maps:get(<<"from">>, M)),
                           last_log_index = LI, last_log_term = LT}};
make_msg(#{<<"kind">> := <<"vote_result">>} = M) ->
    {ok, #request_vote_result{term = maps:get(<<"term">>, M), vote_granted = maps:get(<<"granted">>, M)}};
make_msg(#{<<"kind">> := <<"written">>} = M) ->
    [A, B] = maps:get(<<"range">>, M),
    Seq = case A =:= B of true -> [A]; false -> [{A, B}] end,
    {ok, {ra_log_event, {written, maps:get(<<"term">>, M), Seq}}};
make_msg(#{<<"kind">> := <<"pre_vote_rpc">>} = M) ->
    [LI, LT] = maps:get(<<"last">>, M),
    {ok, #pre_vote_rpc{version = maps:get(<<"version">>, M), machine_version = maps:get(<<"machine_version">>, M),
                       term = maps:get(<<"term">>, M), token = token(maps:get(<<"token">>, M), x),
                       candidate_id = sid(maps:get(<<"from">>, M)), last_log_index = LI, last_log_term = LT}};
make_msg(#{<<"kind">> := <<"pre_vote_result">>} = M) ->
    {ok, #pre_vote_result{term = maps:get(<<"term">>, M), token = token(maps:get(<<"token">>, M), x),
                          vote_granted = maps:get(<<"granted">>, M)}};
make_msg(#{<<"kind">> := <<"heartbeat_rpc">>} = M) ->
    {ok, #heartbeat_rpc{query_index = maps:get(<<"query_index">>, M), term = maps:get(<<"term">>, M),
                        leader_id = sid(maps:get(<<"from">>, M))}};
make_msg(#{<<"kind">> := <<"heartbeat_reply">>} = M) ->
    {ok, {sid(maps:get(<<"from">>, M)),
          #heartbeat_reply{query_index = maps:get(<<"query_index">>, M), term = maps:get(<<"term">>, M)}}};
make_msg(#{<<"kind">> := <<"append">>} = M) ->
    N = maps:get(<<"n">>, M),
    Cmd = case maps:get(<<"force">>, M, false) of
              true -> {noop, #{ts => 0}, 0};
              false -> usr(<<"c">>)
          end,
    {ok, {commands, lists:duplicate(N, Cmd)}};
make_msg(#{<<"kind">> := <<"election_timeout">>}) -> {ok, election_timeout};
make_msg(#{<<"kind">> := <<"await_timeout">>}) -> {ok, await_condition_timeout};
make_msg(#{<<"kind">> := <<"pipeline_rpcs">>}) -> {ok, pipeline_rpcs};
make_msg(#{<<"kind">> := <<"consistent_query">>}) ->
    {ok, {consistent_query, {self(), make_ref()}, fun (S) -> S end}};
make_msg(#{<<"kind">> := K}) -> {skip, K}.

%% ---------------------------------------------------------------------------------------------------- driver
run_vector(#{<<"id">> := Id, <<"n_members">> := N, <<"self">> := SelfB, <<"init">> := Init,
             <<"steps">> := Steps} = V, Out) ->
    try
        ok = setup_log(),
        erase(),
        Self = sid(SelfB),
        S0 = case Init of
                 <<"empty">> -> empty_state(N, Self);
                 <<"base">> ->
                     B = base_state(N, Self),
                     #{cfg := C} = B,
                     {SelfName, _} = Self,
                     B#{cfg := C#cfg{id = Self, uid = atom_to_binary(SelfName, utf8)}}
             end,
        {S1, Skipped} = apply_tweaks(S0, maps:get(<<"tweak">>, V, #{}), x),
        run_steps(Id, Steps, S1, S1, Skipped, 0, Out)
    catch
        Class:Reason:Stack ->
            emit(Out, #{id => Id, step => -1, status => <<"error">>,
                        reason => iolist_to_binary(io_lib:format("~p:~p ~p", [Class, Reason, hd(Stack)]))})
    end.

run_steps(_Id, [], _S, _Init, _Skipped, _N, _Out) -> ok;
run_steps(Id, [#{<<"as">> := As, <<"msg">> := M} = Step | Rest], S0, Init, Skipped, N, Out) ->
    S = case maps:get(<<"reset">>, Step, false) of true -> Init; _ -> S0 end,
    Next =
        case {Skipped, make_msg(M)} of
            {[_ | _], _} ->
                emit(Out, #{id => Id, step => N, status => <<"skipped">>,
                            reason => iolist_to_binary(io_lib:format("~p", [Skipped]))}),
                S;
            {[], {skip, K}} ->
                emit(Out, #{id => Id, step => N, status => <<"skipped">>,
                            reason => <<"message kind not driven by this harness: ", K/binary>>}),
                S;
            {[], {ok, Msg}} ->
                Handler = binary_to_atom(<<"handle_", As/binary>>, utf8),
                try ra_server:Handler(Msg, S) of
                    {Role, S1, Effects} ->
                        emit(Out, observe(Id, N, Role, S1, Effects)),
                        S1;
                    {Role, S1, Effects, _Actions} ->
                        emit(Out, observe(Id, N, Role, S1, Effects)),
                        S1
                catch
                    exit:Reason ->
                        emit(Out, #{id => Id, step => N, status => <<"exit">>,
                                    reason => iolist_to_binary(io_lib:format("~p", [Reason]))}),
                        S;
                    error:Reason:St ->
                        emit(Out, #{id => Id, step => N, status => <<"error">>,
                                    reason => iolist_to_binary(io_lib:format("~p ~p", [Reason, hd(St)]))}),
                        S
                end
        end,
    %% a "fork" step (a what-if branch of the suite's test) does not carry its state forward
    Carry = case maps:get(<<"fork">>, Step, false) of true -> S; _ -> Next end,
    run_steps(Id, Rest, Carry, Init, Skipped, N + 1, Out).

observe(Id, N, Role, #{log := Log} = S, Effects) ->
    {LI, LT} = ra_log:last_index_term(Log),
    {LWI, LWT} = ra_log:last_written(Log),
    St = #{current_term => maps:get(current_term, S), commit_index => maps:get(commit_index, S),
           last_applied => maps:get(last_applied, S), leader_id => name_of(maps:get(leader_id, S, undefined)),
           voted_for => name_of(maps:get(voted_for, S, undefined)), votes => maps:get(votes, S, 0),
           last_index => LI, last_term => LT, last_written => [LWI, LWT],
           query_index => maps:get(query_index, S, 0),
           pre_vote_token => token_int(maps:get(pre_vote_token, S, undefined))},
    Peers = maps:fold(fun (PId, P, Acc) ->
                              Acc#{name_of(PId) => maps:with([match_index, next_index, commit_index_sent,
                                                               query_index], P)}
                      end, #{}, maps:get(cluster, S, #{})),
    #{id => Id, step => N, status => <<"ok">>, role => atom_to_binary(Role, utf8), state => St, peers => Peers,
      reply => reply_of(Effects), rpcs => rpcs_of(Effects),
      effects => [effect_name(E) || E <- Effects]}.

reply_of(Effects) ->
    lists:foldl(
      fun ({cast, To, {_Id, #append_entries_reply{term = T, success = Su, next_index = NI, last_index = LI,
                                                  last_term = LT}}}, _) ->
              #{to => name_of(To), term => T, success => Su, next_index => NI, last_index => LI, last_term => LT};
          ({reply, #request_vote_result{term = T, vote_granted = G}}, _) ->
              #{vote => true, term => T, success => G};
          ({reply, #pre_vote_result{term = T, token = Tok, vote_granted = G}}, _) ->
              #{pre_vote => true, term => T, success => G, token => token_int(Tok)};
          ({cast, To, {_Id, #heartbeat_reply{term = T, query_index = QI}}}, _) ->
              #{heartbeat => true, to => name_of(To), term => T, query_index => QI};
          (_, Acc) -> Acc
      end, null, Effects).

rpcs_of(Effects) ->
    [#{peer => name_of(To), term => T, prev_log_index => PI, prev_log_term => PT, leader_commit => LC,
       n_entries => length(Es)}
     || {send_rpc, To, #append_entries_rpc{term = T, prev_log_index = PI, prev_log_term = PT,
                                           leader_commit = LC, entries = Es}} <- Effects].

effect_name(E) when is_tuple(E) -> atom_to_binary(element(1, E), utf8);
effect_name(E) when is_atom(E) -> atom_to_binary(E, utf8);
effect_name(_) -> <<"?">>.

%% ---------------------------------------------------------------------------------------------------- output
emit(Out, Map) -> io:put_chars(Out, [json:encode(jsonable(Map)), $\n]).

jsonable(M) when is_map(M) -> maps:fold(fun (K, V, A) -> A#{jkey(K) => jsonable(V)} end, #{}, M);
jsonable(L) when is_list(L) -> [jsonable(X) || X <- L];
jsonable(true) -> true; jsonable(false) -> false; jsonable(null) -> null;
jsonable(undefined) -> null;
jsonable(A) when is_atom(A) -> atom_to_binary(A, utf8);
jsonable(X) -> X.

jkey(K) when is_atom(K) -> atom_to_binary(K, utf8);
jkey(K) -> K.
