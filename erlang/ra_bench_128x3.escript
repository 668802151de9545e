#!/usr/bin/env escript
%%! -name ra_bench_driver@127.0.0.1 -setcookie ra_bench +S 64
%%
%% BASELINE.json configs[0]: the reference's own CPU path -- 128 Raft groups x 3 members with the
%% `{simple, fun erlang:'+'/2, 0}` machine (README.md:121-130) on ONE BEAM node, pure Erlang.
%%
%% NOT RUN IN THIS REPOSITORY'S ENVIRONMENT: there is no Erlang/OTP here (DESIGN.md section 4,
%% BASELINE.md section 2), so the number this prints is reported as "not measured".  It is the
%% driver a maintainer runs on a machine with OTP >= 26 and a built rabbitmq/ra checkout:
%%
%%     cd ra && make            # or rebar3 compile
%%     ERL_LIBS=_build/default/lib escript /path/to/ra_bench_128x3.escript [Seconds] [PipelineDepth]
%%
%% The shipped ra_bench (src/ra_bench.erl:138-152) starts ONE cluster with its own noop machine, so
%% this driver loops ra:start_cluster/4 128 times (three local members each, as the README's
%% quick start does with ra:start_local_cluster/3) and drives every group's leader with one
%% pipelining client, the way ra_bench:client_loop/5 (src/ra_bench.erl:173-189) does: keep
%% PipelineDepth commands in flight, top up as {applied, _} events arrive.
%%
%% What it measures, for comparison with bench.py: every applied command costs the group one
%% leader append, N-1 follower append_entries decisions, N-1 written events, N-1 leader reply
%% decisions (+ quorum) -- about 3(N-1)+1 = 7 "decisions" in bench.py's unit per applied command
%% at N = 3.  It prints applied commands/s, that figure x 7, and the scheduler count.
-mode(compile).

-define(GROUPS, 128).
-define(MEMBERS, 3).

main(Args) ->
    Secs = case Args of [S | _] -> list_to_integer(S); _ -> 30 end,
    Pipe = case Args of [_, P | _] -> list_to_integer(P); _ -> 500 end,
    {ok, _} = application:ensure_all_started(ra),
    _ = ra_system:start_default(),
    Machine = {simple, fun erlang:'+'/2, 0},
    Leaders =
        [begin
             Name = list_to_atom("g" ++ integer_to_list(G)),
             ServerIds = [{list_to_atom(atom_to_list(Name) ++ "_" ++ integer_to_list(M)), node()}
                          || M <- lists:seq(1, ?MEMBERS)],
             {ok, Started, []} = ra:start_cluster(default, Name, Machine, ServerIds),
             {ok, _, Leader} = ra:members(hd(Started)),
             Leader
         end || G <- lists:seq(1, ?GROUPS)],
    Counter = counters:new(1, [write_concurrency]),
    Parent = self(),
    Clients = [spawn_link(fun () -> client(Parent, L, Pipe, Counter) end) || L <- Leaders],
    T0 = erlang:monotonic_time(millisecond),
    [C ! go || C <- Clients],
    timer:sleep(Secs * 1000),
    Applied = counters:get(Counter, 1),
    T1 = erlang:monotonic_time(millisecond),
    [begin unlink(C), exit(C, kill) end || C <- Clients],
    Rate = Applied / ((T1 - T0) / 1000),
    io:format("ra reference path, ~b groups x ~b members on one BEAM node, ~b schedulers~n"
              "applied commands: ~b in ~b ms = ~.1f commands/s~n"
              "~~ ~.1f append_entries decisions/s in bench.py's unit (x ~b per command)~n",
              [?GROUPS, ?MEMBERS, erlang:system_info(schedulers_online), Applied, T1 - T0, Rate,
               Rate * (3 * (?MEMBERS - 1) + 1), 3 * (?MEMBERS - 1) + 1]),
    halt(0).

%% one pipelining client per group (src/ra_bench.erl:160-199)
client(_Parent, Leader, Pipe, Counter) ->
    receive go -> ok end,
    [ra:pipeline_command(Leader, 1, make_ref(), normal) || _ <- lists:seq(1, Pipe)],
    client_loop(Leader, Counter).

client_loop(Leader0, Counter) ->
    receive
        {ra_event, Leader, {applied, Applied}} ->
            N = length(Applied),
            counters:add(Counter, 1, N),
            [ra:pipeline_command(Leader, 1, make_ref(), normal) || _ <- lists:seq(1, N)],
            client_loop(Leader, Counter);
        {ra_event, _, {rejected, {not_leader, NewLeader, _}}} when NewLeader =/= undefined ->
            ra:pipeline_command(NewLeader, 1, make_ref(), normal),
            client_loop(NewLeader, Counter);
        {ra_event, _, _} ->
            client_loop(Leader0, Counter)
    end.
