#!/usr/bin/env python3
"""Known-answer vectors for the ra_gpu_batch hot path, TRANSCRIBED BY HAND from the reference's
own unit tests (rabbitmq/ra app vsn 3.1.10):

    test/ra_server_SUITE.erl      (state-in / message-in / state-out / effects-out cases)
    src/ra_server.erl:4225-4238   (agreed_commit_test)
    test/ra_log_2_SUITE.erl       (real ra_log last_written cursor cases)

The reference is Erlang and cannot be executed in this environment (no OTP), so these vectors
are transcriptions of what its tests ASSERT, not outputs of running it.  Only facts the
reference test asserts are recorded as expectations; everything else is left unconstrained.
Running this script rewrites tests/golden/ra_server_suite_vectors.json.

Vector format
-------------
  n_members  cluster size; member names n1..n7 map to member slots 0..6
  self       name of the server under test
  init       "empty" = empty_state/2 (SUITE:4139-4149); "base" = base_state/2 (SUITE:4151-4192):
             CT=5 CI=3 LA=3 leader=n1 log [0:0,1:1,2:3,3:5] all written, every member NI=4 MI=3
  tweak      overrides applied to the initial state (see tests/vector_runner.py)
  log_model  "real": authored against src/ra_log.erl; "mem": authored against the fake
             test/ra_log_memory.erl (asserted facts hold for the real log too)
  steps      list of {as: ra_state the reference test calls handle_<as>/2 in,
                      msg: message, expect: asserted facts}
  msg kinds  aer / aer_reply / request_vote / vote_result / written / pipeline_rpcs /
             append / await_timeout / election_timeout / pre_vote_rpc / pre_vote_result /
             snapshot_written
  expect     role; state{field: value}; peers{name:{next_index,match_index}};
             reply{...} (only the asserted fields) or no_reply; flags_set / flags_clear
             (names of RGB_F_*); rpcs (exact set when rpcs_exact) each {peer, prev:[i,t],
             commit, entries:[from,to]}
"""
import json
import os

V = []


def vec(id, source, n_members, self, init, steps, tweak=None, log_model="mem", note=None):
    d = dict(id=id, source=source, n_members=n_members, self=self, init=init,
             log_model=log_model, steps=steps)
    if tweak:
        d["tweak"] = tweak
    if note:
        d["note"] = note
    V.append(d)


def aer(term, leader, prev, commit, entries=()):
    return dict(kind="aer", term=term, **{"from": leader}, prev=list(prev), commit=commit,
                entries=[list(e) for e in entries])


def reply(peer, term, success, next_index, last_index, last_term):
    return dict(kind="aer_reply", term=term, **{"from": peer}, success=success,
                next_index=next_index, last_index=last_index, last_term=last_term)


def written(term, lo, hi):
    return dict(kind="written", term=term, range=[lo, hi])


def req_vote(term, cand, last):
    return dict(kind="request_vote", term=term, **{"from": cand}, last=list(last))


def vote_result(term, granted, voter="n2"):
    return dict(kind="vote_result", term=term, granted=granted, **{"from": voter})


def pre_vote(term, cand, last, token=77, machine_version=0, version=1):
    return dict(kind="pre_vote_rpc", term=term, **{"from": cand}, last=list(last), token=token,
                machine_version=machine_version, version=version)


def pre_vote_result(term, granted, token, voter="n2"):
    return dict(kind="pre_vote_result", term=term, granted=granted, token=token, **{"from": voter})


def step(as_, msg, **expect):
    return {"as": as_, "msg": msg, "expect": expect}


# ---------------------------------------------------------------- A.2 follower AER ----
vec("F1", "test/ra_server_SUITE.erl:383-456 follower_aer_1", 3, "n1", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=0, last_applied=0)),
    step("follower", aer(1, "n1", (1, 1), 1, [(2, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1)),
    step("follower", written(1, 1, 1), role="follower",
         state=dict(commit_index=1, last_applied=1),
         reply=dict(to="n1", next_index=3, last_term=1, last_index=1), effects_only_reply=True),
    step("follower", aer(1, "n1", (2, 1), 3, [(3, 1)]), role="follower",
         state=dict(commit_index=3, last_applied=3)),
    step("follower", written(1, 2, 2), role="follower", state=dict(commit_index=3, last_applied=3),
         reply=dict(to="n1", next_index=4, last_term=1, last_index=2), effects_only_reply=True),
    step("follower", aer(1, "n1", (3, 1), 3, []), role="follower",
         state=dict(commit_index=3, last_applied=3),
         reply=dict(to="n1", next_index=4, last_term=1, last_index=2)),
    step("follower", written(1, 3, 3), role="follower", state=dict(commit_index=3, last_applied=3),
         reply=dict(to="n1", next_index=4, last_term=1, last_index=3), effects_only_reply=True),
])

vec("F2", "test/ra_server_SUITE.erl:459-489 follower_aer_2", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=0, last_applied=0)),
    step("follower", written(1, 1, 1), role="follower", state=dict(commit_index=0, last_applied=0),
         reply=dict(to="n1", next_index=2, last_term=1, last_index=1), effects_only_reply=True),
    step("follower", aer(1, "n1", (1, 1), 1, []), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1)),
])

vec("F3", "test/ra_server_SUITE.erl:491-556 follower_aer_3", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 1, [(1, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1)),
    step("follower", written(1, 1, 1), role="follower", state=dict(commit_index=1, last_applied=1),
         reply=dict(to="n1", next_index=2, last_term=1, last_index=1), effects_only_reply=True),
    step("follower", aer(1, "n1", (2, 1), 3, [(3, 1)]), role="await_condition",
         state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1),
         reply=dict(to="n1", success=False, next_index=2, last_term=1, last_index=1),
         flags_set=["LEADER_MSG"]),
    # the reference test feeds the await_condition state straight to handle_follower/2
    step("follower", aer(1, "n1", (1, 1), 3, [(2, 1), (3, 1), (4, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=3, last_applied=3)),
    step("follower", written(1, 4, 4), role="follower", state=dict(commit_index=3, last_applied=3),
         reply=dict(to="n1", success=True, next_index=5, last_term=1, last_index=4)),
    step("follower", aer(1, "n1", (1, 1), 4, [(2, 1), (3, 1), (4, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=4, last_applied=4)),
])

vec("F4", "test/ra_server_SUITE.erl:561-588 follower_aer_4", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 10, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=10, last_applied=4)),
    step("follower", written(1, 4, 4), role="follower",
         state=dict(commit_index=10, last_applied=4),
         reply=dict(to="n1", next_index=5, last_term=1, last_index=4)),
], note="commit_index is NOT clamped to the log: CI=10, LA=4")

vec("F5", "test/ra_server_SUITE.erl:590-620 follower_aer_5", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 2, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower"),
    step("follower", written(1, 4, 4), role="follower"),
    step("follower", aer(2, "n5", (3, 1), 3, []), role="follower",
         reply=dict(to="n5", next_index=4, last_term=1, last_index=3), flags_set=["TRUNCATED"]),
])

vec("F6", "test/ra_server_SUITE.erl:622-656 follower_aer_6", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 3, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower"),
    step("follower", written(1, 4, 4), role="follower", state=dict(last_applied=3)),
    step("follower", aer(2, "n5", (3, 1), 3, []), role="follower",
         reply=dict(to="n5", next_index=4, last_term=1, last_index=3)),
])

vec("F7", "test/ra_server_SUITE.erl:658-697 follower_aer_7", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 3, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower"),
    step("follower", written(1, 4, 4), role="follower", state=dict(last_applied=3)),
    step("follower", aer(2, "n5", (3, 1), 4, [(4, 2)]), role="follower"),
    step("follower", written(2, 4, 4), role="follower", state=dict(last_applied=4),
         reply=dict(to="n5", next_index=5, last_term=2, last_index=4)),
])

vec("Fdup", "test/ra_server_SUITE.erl:1292-1323 follower_aer_dupe", 3, "n1", "empty", [
    step("follower", aer(1, "n2", (0, 0), 1, [(1, 1), (2, 1), (3, 1)]), role="follower",
         state=dict(leader_id="n2", current_term=1, commit_index=1, last_applied=1)),
    step("follower", aer(1, "n2", (1, 1), 1, [(2, 1)]), role="follower",
         state=dict(leader_id="n2", current_term=1, commit_index=1, last_applied=1),
         reply=dict(to="n2", success=True, next_index=3, last_term=1, last_index=2),
         effects_only_reply=True),
], note="not-validated branch: success reply up to max(last_applied, last valid index)")

vec("Fchg", "test/ra_server_SUITE.erl:1325-1368 follower_leader_change_before_written",
    3, "n3", "empty", [
        step("follower", aer(1, "n1", (0, 0), 1, [(1, 1), (2, 1)]), role="follower",
             state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1)),
        step("follower", aer(2, "n2", (0, 0), 1, [(2, 2), (3, 2)]), role="follower",
             state=dict(leader_id="n2", current_term=2, commit_index=1, last_applied=1)),
        step("follower", written(1, 1, 2), role="follower", state=dict(leader_id="n2", last_applied=1),
             reply=dict(to="n2", success=True, term=2, last_index=1, last_term=1),
             effects_only_reply=True),
        step("follower", written(2, 2, 3), role="follower", state=dict(leader_id="n2", last_applied=1),
             reply=dict(to="n2", success=True, term=2, last_index=3, last_term=2),
             effects_only_reply=True),
    ])

# ------------------------------------- A.3 follower AER: term checks, mismatch, snapshot ----
BASE_CI1 = dict(commit_index=1)
vec("T1-T5", "test/ra_server_SUITE.erl:854-892 follower_handles_append_entries_rpc", 3, "n1", "base",
    [
        dict(reset=True, **step("follower", aer(5, "n1", (3, 5), 3, []), role="follower",
                                 state=dict(leader_id="n1", current_term=5))),
        dict(reset=True, **step("follower", aer(6, "n1", (3, 5), 3, []), role="follower",
                                 state=dict(current_term=6),
                                 reply=dict(to="n1", term=6, success=True, next_index=4,
                                            last_index=3, last_term=5))),
        dict(reset=True, **step("follower", aer(4, "n1", (3, 5), 3, []), role="follower",
                                 reply=dict(to="n1", term=5, success=False),
                                 effects_only_reply=True)),
        dict(reset=True, **step("follower", aer(5, "n1", (4, 5), 3, []), role="await_condition",
                                 reply=dict(to="n1", term=5, success=False),
                                 flags_set=["LEADER_MSG"])),
        dict(reset=True, **step("follower", aer(5, "n1", (3, 4), 3, []), role="await_condition",
                                 reply=dict(to="n1", term=5, success=False),
                                 flags_set=["LEADER_MSG"])),
    ], tweak=BASE_CI1, note="each step starts again from the tweaked base state (reset)")

vec("T6", "test/ra_server_SUITE.erl:747-765 follower_aer_term_mismatch", 3, "n1", "base", [
    step("follower", aer(6, "n1", (3, 6), 3, []), role="await_condition",
         reply=dict(to="n1", term=6, success=False, next_index=3, last_index=2, last_term=3)),
], tweak=dict(last_applied=2, commit_index=3))

vec("T7", "test/ra_server_SUITE.erl:818-852 follower_aer_term_mismatch_snapshot", 3, "n1", "base", [
    step("follower", aer(6, "n1", (3, 6), 3, []), role="await_condition",
         reply=dict(to="n1", term=6, success=False, next_index=4, last_index=3, last_term=5)),
], tweak=dict(last_applied=3, commit_index=3, install_snapshot=[3, 5]),
    note="term of last_applied is served by the snapshot")

vec("T8", "test/ra_server_SUITE.erl:767-816 follower_aer_term_mismatch_at_snapshot", 3, "n1", "base", [
    step("follower", aer(5, "n1", (3, 5), 3, [(4, 5), (5, 5), (6, 5)]), role="follower"),
    step("follower", written(5, 4, 6), role="follower",
         reply=dict(to="n1", term=5, success=True, next_index=7)),
    step("follower", aer(6, "n2", (3, 5), 3, []), role="follower",
         state=dict(last_applied=3, commit_index=3),
         reply=dict(to="n2", term=6, success=True, next_index=4, last_index=3, last_term=5)),
], tweak=dict(last_applied=3, commit_index=3, install_snapshot=[3, 5]))

vec("T9", "test/ra_server_SUITE.erl:699-745 follower_aer_diverged", 3, "n1", "base", [
    step("follower", aer(6, "n1", (1, 1), 3, [(2, 3)]), role="follower",
         state=dict(last_applied=2, commit_index=2),
         reply=dict(to="n1", success=True, next_index=3)),
    step("follower", aer(6, "n1", (3, 6), 3, []), role="await_condition",
         reply=dict(to="n1", success=False, next_index=3)),
    step("follower", aer(6, "n1", (2, 3), 3, [(3, 6)]), role="follower",
         state=dict(last_applied=3, commit_index=3), no_reply=True,
         flags_set=["AUX_EVAL", "LEADER_MSG"]),
], tweak=dict(last_applied=2, commit_index=2))

vec("T10", "test/ra_server_SUITE.erl:894-912 follower_handles_append_entries_rpc (overwrite)",
    3, "n1", "base", [
        step("follower", aer(5, "n1", (1, 1), 2, [(2, 4)]), role="follower"),
        step("follower", written(4, 2, 2), role="follower",
             reply=dict(to="n1", term=5, success=True, next_index=3, last_index=2, last_term=4),
             state=dict(log=[[0, 0], [1, 1], [2, 4]])),
    ], tweak=dict(commit_index=1, last_applied=1))

vec("T11", "test/ra_server_SUITE.erl:914-931 follower_handles_append_entries_rpc (commit ahead)",
    3, "n1", "base", [
        step("follower", aer(5, "n1", (3, 5), 5, [(4, 5)]), role="follower"),
        step("follower", written(5, 4, 4), role="follower",
             state=dict(commit_index=5, last_applied=4),
             reply=dict(to="n1", term=5, success=True, last_index=4, last_term=5)),
    ], tweak=dict(commit_index=1, last_applied=1))

vec("T12", "test/ra_server_SUITE.erl:3147-3194 snapshotted_follower_received_append_entries",
    3, "n3", "empty", [
        step("follower", aer(2, "n1", (3, 2), 4, [(4, 2)]), role="follower"),
        step("follower", written(2, 4, 4), role="follower",
             reply=dict(to="n1", success=True), effects_only_reply=True),
    ], tweak=dict(current_term=2, install_snapshot=[3, 2], last_applied=3, commit_index=3,
                  leader_id="n1"),
    note="prev_log matched through the snapshot fallback")

# ------------------------------------------- A.4 await_condition catch-up predicate ----
AWAIT_TWEAK = dict(commit_index=1, role="await_condition", cond_reason="missing",
                   cond_reply=[5, 4, 3, 5], cond_leader="n1")
vec("A4", "test/ra_server_SUITE.erl:934-1001 follower_catchup_condition", 3, "n1", "base", [
    dict(reset=True, **step("follower", aer(4, "n1", (4, 5), 3, []), role="follower",
                             reply=dict(success=False), effects_only_reply=True)),
    dict(reset=True, **step("follower", aer(6, "n1", (3, 4), 3, []), role="await_condition",
                             reply=dict(success=False), flags_set=["LEADER_MSG"])),
    dict(reset=True, **step("await_condition", aer(5, "n1", (4, 5), 3, []), role="await_condition",
                             no_reply=True, flags_clear=["LEADER_MSG", "REPROCESSED", "PERSIST"])),
    dict(reset=True, **step("await_condition", aer(5, "n1", (3, 5), 3, []), role="follower",
                             flags_set=["REPROCESSED"])),
    dict(reset=True, **step("await_condition", written(5, 99, 99), role="await_condition",
                             no_reply=True)),
    dict(reset=True, **step("await_condition", req_vote(6, "n2", (3, 5)), role="follower",
                             flags_set=["REPROCESSED"])),
    dict(reset=True, **step("await_condition", dict(kind="await_timeout"), role="follower",
                             reply=dict(to="n1", success=False, next_index=4),
                             flags_set=["LEADER_MSG"])),
], tweak=AWAIT_TWEAK,
    note="state = output of T4 (missing entry at 4); each step restarts from it")

# -------------------------------------------------------- A.5 leader: AER replies ----
L1_PEERS = dict(n1=dict(next_index=5, match_index=4),
                n2=dict(next_index=1, match_index=0, commit_index_sent=3),
                n3=dict(next_index=2, match_index=1))
vec("L1-L2", "test/ra_server_SUITE.erl:1370-1405 append_entries_reply_success", 3, "n1", "base", [
    step("leader", reply("n2", 5, True, 4, 3, 5), role="leader",
         peers=dict(n2=dict(next_index=4, match_index=3)),
         state=dict(commit_index=3, last_applied=3),
         flags_set=["PIPELINE", "AUX_EVAL", "APPLIED"], no_reply=True, rpcs=[], rpcs_exact=True),
    step("leader", dict(kind="pipeline_rpcs"), role="leader",
         peers=dict(n2=dict(next_index=4, match_index=3)),
         state=dict(commit_index=3, last_applied=3),
         rpcs=[dict(peer="n3", term=5, prev=[1, 1], commit=3, entries=[2, 3])], rpcs_exact=True),
], tweak=dict(commit_index=1, last_applied=1, peers=L1_PEERS))

vec("L3", "test/ra_server_SUITE.erl:1407-1416 append_entries_reply_success (term 7)", 3, "n1", "base", [
    step("leader", reply("n2", 7, True, 4, 3, 5), role="leader",
         peers=dict(n2=dict(next_index=4, match_index=3)),
         state=dict(commit_index=1, last_applied=1, current_term=7)),
], tweak=dict(commit_index=1, last_applied=1, peers=L1_PEERS, current_term=7),
    note="Raft 5.4.2: entry 3 has term 5 != current term 7 -> no commit")

vec("L4", "test/ra_server_SUITE.erl:1419-1447 append_entries_reply_no_success", 3, "n1", "base", [
    step("leader", reply("n2", 5, False, 2, 1, 1), role="leader",
         peers=dict(n2=dict(next_index=4, match_index=1)),
         state=dict(commit_index=1, last_applied=1),
         rpcs=[dict(peer="n3", term=5, prev=[1, 1], commit=1, entries=[2, 3]),
               dict(peer="n2")], rpcs_exact=True),
], tweak=dict(commit_index=1, last_applied=1,
              peers=dict(n1=dict(next_index=1, match_index=0),
                         n2=dict(next_index=3, match_index=0),
                         n3=dict(next_index=2, match_index=1, commit_index_sent=1))))

vec("L5", "test/ra_server_SUITE.erl:1449-1462 append_entries_reply_no_success_from_unknown_peer",
    3, "n1", "base", [
        step("leader", reply("n2", 5, False, 2, 1, 1), role="leader", state_unchanged=True,
             no_reply=True, rpcs=[], rpcs_exact=True),
    ], tweak=dict(commit_index=1, last_applied=1, members_present=["n1"],
                  peers=dict(n1=dict(next_index=1, match_index=0))))

vec("L6", "test/ra_server_SUITE.erl:3196-3249 leader_received_append_entries_reply_with_stale_last_index",
    3, "n1", "empty", [
        step("leader", reply("n2", 2, False, 3, 2, 1), role="leader",
             peers=dict(n2=dict(next_index=4)),
             rpcs=[dict(peer="n2", entries=[2, 3])], rpcs_exact=True),
    ], tweak=dict(current_term=2, commit_index=3, last_applied=4,
                  log=[[0, 0], [1, 1], [2, 2], [3, 5]], last_written=[3, 5],
                  peers=dict(n1=dict(next_index=1, match_index=0),
                             n2=dict(next_index=3, match_index=0),
                             n3=dict(next_index=4, match_index=3, commit_index_sent=3))),
    log_model="real")

vec("L7", "test/ra_server_SUITE.erl:1715-1732 leader_does_not_abdicate_to_unknown_peer", 3, "n1", "base", [
    dict(reset=True, **step("leader", reply("n4", 6, False, 4, 3, 5), role="leader",
                             state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("leader", req_vote(6, "n4", (3, 5)), role="leader",
                             state_unchanged=True, no_reply=True)),
])

vec("S2", "test/ra_server_SUITE.erl:2665-2688 leader_pre_vote_sends_snapshot_to_backoff_peer", 3, "n1", "base", [
    # a leader answering a pre_vote_rpc of its own term enforces its leadership with make_all_rpcs/1, which
    # also contacts the peer in {snapshot_backoff, 2} and cancels its retry timer
    dict(reset=True, **step("leader", pre_vote(5, "n1", (3, 5)), role="leader", no_reply=True,
                             cancel_backoff=["n2"], flags_set=["CANCEL_SNAPSHOT_RETRY"],
                             rpcs=[dict(peer="n2", prev=[3, 5], commit=3),
                                   dict(peer="n3", prev=[3, 5], commit=3)], rpcs_exact=True)),
    # the tick (make_rpcs/1 over stale_peers/1) only takes normal peers: n2 stays out
    dict(reset=True, **step("leader", dict(kind="tick"), role="leader", cancel_backoff=[],
                             rpcs=[dict(peer="n3", prev=[3, 5], commit=3)], rpcs_exact=True)),
], tweak=dict(peers_backoff=["n2"], votes=1))

vec("S1", "test/ra_server_SUITE.erl:2438-2467 follower_state_resets_peer_status", 3, "n1", "base", [
    # the reference calls handle_state_enter(follower, leader, State) on a leader whose peers are
    # {sending_snapshot, _, _} and disconnected; here the state-enter half is part of whatever transition
    # makes the leader a follower (a reply from a higher term): every peer status is normal afterwards
    dict(reset=True, **step("leader", reply("n2", 6, False, 4, 3, 5), role="follower",
                             state=dict(current_term=6, leader_id=None, status_mask=255), no_reply=True)),
    dict(reset=True, **step("leader", aer(6, "n3", (3, 5), 3, []), role="follower",
                             state=dict(current_term=6, status_mask=255))),
], tweak=dict(peers_not_normal=["n2", "n3"]))

vec("L8", "test/ra_server_SUITE.erl:1752-1794 higher_term_detected", 3, "n1", "base", [
    dict(reset=True, **step("leader", reply("n2", 6, False, 4, 3, 5), role="follower",
                             state=dict(current_term=6, leader_id=None), no_reply=True,
                             rpcs=[], rpcs_exact=True)),
    dict(reset=True, **step("follower", reply("n2", 6, False, 4, 3, 5), role="follower",
                             state=dict(current_term=6), no_reply=True)),
    dict(reset=True, **step("candidate", reply("n2", 6, False, 4, 3, 5), role="follower",
                             state=dict(current_term=6), no_reply=True)),
    # {follower, #{current_term := 6}, [{next_event, AERpc}]}: the engine runs the next_event
    # itself (REPROCESSED); role and term are what the reference asserts
    dict(reset=True, **step("leader", aer(6, "n3", (3, 5), 3, []), role="follower",
                             state=dict(current_term=6), flags_set=["REPROCESSED"])),
    dict(reset=True, **step("candidate", aer(6, "n3", (3, 5), 3, []), role="follower",
                             state=dict(current_term=6), flags_set=["REPROCESSED"])),
])

vec("L9", "test/ra_server_SUITE.erl:1735-1750 leader_replies_to_append_entries_rpc_with_lower_term",
    3, "n1", "base", [
        step("leader", aer(4, "n3", (3, 5), 3, []), role="leader",
             reply=dict(to="n3", term=5, success=False), effects_only_reply=True,
             state_unchanged=True),
    ])

vec("L10", "test/ra_server_SUITE.erl:1203-1290 append_entries_reply_success_promotes_nonvoter",
    3, "n1", "base", [
        step("leader", reply("n2", 5, True, 4, 3, 5), role="leader",
             peers=dict(n2=dict(next_index=4, match_index=3)),
             state=dict(commit_index=1), flags_set=["PIPELINE"]),
    ], tweak=dict(commit_index=1, last_applied=1, peers=L1_PEERS, nonvoters=["n2"]),
    note="non-voter n2 is excluded from match_indexes: [LW=3, n3=1] -> 1")

vec("L11-L12", "test/ra_server_SUITE.erl:2404-2421, 2469-2500 command / written / reply", 3, "n1", "base", [
    step("leader", dict(kind="append", n=1), role="leader",
         rpcs=[dict(peer="n3", term=5, prev=[3, 5], commit=3, entries=[4, 4]),
               dict(peer="n2", term=5, prev=[3, 5], commit=3, entries=[4, 4])], rpcs_exact=True),
    step("leader", written(5, 4, 4), role="leader", flags_set=["PIPELINE"]),
    step("leader", reply("n2", 5, True, 5, 4, 5), role="leader",
         state=dict(commit_index=4, last_applied=4), flags_set=["AUX_EVAL", "APPLIED", "PIPELINE"]),
])

# ------------------------------------------------------------------- A.6 votes ----
vec("V1-V3", "test/ra_server_SUITE.erl:1464-1482 follower_request_vote", 3, "n1", "base", [
    step("follower", req_vote(6, "n2", (3, 5)), role="follower",
         state=dict(voted_for="n2", current_term=6),
         reply=dict(vote=True, term=6, success=True), flags_set=["PERSIST"]),
    step("follower", req_vote(6, "n2", (3, 5)), role="follower",
         state=dict(voted_for="n2", current_term=6),
         reply=dict(vote=True, term=6, success=True), flags_clear=["PERSIST"]),
    step("follower", req_vote(6, "n3", (3, 5)), role="follower",
         state=dict(voted_for="n2", current_term=6),
         reply=dict(vote=True, term=6, success=False)),
])

vec("V4-V7", "test/ra_server_SUITE.erl:1484-1510 follower_request_vote", 3, "n1", "base", [
    dict(reset=True, **step("follower", req_vote(4, "n2", (3, 5)), role="follower",
                             state=dict(current_term=5),
                             reply=dict(vote=True, term=5, success=False))),
    dict(reset=True, **step("follower", req_vote(6, "n2", (3, 4)), role="follower",
                             state=dict(current_term=6),
                             reply=dict(vote=True, term=6, success=False))),
    dict(reset=True, **step("follower", req_vote(6, "n2", (4, 5)), role="follower",
                             state=dict(current_term=6, voted_for="n2"),
                             reply=dict(vote=True, term=6, success=True))),
])

vec("V7", "test/ra_server_SUITE.erl:1508-1510 follower_request_vote (non-voter)", 3, "n1", "base", [
    step("follower", req_vote(6, "n2", (3, 5)), role="follower", state_unchanged=True,
         no_reply=True),
], tweak=dict(self_nonvoter=True))

vec("V8", "test/ra_server_SUITE.erl:1700-1713 request_vote_rpc_with_lower_term", 3, "n1", "base", [
    dict(reset=True, **step("candidate", req_vote(5, "n2", (3, 5)), role="candidate",
                             reply=dict(vote=True, term=6, success=False), state_unchanged=True)),
    dict(reset=True, **step("leader", req_vote(5, "n2", (3, 5)), role="leader",
                             reply=dict(vote=True, term=6, success=False), state_unchanged=True)),
], tweak=dict(current_term=6, voted_for="n1"))

vec("V9", "test/ra_server_SUITE.erl:1779-1787 higher_term_detected (request_vote)", 3, "n1", "base", [
    dict(reset=True, **step("leader", req_vote(6, "n2", (3, 5)), role="follower",
                             state=dict(current_term=6), flags_set=["REPROCESSED"])),
    dict(reset=True, **step("candidate", req_vote(6, "n2", (3, 5)), role="follower",
                             state=dict(current_term=6), flags_set=["REPROCESSED"])),
])

vec("V10", "test/ra_server_SUITE.erl:2503-2548 candidate_election", 5, "n1", "base", [
    step("candidate", vote_result(6, True, "n2"), role="candidate", state=dict(votes=2)),
    step("candidate", vote_result(6, False, "n3"), role="candidate", state=dict(votes=2)),
    dict(fork=True, **step("candidate", vote_result(7, False, "n3"), role="follower",
                            state=dict(current_term=7))),
    step("candidate", vote_result(6, True, "n4"), role="leader",
         peers=dict(n2=dict(next_index=4, match_index=0), n3=dict(next_index=4, match_index=0),
                    n4=dict(next_index=4, match_index=0), n5=dict(next_index=4, match_index=0)),
         flags_set=["BECAME_LEADER"]),
], tweak=dict(current_term=6, votes=1),
    note="5 members: quorum 3; a fork step does not carry its state forward")

# ------------------------------------------------ elections: pre-vote, timeouts ----
PV = dict(vote=False, pre_vote=True)
vec("E1", "test/ra_server_SUITE.erl:1514-1620 follower_pre_vote", 3, "n1", "base", [
    dict(reset=True, **step("follower", pre_vote(5, "n2", (3, 5)), role="follower",
                             state=dict(current_term=5),
                             reply=dict(pre_vote=True, term=5, token=77, success=True),
                             effects_only_reply=True)),
    dict(reset=True, **step("follower", pre_vote(5, "n2", (3, 5), version=2), role="follower",
                             reply=dict(pre_vote=True, term=5, token=77, success=False))),
    dict(reset=True, **step("follower", pre_vote(5, "n2", (3, 5), version=0), role="follower",
                             reply=dict(pre_vote=True, term=5, token=77, success=True))),
    dict(reset=True, **step("follower", pre_vote(5, "n2", (3, 5), machine_version=99), role="follower",
                             reply=dict(pre_vote=True, term=5, token=77, success=False),
                             flags_set=["START_ELECTION_TIMEOUT"])),
    dict(reset=True, **step("follower", pre_vote(4, "n2", (3, 5)), role="follower",
                             state=dict(current_term=5),
                             reply=dict(pre_vote=True, term=5, token=77, success=False),
                             effects_only_reply=True)),
    dict(reset=True, **step("follower", pre_vote(6, "n2", (3, 4)), role="follower",
                             state=dict(current_term=6), no_reply=True,
                             flags_set=["START_ELECTION_TIMEOUT"])),
    dict(reset=True, **step("follower", pre_vote(5, "n2", (4, 5)), role="follower",
                             state=dict(current_term=5),
                             reply=dict(pre_vote=True, term=5, token=77, success=True))),
    dict(reset=True, **step("pre_vote", pre_vote(5, "n2", (3, 5)), role="pre_vote",
                             state=dict(current_term=5),
                             reply=dict(pre_vote=True, term=5, token=77, success=True))),
    dict(reset=True, **step("await_condition", pre_vote(5, "n2", (3, 5)), role="await_condition",
                             state=dict(current_term=5),
                             reply=dict(pre_vote=True, term=5, token=77, success=True))),
], note="also pre_vote_receives_pre_vote :1665-1680, await_condition_receives_pre_vote :1682-1698")

for _i, (_eff, _ours, _theirs, _ok) in enumerate([(1, 0, 1, True), (1, 1, 0, False), (3, 2, 2, False),
                                                   (1, 3, 2, True), (0, 1, 0, True), (2, 2, 2, True)]):
    vec(f"E2.{_i}", "test/ra_server_SUITE.erl:1548-1596 follower_pre_vote (machine versions)", 3, "n1",
        "base", [step("follower", pre_vote(5, "n2", (3, 5), machine_version=_theirs), role="follower",
                      reply=dict(pre_vote=True, term=5, token=77, success=_ok))],
        tweak=dict(effective_machine_version=_eff, machine_version=_ours))

vec("E3", "test/ra_server_SUITE.erl:1618-1620 follower_pre_vote (non-voter)", 3, "n1", "base", [
    step("follower", pre_vote(5, "n2", (3, 5)), role="follower", state_unchanged=True, no_reply=True),
], tweak=dict(self_nonvoter=True))

vec("E4", "test/ra_server_SUITE.erl:1634-1663 pre_vote_does_not_set_voted_for", 3, "n1", "base", [
    step("follower", pre_vote(5, "n2", (3, 5)), role="follower", state=dict(voted_for=None),
         reply=dict(pre_vote=True, term=5, token=77, success=True)),
    step("follower", req_vote(5, "n3", (3, 5)), role="follower", state=dict(voted_for="n3"),
         reply=dict(vote=True, term=5, success=True)),
])

vec("E5", "test/ra_server_SUITE.erl:2550-2577 pre_vote_election", 5, "n1", "base", [
    step("pre_vote", pre_vote_result(5, True, 1234), role="pre_vote", state=dict(votes=2), no_reply=True),
    dict(fork=True, **step("pre_vote", pre_vote_result(5, True, 999), role="pre_vote",
                            state=dict(votes=2), no_reply=True)),
    step("pre_vote", pre_vote_result(5, False, 1234), role="pre_vote", state=dict(votes=2)),
    dict(fork=True, **step("pre_vote", pre_vote_result(6, False, 1234), role="follower",
                            state=dict(current_term=6, votes=0), no_reply=True)),
    step("pre_vote", pre_vote_result(5, True, 1234), role="candidate", state=dict(current_term=6),
         flags_set=["SEND_VOTE_REQUESTS", "PERSIST"], flags_clear=["PRE_VOTE_REQS"]),
], tweak=dict(votes=1, pre_vote_token=1234, role="pre_vote"))

vec("E6", "test/ra_server_SUITE.erl:2579-2587 pre_vote_election_non_voter", 5, "n1", "base", [
    step("pre_vote", pre_vote_result(5, True, 1234), role="pre_vote", state=dict(votes=1), no_reply=True),
], tweak=dict(votes=1, pre_vote_token=1234, role="pre_vote", self_nonvoter=True))

vec("E7", "test/ra_server_SUITE.erl:2589-2606 pre_vote_election_reverts", 5, "n1", "base", [
    dict(reset=True, **step("pre_vote", req_vote(6, "n2", (3, 5)), role="follower",
                             state=dict(current_term=6), flags_set=["REPROCESSED"])),
    dict(reset=True, **step("pre_vote", aer(5, "n2", (3, 5), 3, []), role="follower",
                             state=dict(current_term=5), flags_set=["REPROCESSED"])),
], tweak=dict(votes=1, pre_vote_token=1234, role="pre_vote"))

vec("E8", "test/ra_server_SUITE.erl:335-381 election_timeout", 3, "n1", "base", [
    dict(reset=True, **step("follower", dict(kind="election_timeout", token=4242), role="pre_vote",
                             state=dict(current_term=5, votes=1, pre_vote_token=4242),
                             vote_requests=dict(pre_vote=True, term=5, token=4242, last=[3, 5]))),
    dict(reset=True, **step("pre_vote", dict(kind="election_timeout", token=4343), role="pre_vote",
                             state=dict(current_term=5, votes=1, pre_vote_token=4343),
                             vote_requests=dict(pre_vote=True, term=5, token=4343, last=[3, 5]))),
    dict(reset=True, **step("candidate", dict(kind="election_timeout", token=1), role="candidate",
                             state=dict(current_term=6, votes=1, voted_for="n1"),
                             vote_requests=dict(pre_vote=False, term=6, last=[3, 5]))),
], note="the reference returns votes=0 plus a {next_event, cast, VoteForSelf}; the engine applies "
        "that self vote in the same decision, hence votes=1")

vec("E9", "test/ra_server_SUITE.erl:350-353 election_timeout (non-voters ignore it)", 3, "n1", "base", [
    dict(reset=True, **step("follower", dict(kind="election_timeout", token=5), role="follower",
                             state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", dict(kind="election_timeout", token=5),
                             role="await_condition", state_unchanged=True, no_reply=True)),
], tweak=dict(self_nonvoter=True))

# -------------------------------------------- A.6b role-specific clauses the suites also pin ----
vec("C1", "test/ra_server_SUITE.erl:1188-1201 candidate_handles_append_entries_rpc", 3, "n1", "base", [
    step("candidate", aer(4, "n1", (3, 5), 3, []), role="candidate",
         reply=dict(to="n1", term=5, success=False, last_index=3, last_term=5), effects_only_reply=True),
], tweak=dict(commit_index=1))

vec("E10", "test/ra_server_SUITE.erl:1666-1680 pre_vote_receives_pre_vote", 3, "n1", "base", [
    step("pre_vote", pre_vote(5, "n2", (3, 5), token=4242), role="pre_vote", state=dict(current_term=5),
         reply=dict(pre_vote=True, term=5, token=4242, success=True)),
])

vec("E11", "test/ra_server_SUITE.erl:1682-1698 await_condition_receives_pre_vote", 3, "n1", "base", [
    step("await_condition", pre_vote(5, "n2", (3, 5), token=4242), role="await_condition",
         state=dict(current_term=5), reply=dict(pre_vote=True, term=5, token=4242, success=True)),
], tweak=dict(role="await_condition", cond_reason="missing"))

vec("E12", "test/ra_server_SUITE.erl:2620-2642 candidate_receives_pre_vote", 5, "n1", "base", [
    dict(reset=True, **step("candidate", pre_vote(5, "n1", (3, 5), token=4242), role="candidate",
                             reply=dict(pre_vote=True, token=4242, success=True))),
    dict(reset=True, **step("candidate", pre_vote(5, "n1", (2, 5), token=4242), role="candidate",
                             reply=dict(pre_vote=True, token=4242, success=False))),
    dict(reset=True, **step("candidate", pre_vote(6, "n1", (3, 5), token=4242), role="follower",
                             state=dict(current_term=6))),
], tweak=dict(votes=1))

vec("E13", "test/ra_server_SUITE.erl:2644-2662 leader_receives_pre_vote", 5, "n1", "base", [
    dict(reset=True, **step("leader", pre_vote(5, "n1", (3, 5), token=4242), role="leader", no_reply=True,
                             rpcs=[dict(peer="n2"), dict(peer="n3"), dict(peer="n4"), dict(peer="n5")],
                             rpcs_exact=True)),
    dict(reset=True, **step("leader", pre_vote(6, "n1", (3, 5), token=4242), role="follower",
                             state=dict(current_term=6))),
], tweak=dict(votes=1), note="a leader answers a same-term pre-vote with rpcs to every peer (make_all_rpcs)")

vec("PLA", "test/ra_server_SUITE.erl:3888-3909 persist_last_applied_with_unwritten", 3, "n1", "empty", [
    step("follower", aer(1, "n1", (0, 0), 1, [(1, 1)]), role="follower",
         state=dict(leader_id="n1", current_term=1, commit_index=1, last_applied=1, last_written=[0, 0])),
    step("follower", written(1, 1, 1), role="follower", state=dict(last_applied=1, last_written=[1, 1])),
], note="persist_last_applied/1 (src/ra_server.erl:2539-2559) stays on the host: it stores "
        "min(last_applied, last_written_index) when that exceeds the persisted value -- 0 after the "
        "first step, 1 after the second, from exactly the two cursors this vector pins")

# -------------------------------------------- A.7 real-log last_written cursor ----
vec("R1", "test/ra_log_2_SUITE.erl:189-211 (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1)]), role="follower"),
    step("follower", written(1, 1, 2), role="follower",
         state=dict(last_written=[2, 1])),
    step("follower", aer(2, "n1", (0, 0), 0, [(1, 2)]), role="follower",
         state=dict(last_written=[0, 0], last_index=1, last_term=2)),
], log_model="real", note="overwrite lowers last_written immediately: min(first-1, LW)")

vec("R2", "test/ra_log_2_SUITE.erl:213-265 (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(2, "n1", (10, 1), 10, [(11, 2), (12, 2), (13, 2), (14, 2), (15, 2)]),
         role="follower"),
    step("follower", written(2, 11, 15), role="follower", state=dict(last_written=[15, 2])),
    step("follower", aer(3, "n1", (12, 2), 10, [(13, 3)]), role="follower",
         state=dict(last_written=[12, 2], last_index=13, last_term=3)),
], tweak=dict(current_term=2, install_snapshot=[10, 1], last_applied=10, commit_index=10),
    log_model="real")

vec("R3", "test/ra_log_2_SUITE.erl:267-293 (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1)]), role="follower"),
    step("follower", aer(2, "n1", (0, 0), 0, [(1, 2)]), role="follower",
         state=dict(last_index=1, last_term=2)),
    step("follower", aer(2, "n1", (1, 2), 0, [(2, 2)]), role="follower"),
    step("follower", written(1, 1, 2), role="follower", state=dict(last_written=[0, 0]),
         no_reply=True),
    step("follower", written(2, 1, 2), role="follower", state=dict(last_written=[2, 2])),
], log_model="real")

vec("R4", "test/ra_log_2_SUITE.erl:295-327 (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1)]), role="follower"),
    step("follower", aer(3, "n1", (1, 1), 0, []), role="follower",
         state=dict(last_written=[0, 0], last_index=1, last_term=1), flags_set=["TRUNCATED"]),
    step("follower", aer(3, "n1", (1, 1), 0, [(2, 3), (3, 3)]), role="follower"),
    step("follower", written(1, 1, 2), role="follower", state=dict(last_written=[1, 1])),
    step("follower", written(3, 2, 3), role="follower", state=dict(last_written=[3, 3])),
], log_model="real")

vec("R6", "test/ra_log_2_SUITE.erl:928-940 last_written_overwrite (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower"),
    step("follower", written(1, 1, 4), role="follower", state=dict(last_written=[4, 1], pending_first=5)),
    step("follower", aer(2, "n1", (2, 1), 0, [(3, 2)]), role="follower",
         state=dict(last_written=[2, 1], last_index=3, last_term=2, pending_first=3)),
    step("follower", written(2, 3, 3), role="follower", state=dict(last_written=[3, 2], pending_first=4)),
], log_model="real")

vec("R7", "test/ra_log_2_SUITE.erl:942-968 last_written_overwrite_2 (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower",
         state=dict(last_written=[0, 0], pending_first=1)),
    step("follower", aer(2, "n1", (3, 1), 0, [(4, 2), (5, 2)]), role="follower",
         state=dict(last_written=[0, 0], last_index=5, last_term=2, pending_first=1)),
    # the first batch's written event arrives after the overwrite: applied up to the last index
    # whose term still matches
    step("follower", written(1, 1, 4), role="follower", state=dict(last_written=[3, 1], pending_first=4)),
    step("follower", written(2, 4, 5), role="follower", state=dict(last_written=[5, 2], pending_first=6)),
], log_model="real")

vec("R8", "test/ra_log_2_SUITE.erl:970-1000 last_index_reset (driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1)]), role="follower"),
    step("follower", written(1, 1, 5), role="follower", state=dict(last_written=[5, 1])),
    step("follower", aer(2, "n1", (3, 1), 0, []), role="follower", flags_set=["TRUNCATED"],
         state=dict(last_written=[3, 1], last_index=3, last_term=1)),
    step("follower", aer(2, "n1", (3, 1), 0, [(4, 2)]), role="follower",
         state=dict(last_index=4, last_term=2, first_index=0, log=[[0, 0], [1, 1], [2, 1], [3, 1], [4, 2]])),
], log_model="real")

vec("R9", "test/ra_log_2_SUITE.erl:1208-1230 last_index_reset_before_written (driven through follower AERs)",
    3, "n2", "empty", [
        step("follower", aer(1, "n1", (0, 0), 0, [(1, 1), (2, 1), (3, 1), (4, 1)]), role="follower",
             state=dict(last_written=[0, 0], last_index=4)),
        step("follower", aer(2, "n1", (3, 1), 0, []), role="follower", flags_set=["TRUNCATED"],
             state=dict(last_written=[0, 0], last_index=3, last_term=1)),
        # the written event of 1..4 must not move last_written beyond the reset
        step("follower", written(1, 1, 4), role="follower", state=dict(last_written=[3, 1], last_index=3)),
    ], log_model="real")

vec("R5", "test/ra_log_2_SUITE.erl:157-186 snapshot_before_written (driven as follower log events)",
    3, "n2", "empty", [
        step("follower", dict(kind="snapshot_written", index=10, term=1), role="follower",
             state=dict(last_written=[10, 1], snapshot=[10, 1], first_index=11, last_index=19)),
        step("follower", written(1, 6, 19), role="follower", state=dict(last_written=[19, 1])),
    ], tweak=dict(current_term=1, leader_id="n1", log=[[0, 0]] + [[i, 1] for i in range(1, 20)],
                  last_written=[5, 1], commit_index=10, last_applied=10),
    log_model="real", note="snapshot_written overtakes the written events of lower entries")

# -------------------------------------------- A.8 real-log `pending` / written-event gaps ----
# The reference pins the END state of these WAL scenarios (ra_log:last_written/1); the
# intermediate expectations (resend request, cursors untouched) follow
# src/ra_log.erl:897-944 + ra_seq:remove_prefix/2 (src/ra_seq.erl:144-147, 278-291).
vec("P1", "test/ra_log_2_SUITE.erl:1556-1573 missed_written_then_write (driven as leader appends + log events)",
    3, "n1", "empty", [
        step("leader", dict(kind="append", n=9), role="leader",
             state=dict(last_index=9, last_term=2, last_written=[0, 0], pending_first=1)),
        step("leader", dict(kind="append", n=5), role="leader",
             state=dict(last_index=14, last_term=2, last_written=[0, 0], pending_first=1)),
        # the written event of 1..9 was lost; the one for 10..14 is not a prefix of pending
        step("leader", written(2, 10, 14), role="leader",
             state=dict(last_written=[0, 0], pending_first=1), flags_set=["RESEND_PENDING"]),
        # after the resend the WAL confirms everything
        step("leader", written(2, 1, 14), role="leader",
             state=dict(last_written=[14, 2], pending_first=15), flags_clear=["RESEND_PENDING"]),
    ], tweak=dict(current_term=2, role="leader", leader_id="n1"),
    log_model="real", note="{14,2} == last_written is the reference's assertion")

vec("P2", "test/ra_log_2_SUITE.erl:1031-1085 set_last_index_limits_pending_before_new_write "
          "(driven through follower AERs)", 3, "n2", "empty", [
    step("follower", aer(1, "n1", (0, 0), 0, [(i, 1) for i in range(1, 11)]), role="follower",
         state=dict(last_index=10, pending_first=1)),
    # set_last_index(5) while 6..10 are still pending: pending is limited to the new last index
    step("follower", aer(1, "n1", (5, 1), 0, []), role="follower",
         state=dict(last_index=5, last_term=1, pending_first=1), flags_set=["TRUNCATED"]),
    step("follower", written(1, 1, 10), role="follower",
         state=dict(last_written=[5, 1], pending_first=6)),
], log_model="real", note="{5,1} == last_written and nothing left to resend are the reference's assertions")

vec("P3", "test/ra_log_2_SUITE.erl:710-760 written_event_after_snapshot (driven as leader appends + log events)",
    3, "n1", "empty", [
        step("leader", dict(kind="append", n=2), role="leader",
             state=dict(last_index=2, pending_first=1)),
        step("leader", dict(kind="snapshot_written", index=2, term=1), role="leader",
             state=dict(last_written=[2, 1], snapshot=[2, 1], pending_first=3)),
        # the written event for [1,2] arrives after the snapshot: nothing to do
        step("leader", written(1, 1, 2), role="leader",
             state=dict(last_written=[2, 1], pending_first=3), flags_clear=["RESEND_PENDING"]),
        step("leader", dict(kind="append", n=2), role="leader",
             state=dict(last_index=4, pending_first=3)),
        step("leader", written(1, 3, 4), role="leader",
             state=dict(last_written=[4, 1], pending_first=5)),
    ], tweak=dict(current_term=1, role="leader", leader_id="n1", commit_index=2, last_applied=2),
    log_model="real")

# -------------------------------------------- A.9 heartbeats / consistent-query quorum ----
# base_state/2: self n1, term 5, leader n1, query_index 0, every peer query_index 0.
def hb(term, leader, qi):
    return dict(kind="heartbeat_rpc", term=term, **{"from": leader}, query_index=qi)


def hb_reply(peer, term, qi):
    d = dict(kind="heartbeat_reply", term=term, query_index=qi)
    if peer is not None:
        d["from"] = peer
    return d


vec("H1", "test/ra_server_SUITE.erl:3349-3388 follower_heartbeat", 3, "n1", "base", [
    dict(reset=True, **step("follower", hb(4, "n1", 1), role="follower", state_unchanged=True,
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1), effects_only_reply=True)),
    dict(reset=True, **step("follower", hb(5, "n1", 1), role="follower", state_unchanged=True,
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1), effects_only_reply=True)),
    dict(reset=True, **step("follower", hb(6, "n1", 1), role="follower",
                             state=dict(current_term=6, voted_for=None),
                             reply=dict(heartbeat=True, to="n1", term=6, query_index=1), effects_only_reply=True)),
])

vec("H2", "test/ra_server_SUITE.erl:3390-3406 follower_heartbeat_reply", 3, "n1", "base", [
    dict(reset=True, **step("follower", hb_reply("n1", 5, 2), role="follower", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("follower", hb_reply("n1", 4, 2), role="follower", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("follower", hb_reply("n1", 6, 2), role="follower",
                             state=dict(current_term=6, voted_for=None), no_reply=True)),
])

vec("H3", "test/ra_server_SUITE.erl:3408-3440 candidate_heartbeat", 3, "n1", "base", [
    dict(reset=True, **step("candidate", hb(5, "n1", 1), role="follower", state=dict(current_term=5),
                             flags_set=["REPROCESSED"],
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1))),
    dict(reset=True, **step("candidate", hb(6, "n1", 1), role="follower",
                             state=dict(current_term=6, voted_for=None), flags_set=["REPROCESSED"],
                             reply=dict(heartbeat=True, to="n1", term=6, query_index=1))),
    dict(reset=True, **step("candidate", hb(4, "n1", 1), role="candidate", state_unchanged=True,
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1), effects_only_reply=True)),
], note="the reference returns {next_event, Heartbeat}; the engine re-processes it as follower in the same "
        "decision (the follower clause replies)")

vec("H4", "test/ra_server_SUITE.erl:3442-3459 candidate_heartbeat_reply", 3, "n1", "base", [
    dict(reset=True, **step("candidate", hb_reply(None, 5, 2), role="candidate", state_unchanged=True,
                             no_reply=True, flags_set=["UNHANDLED"])),
    dict(reset=True, **step("candidate", hb_reply(None, 4, 2), role="candidate", state_unchanged=True,
                             no_reply=True, flags_set=["UNHANDLED"])),
    dict(reset=True, **step("candidate", hb_reply(None, 6, 2), role="follower",
                             state=dict(current_term=6, voted_for=None), no_reply=True)),
])

vec("H5", "test/ra_server_SUITE.erl:3488-3520 pre_vote_heartbeat", 3, "n1", "base", [
    dict(reset=True, **step("pre_vote", hb(5, "n1", 1), role="follower", state=dict(votes=0, current_term=5),
                             flags_set=["REPROCESSED"],
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1))),
    dict(reset=True, **step("pre_vote", hb(6, "n1", 1), role="follower",
                             state=dict(votes=0, current_term=6, voted_for=None), flags_set=["REPROCESSED"],
                             reply=dict(heartbeat=True, to="n1", term=6, query_index=1))),
    dict(reset=True, **step("pre_vote", hb(4, "n1", 1), role="pre_vote", state_unchanged=True,
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1), effects_only_reply=True)),
], tweak=dict(votes=1))

vec("H6", "test/ra_server_SUITE.erl:3522-3546 pre_vote_heartbeat_reply", 3, "n1", "base", [
    dict(reset=True, **step("pre_vote", hb_reply(None, 5, 2), role="pre_vote", state_unchanged=True,
                             no_reply=True, flags_set=["UNHANDLED"])),
    dict(reset=True, **step("pre_vote", hb_reply(None, 4, 2), role="pre_vote", state_unchanged=True,
                             no_reply=True, flags_set=["UNHANDLED"])),
    dict(reset=True, **step("pre_vote", hb_reply(None, 6, 2), role="follower",
                             state=dict(votes=0, current_term=6, voted_for=None), no_reply=True)),
])

vec("H7", "test/ra_server_SUITE.erl:3548-3586 leader_heartbeat", 3, "n1", "base", [
    dict(reset=True, **step("leader", hb(5, "n1", 1), invariant=11)),
    dict(reset=True, **step("leader", hb(6, "n1", 1), role="follower",
                             state=dict(current_term=6, voted_for=None), flags_set=["REPROCESSED"],
                             reply=dict(heartbeat=True, to="n1", term=6, query_index=1))),
    dict(reset=True, **step("leader", hb(4, "n1", 1), role="leader", state_unchanged=True,
                             reply=dict(heartbeat=True, to="n1", term=5, query_index=1), effects_only_reply=True)),
], note="same term: exit(leader_saw_heartbeat_rpc_in_same_term); higher term: the reference clears "
        "leader_id and returns {next_event, Msg}, re-processed here as follower")

vec("H8", "test/ra_server_SUITE.erl:3614-3694 leader_heartbeat_reply_same_term", 3, "n1", "base", [
    dict(reset=True, **step("leader", hb_reply("n2", 5, 2), role="leader", no_reply=True,
                             peers=dict(n2=dict(peer_query_index=2)), query_quorum=2)),
    dict(reset=True, **step("leader", hb_reply(None, 5, 2), role="leader", no_reply=True, state_unchanged=True,
                             query_quorum=0)),
    dict(reset=True, **step("leader", hb_reply("n2", 5, 1), role="leader", no_reply=True,
                             peers=dict(n2=dict(peer_query_index=1)), query_quorum=1)),
    dict(reset=True, **step("leader", hb_reply("n2", 5, 3), role="leader", no_reply=True,
                             peers=dict(n2=dict(peer_query_index=3)), query_quorum=3)),
], tweak=dict(query_index=3),
    note="query_index=3 on the leader; a single reply is a consensus in a 3-node cluster: queued queries "
         "with an index <= the consensus index are released by the host (queries_waiting_heartbeats)")

vec("H9", "test/ra_server_SUITE.erl:3588-3612 leader_heartbeat_reply_node_size_5", 5, "n1", "base", [
    step("leader", hb_reply("n2", 5, 2), role="leader", no_reply=True,
         peers=dict(n2=dict(peer_query_index=2)), query_quorum=0),
    step("leader", hb_reply("n3", 5, 2), role="leader", no_reply=True,
         peers=dict(n3=dict(peer_query_index=2)), query_quorum=2),
], tweak=dict(query_index=2), note="two of four peers are needed before query 2 is released")

vec("H10", "test/ra_server_SUITE.erl:3922-3963 leader_heartbeat_reply_lower_term / _higher_term", 3, "n1", "base", [
    dict(reset=True, **step("leader", hb_reply("n2", 4, 0), role="leader", state_unchanged=True, no_reply=True,
                             flags_clear=["QUERY_QUORUM"])),
    dict(reset=True, **step("leader", hb_reply("n2", 4, 1), role="leader", state_unchanged=True, no_reply=True,
                             flags_clear=["QUERY_QUORUM"])),
    dict(reset=True, **step("leader", hb_reply("n2", 6, 0), role="follower", no_reply=True,
                             state=dict(current_term=6, voted_for=None, leader_id=None),
                             flags_clear=["REPROCESSED"])),
    dict(reset=True, **step("leader", hb_reply("n2", 6, 1), role="follower", no_reply=True,
                             state=dict(current_term=6, voted_for=None, leader_id=None))),
])

vec("H11", "test/ra_server_SUITE.erl:3752-3790 leader_consistent_query", 3, "n1", "base", [
    step("leader", dict(kind="consistent_query"), role="leader", state=dict(query_index=1), no_reply=True,
         heartbeats=dict(to=["n2", "n3"], term=5, query_index=1)),
    step("leader", dict(kind="consistent_query"), role="leader", state=dict(query_index=2), no_reply=True,
         heartbeats=dict(to=["n2", "n3"], term=5, query_index=2)),
], note="cluster_change_permitted = true; the QueryRef is queued by the host under the returned index")

vec("H12", "test/ra_server_SUITE.erl:3800-3840 await_condition_heartbeat_dropped / _reply_dropped", 3, "n1", "base", [
    dict(reset=True, **step("await_condition", hb(5, "n1", 0), role="await_condition", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", hb(6, "n1", 0), role="await_condition", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", hb(4, "n1", 0), role="await_condition", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", hb_reply("n2", 5, 0), role="await_condition", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", hb_reply("n2", 6, 0), role="await_condition", state_unchanged=True, no_reply=True)),
    dict(reset=True, **step("await_condition", hb_reply("n2", 4, 0), role="await_condition", state_unchanged=True, no_reply=True)),
], tweak=dict(role="await_condition", cond_reason="missing"))

WAL_AER = aer(5, "n1", (3, 5), 3, [(4, 5)])
vec("W1", "test/ra_server_SUITE.erl:1004-1033 wal_down_condition_follower", 3, "n1", "base", [
    dict(reset=True, **step("await_condition", WAL_AER, role="await_condition", state_unchanged=True, no_reply=True,
                             flags_clear=["REPROCESSED", "WROTE", "LEADER_MSG"])),
    dict(reset=True, **step("await_condition", dict(WAL_AER, can_write=True), role="follower",
                             state=dict(last_index=4, last_term=5, commit_index=3, last_applied=3),
                             flags_set=["REPROCESSED", "WROTE", "LEADER_MSG"], no_reply=True)),
    dict(reset=True, **step("await_condition", dict(kind="await_timeout"), role="follower", no_reply=True,
                             flags_clear=["LEADER_MSG"])),
    dict(reset=True, **step("await_condition", dict(hb(5, "n1", 0), can_write=True), role="follower",
                             flags_set=["REPROCESSED"], reply=dict(heartbeat=True, to="n1", term=5))),
], tweak=dict(commit_index=3, role="await_condition", cond_reason="wal_down"),
    note="the suite mocks ra_log:write/2 -> {error, wal_down}: that step is the HOST's (it re-uploads the server in "
         "await_condition / RGB_COND_WAL_DOWN with the log as before the write and commit_index = LeaderCommit); steps 0-1 "
         "are the suite's two handle_await_condition calls (can_write false, then true); the timeout and the "
         "heartbeat steps follow src/ra_server.erl:1932-1959 with a condition map that has no timeout effects")

vec("W2", "test/ra_server_SUITE.erl:1035-1073 wal_down_condition_leader", 3, "n1", "base", [
    dict(reset=True, **step("await_condition", dict(kind="await_timeout"), role="leader", no_reply=True,
                             state=dict(commit_index=1, current_term=5),
                             flags_set=["TRANSFER_LEADERSHIP", "ROLE_CHANGED"], flags_clear=["REPROCESSED"])),
    dict(reset=True, **step("await_condition", dict(kind="await_timeout", can_write=True), role="leader", no_reply=True,
                             state=dict(commit_index=1, current_term=5),
                             flags_set=["ROLE_CHANGED"], flags_clear=["TRANSFER_LEADERSHIP", "REPROCESSED"])),
    dict(reset=True, **step("await_condition", reply("n2", 5, True, 4, 3, 5), role="await_condition",
                             state_unchanged=True, no_reply=True, flags_clear=["REPROCESSED", "PIPELINE"])),
    dict(reset=True, **step("await_condition", dict(reply("n2", 5, True, 4, 3, 5), can_write=True), role="leader",
                             state=dict(commit_index=3, last_applied=3),
                             flags_set=["REPROCESSED", "PIPELINE", "AUX_EVAL", "APPLIED"], no_reply=True)),
], tweak=dict(commit_index=1, last_applied=1, role="await_condition", cond_reason="wal_down_leader", peers=L1_PEERS),
    note="the suite mocks ra_log:append/2 -> error(wal_down): that step is the HOST's (it re-uploads the leader as it "
         "was before the {command, _}, in await_condition / RGB_COND_WAL_DOWN_LEADER); steps 0-1 are the suite's two "
         "handle_await_condition(await_condition_timeout, _) calls (can_write false: back to leader with the "
         "transfer_leadership effect; true: back to leader, no effects); steps 2-3 follow src/ra_server.erl:1950-1959 "
         "with transition_to => leader: the reply of L1-L2 is dropped while the WAL is down and re-processed by "
         "handle_leader/2 once it is back")

AGREED_COMMIT = [([4], 4), ([4, 3], 3), ([4, 4, 4], 4), ([4, 4, 3], 4), ([3, 4, 4], 4),
                 ([4, 2, 3], 3)]

if __name__ == "__main__":
    out = dict(
        reference="rabbitmq/ra 3.1.10",
        transcribed_from=["test/ra_server_SUITE.erl", "src/ra_server.erl:4225-4238",
                          "test/ra_log_2_SUITE.erl"],
        agreed_commit=[dict(indexes=i, expected=e) for i, e in AGREED_COMMIT],
        vectors=V)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "ra_server_suite_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {len(V)} vectors, {sum(len(v['steps']) for v in V)} steps -> {path}")
