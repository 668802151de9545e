"""ra_amd/effects.py: the reference's records and effects on either side of the batched path.  The
scenarios below are the reference's own (test/ra_server_SUITE.erl), written the way its tests are: send a
record to a server in a role, match the next role, the state fields and the effects list."""
import numpy as np
import pytest

from ra_amd import abi, effects as fx
from ra_amd.effects import (AppendEntriesRpc, AppendEntriesReply, RequestVoteRpc, RequestVoteResult, PreVoteRpc,
                            PreVoteResult, HeartbeatRpc, HeartbeatReply, Written, Commands, ElectionTimeout)

N1, N2, N3 = 0, 1, 2


class Server:
    """ra_server:handle_<role>/2 for one member of a three-member group, backed by the checker."""

    def __init__(self, oracle_lib, states, me):
        self.o = oracle_lib.Oracle(1, 3)
        self.o.set_state(0, states)
        self.me = me

    def handle(self, record, from_slot=abi.NONE):
        m = fx.encode(self.me, record, from_slot)
        dec, rpcs = self.o.step(np.array([m], dtype=abi.MSG_DTYPE))
        st = self.o.get_state()[self.me]
        return abi.ROLE_NAMES[int(st["role"])], st, fx.decode(m, dec[0], list(rpcs), st, 3)


def empty_state(oracle_lib, me):                      # empty_state(3, Id), SUITE:4139-4149
    return Server(oracle_lib, abi.empty_server_states(1, 3), me)


def base_state(oracle_lib):                           # base_state(3), SUITE:4151-4192: n1 leads term 5, log [1:1, 2:3, 3:5]
    st = abi.empty_server_states(1, 3)
    for i in range(3):
        st["current_term"][i] = 5
        st["commit_index"][i] = st["last_applied"][i] = 3
        abi.set_log(st, i, [(1, 1), (2, 3), (3, 5)])
        st["next_index"][i, :3] = 4
        st["match_index"][i, :3] = 3
        st["leader_id"][i] = N1
    st["role"][N1] = abi.ROLE_LEADER
    return Server(oracle_lib, st, N1)


def test_follower_aer_1(oracle_lib):
    """SUITE:383-423 follower_aer_1, scenario 1 (self = n1 as in the reference)."""
    s = empty_state(oracle_lib, N1)
    role, st, effs = s.handle(AppendEntriesRpc(term=1, leader_id=N1, prev_log_index=0, prev_log_term=0,
                                               leader_commit=0, entries=((1, 1),)))
    assert role == "follower" and (int(st["leader_id"]), int(st["current_term"])) == (N1, 1)
    assert (int(st["commit_index"]), int(st["last_applied"])) == (0, 0)
    assert ("record_leader_msg", N1) in effs and not [e for e in effs if e[0] == "cast"]   # replies only on `written`
    role, st, _ = s.handle(AppendEntriesRpc(term=1, leader_id=N1, prev_log_index=1, prev_log_term=1, leader_commit=1,
                                            entries=((2, 1),)))
    assert (int(st["commit_index"]), int(st["last_applied"])) == (1, 1)
    role, st, effs = s.handle(Written(term=1, first=1, last=1))
    assert effs == [("cast", N1, (N1, AppendEntriesReply(term=1, success=True, next_index=3, last_index=1, last_term=1)))]
    role, st, _ = s.handle(AppendEntriesRpc(term=1, leader_id=N1, prev_log_index=2, prev_log_term=1, leader_commit=3,
                                            entries=((3, 1),)))
    assert (int(st["commit_index"]), int(st["last_applied"])) == (3, 3)


def test_follower_request_vote_and_pre_vote(oracle_lib):
    """follower_request_vote (SUITE:1240-1290) and follower_pre_vote: grant to an up-to-date candidate,
    refuse a stale term with the current term."""
    s = base_state(oracle_lib); s.me = N2
    role, st, effs = s.handle(RequestVoteRpc(term=6, candidate_id=N3, last_log_index=3, last_log_term=5))
    assert role == "follower" and int(st["voted_for"]) == N3 and int(st["current_term"]) == 6
    assert effs == [("reply", RequestVoteResult(term=6, vote_granted=True))]
    role, st, effs = s.handle(RequestVoteRpc(term=5, candidate_id=N1, last_log_index=3, last_log_term=5))
    assert effs == [("reply", RequestVoteResult(term=6, vote_granted=False))]
    role, st, effs = s.handle(PreVoteRpc(term=6, token=77, candidate_id=N1, last_log_index=3, last_log_term=5))
    assert effs == [("reply", PreVoteResult(term=6, token=77, vote_granted=True))]
    assert int(st["voted_for"]) == N3                               # a pre-vote never sets voted_for


def test_election_from_timeout_to_noop(oracle_lib):
    """pre_vote_election / candidate_election (SUITE:1514-1632): timeout -> pre-vote requests -> quorum ->
    vote requests -> quorum -> leader, whose first act is the noop command."""
    s = base_state(oracle_lib); s.me = N2
    role, st, effs = s.handle(ElectionTimeout(token=9))
    assert role == "pre_vote"
    (tag, reqs), = [e for e in effs if e[0] == "send_vote_requests"]
    assert reqs == [(N1, PreVoteRpc(5, 9, N2, 3, 5, 0)), (N3, PreVoteRpc(5, 9, N2, 3, 5, 0))]
    role, st, effs = s.handle(PreVoteResult(term=5, token=9, vote_granted=True), from_slot=N3)
    assert role == "candidate" and int(st["current_term"]) == 6 and int(st["voted_for"]) == N2
    (tag, reqs), = [e for e in effs if e[0] == "send_vote_requests"]
    assert reqs == [(N1, RequestVoteRpc(6, N2, 3, 5)), (N3, RequestVoteRpc(6, N2, 3, 5))]
    role, st, effs = s.handle(RequestVoteResult(term=6, vote_granted=True), from_slot=N3)
    assert role == "leader" and int(st["leader_id"]) == N2
    assert ("next_event", "cast", ("command", "noop")) in effs
    role, st, effs = s.handle(Commands(1, noop=True))
    rpcs = sorted((e[1], e[2]) for e in effs if e[0] == "send_rpc")
    assert rpcs == [(N1, AppendEntriesRpc(6, N2, 3, 3, 5, ((4, 6),))), (N3, AppendEntriesRpc(6, N2, 3, 3, 5, ((4, 6),)))]


def test_leader_replication_round_trip(oracle_lib):
    """leader_receives_append_entries_reply / command (SUITE:2105-2180): a command goes out to both
    peers, the written event plus one success reply commit it, the new commit index is pipelined."""
    s = base_state(oracle_lib)
    role, st, effs = s.handle(Commands(1))
    assert sorted((e[1], e[2]) for e in effs if e[0] == "send_rpc") == [
        (N2, AppendEntriesRpc(5, N1, 3, 3, 5, ((4, 5),))), (N3, AppendEntriesRpc(5, N1, 3, 3, 5, ((4, 5),)))]
    role, st, effs = s.handle(Written(5, 4, 4))
    assert int(st["commit_index"]) == 3                             # only the leader has it
    role, st, effs = s.handle(AppendEntriesReply(term=5, success=True, next_index=5, last_index=4, last_term=5),
                              from_slot=N2)
    assert int(st["commit_index"]) == 4 and int(st["last_applied"]) == 4
    assert ("aux", "eval") in effs and ("next_event", "info", fx.PIPELINE_RPCS) in effs
    role, st, effs = s.handle(fx.PIPELINE_RPCS)                     # :793-801: the new commit index goes out
    updates = [(e[1], e[2]) for e in effs if e[0] == "send_rpc"]
    assert updates and all(r.leader_commit == 4 and r.entries == () for _, r in updates)
    # a reply from a higher term: the leader abdicates (SUITE leader_..._higher_term)
    role, st, effs = s.handle(AppendEntriesReply(term=6, success=False, next_index=4, last_index=3, last_term=5),
                              from_slot=N3)
    assert role == "follower" and int(st["current_term"]) == 6 and int(st["leader_id"]) == abi.NONE


def test_heartbeats_and_tick(oracle_lib):
    """leader_heartbeat / follower_heartbeat (SUITE:3588-3696) and the leader's tick."""
    s = base_state(oracle_lib)
    role, st, effs = s.handle(fx.CONSISTENT_QUERY)
    hb = sorted((e[1], e[2]) for e in effs if e[0] == "send_rpc")
    assert hb == [(N2, HeartbeatRpc(query_index=1, term=5, leader_id=N1)), (N3, HeartbeatRpc(1, 5, N1))]
    role, st, effs = s.handle(HeartbeatReply(query_index=1, term=5), from_slot=N2)
    assert ("query_quorum", 1) in effs
    # tick: stale_peers/1 -- base_state's peers have commit_index_sent = 0 < commit_index = 3, so each gets
    # the batch-of-one rpc of make_rpcs_for/2 (empty here: next_index is past the log), peers not advanced
    role, st, effs = s.handle(fx.TICK_TIMEOUT)
    assert sorted((e[1], e[2]) for e in effs if e[0] == "send_rpc" and isinstance(e[2], AppendEntriesRpc)) == [
        (N2, AppendEntriesRpc(5, N1, 3, 3, 5, ())), (N3, AppendEntriesRpc(5, N1, 3, 3, 5, ()))]
    assert [int(x) for x in st["commit_index_sent"][:3]] == [0, 0, 0]
    f = base_state(oracle_lib); f.me = N2
    role, st, effs = f.handle(HeartbeatRpc(query_index=3, term=5, leader_id=N1))
    assert effs == [("cast", N1, (N2, HeartbeatReply(query_index=3, term=5)))]
    # the reference's exit/1 surfaces as an effect the shell must act on
    role, st, effs = base_state(oracle_lib).handle(HeartbeatRpc(query_index=3, term=5, leader_id=N2))
    assert effs == [("exit", abi.INV_LEADER_SAW_HEARTBEAT_SAME_TERM)]


def test_encode_rejects_what_one_message_cannot_carry():
    with pytest.raises(ValueError):
        fx.encode(0, AppendEntriesRpc(1, 0, 0, 0, 0, ((1, 1), (2, 2), (3, 3))))       # three term runs
    with pytest.raises(ValueError):
        fx.encode(0, AppendEntriesRpc(1, 0, 0, 3, 1, ((2, 1),)))                      # entries at or below prev
    with pytest.raises(TypeError):
        fx.encode(0, "install_snapshot_rpc")
    pieces = fx.split_entries(AppendEntriesRpc(7, 1, 9, 10, 2, ((11, 2), (12, 3), (13, 3), (14, 4), (15, 5), (16, 5))))
    assert [p.entries for p in pieces] == [((11, 2), (12, 3), (13, 3)), ((14, 4), (15, 5), (16, 5))]
    assert [(p.prev_log_index, p.prev_log_term) for p in pieces] == [(10, 2), (13, 3)]
    m = fx.encode(4, pieces[1])
    assert (int(m["a"]), int(m["b"]), int(m["n_entries"]), int(m["n_run0"]), int(m["run0_term"]), int(m["run1_term"])) == \
        (13, 3, 3, 1, 4, 5)


def test_leader_pre_vote_sends_rpc_to_backoff_peer(oracle_lib):
    """SUITE:2665-2688 leader_pre_vote_sends_snapshot_to_backoff_peer: answering a pre_vote_rpc of its own
    term the leader enforces its leadership with make_all_rpcs/1 -- a cancel_snapshot_retry_timer effect and
    a send_rpc for the peer in {snapshot_backoff, 2}."""
    s = base_state(oracle_lib)
    st = s.o.get_state()
    st["status_mask"][N1] &= ~(1 << N2) & 0xFF
    st["backoff_mask"][N1] = 1 << N2
    s.o.set_state(0, st)
    role, st, effs = s.handle(PreVoteRpc(term=5, token=123, candidate_id=N1, last_log_index=3, last_log_term=5))
    assert role == "leader"
    assert ("cancel_snapshot_retry_timer", N2) in effs
    assert any(e[0] == "send_rpc" and e[1] == N2 and isinstance(e[2], AppendEntriesRpc) for e in effs)
    assert int(st["backoff_mask"]) == 1 << N2                         # statuses only change on role changes
    role, st, effs = s.handle(AppendEntriesReply(term=6, success=False, next_index=4, last_index=3, last_term=5),
                              from_slot=N3)
    assert role == "follower" and int(st["status_mask"]) == 0xFF and int(st["backoff_mask"]) == 0
