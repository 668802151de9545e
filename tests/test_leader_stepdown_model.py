"""Third restatement of the leader's clauses for messages from other would-be leaders and candidates
(handle_leader/2, src/ra_server.erl:835-960): append_entries_rpc and heartbeat_rpc by term (abdicate /
exit / failed reply), heartbeat_reply of a higher term, request_vote_rpc and pre_vote_rpc (abdicate for
a known peer, ignore an unknown one, refuse, or enforce leadership with make_all_rpcs), vote results.
Against the checker on random leaders; {next_event, Msg} as "one decision = two steps"."""
import numpy as np
import pytest

from ra_amd import abi
import fuzz
from test_election_model import Srv, random_msg, ELECTION_FLAGS

KINDS = [abi.MSG_AER, abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY, abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC,
         abi.MSG_VOTE_RESULT, abi.MSG_PRE_VOTE_RESULT]


@pytest.mark.parametrize("n,seed", [(3, 1), (5, 2), (7, 3), (4, 4)])
def test_leader_stepdown_clauses_match_the_model(oracle_lib, n, seed):
    rng = np.random.default_rng(6000 + seed)
    G = 200
    st = fuzz.random_states(rng, G, n, max_runs=6)
    S = len(st)
    st["role"] = abi.ROLE_LEADER
    st["cond_reason"] = 0
    st["leader_id"] = st["self"]
    st["voted_for"] = st["self"]
    # some groups have lost a member: a sender the leader does not know
    gone = rng.random(S) < 0.15
    for sv in np.flatnonzero(gone):
        others = [i for i in range(n) if i != int(st["self"][sv])]
        st["present_mask"][sv] &= np.uint8(~(1 << int(rng.choice(others))) & 0xFF)
    # and some have peers in snapshot back-off (not normal, not self)
    for sv in np.flatnonzero(rng.random(S) < 0.4):
        st["status_mask"][sv] = int(rng.integers(0, 256))
        st["backoff_mask"][sv] = ~int(st["status_mask"][sv]) & int(st["present_mask"][sv]) & ~(1 << int(st["self"][sv])) & 0xFF
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    before = cpu.get_state()
    msgs = np.array([random_msg(rng, sv, before[sv], n, KINDS)[0] for sv in range(S)], dtype=abi.MSG_DTYPE)
    dec, rpcs = cpu.step(msgs)
    after = cpu.get_state()
    n_rpcs = np.bincount(rpcs["msg_index"], minlength=S) if len(rpcs) else np.zeros(S, dtype=int)
    seen = {"abdicate": 0, "exit": 0, "reply": 0, "unknown": 0, "enforce": 0, "ignored": 0, "backoff": 0}
    for i, (m, d) in enumerate(zip(msgs, dec)):
        sv = int(m["server"])
        row0, row1 = before[sv], after[sv]
        k, term, frm, fl = int(m["kind"]), int(m["term"]), int(m["from"]), int(d["flags"])
        cur = int(row0["current_term"])
        known = bool((int(row0["present_mask"]) >> frm) & 1)
        tag = f"N={n} server {sv} cur {cur} msg {m}"
        s = Srv(row0)
        next_event = enforce = False
        want_inv = 0
        if k in (abi.MSG_AER, abi.MSG_HEARTBEAT_RPC):
            if term > cur:                                            # :835-844, :880-889
                next_event = True
            elif term == cur:                                         # :845-849, :898-903: exit(...)
                want_inv = abi.INV_LEADER_SAW_AER_SAME_TERM if k == abi.MSG_AER else abi.INV_LEADER_SAW_HEARTBEAT_SAME_TERM
            elif k == abi.MSG_AER:                                    # :850-854
                s.aer_reply_false(frm)
            else:                                                     # :890-897
                s.reply = ("hb", frm, cur, int(m["a"]))
        elif k == abi.MSG_HEARTBEAT_REPLY:
            if term == cur:
                continue                                              # the quorum: tests/test_query_quorum_model.py
            if term > cur:                                            # :918-924
                s.leader_id = abi.NONE
                s.update_term(term)
                s.role = abi.ROLE_FOLLOWER
        elif k in (abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC):
            if term > cur:                                            # :926-940, :944-958
                if known:
                    next_event = True
                else:
                    seen["unknown"] += 1
            elif k == abi.MSG_REQUEST_VOTE:                           # :941-943
                s.reply = ("vote", frm, cur, False)
            else:                                                     # :961-966 make_all_rpcs: enforce leadership
                enforce = True
        if want_inv:
            assert fl & abi.F_INVARIANT and int(d["invariant"]) == want_inv, tag
            assert row1.tobytes() == row0.tobytes(), tag
            seen["exit"] += 1
            continue
        if next_event:
            mid = row0.copy()
            mid["role"], mid["leader_id"], mid["status_mask"] = abi.ROLE_FOLLOWER, abi.NONE, 0xFF
            mid["current_term"], mid["voted_for"] = term, abi.NONE
            mid["peer_query_index"] = 0
            two = oracle_lib.Oracle(1, n)
            base = (sv // n) * n
            grp = before[base:base + n].copy()
            grp[sv - base] = mid
            two.set_state(0, grp)
            m2 = m.copy(); m2["server"] = sv - base
            d2, _ = two.step(np.array([m2], dtype=abi.MSG_DTYPE))
            got = two.get_state()[sv - base]
            if fl & abi.F_INVARIANT:                                  # the second half would crash: all undone
                assert int(d2["flags"][0]) & abi.F_INVARIANT and row1.tobytes() == row0.tobytes(), tag
                continue
            diff = [f for f in abi.SERVER_STATE_DTYPE.names if got[f].tobytes() != row1[f].tobytes()]
            assert not diff, (tag, diff)
            assert fl & abi.F_REPROCESSED and fl & abi.F_ROLE_CHANGED and fl & abi.F_PERSIST, tag
            same = ~(abi.F_REPROCESSED | abi.F_ROLE_CHANGED | abi.F_PERSIST | abi.F_LEADER_CHANGED)
            assert (fl & same) == (int(d2["flags"][0]) & same), (tag, hex(fl), hex(int(d2["flags"][0])))
            for f in ("reply_to", "reply_term", "reply_next_index", "reply_last_index", "reply_last_term"):
                assert int(d[f]) == int(d2[f][0]), (tag, f)
            seen["abdicate"] += 1
            continue
        if enforce:
            if fl & abi.F_INVARIANT:
                assert int(d["invariant"]) == abi.INV_PIPELINE_PREV_UNDEFINED, tag
                continue
            # make_all_rpcs/1 :2353-2367: peers that are normal or in {snapshot_backoff, _}
            normal = [p for p in range(n) if p != s.me and (int(row0["present_mask"]) >> p) & 1
                      and ((int(row0["status_mask"]) >> p) & 1 or (int(row0["backoff_mask"]) >> p) & 1)]
            backed_off = sum(1 << p for p in normal if not (int(row0["status_mask"]) >> p) & 1)
            assert int(d["cancel_backoff"]) == backed_off, tag
            assert bool(fl & abi.F_CANCEL_SNAPSHOT_RETRY) == bool(backed_off), tag
            seen["backoff"] += bool(backed_off)
            assert int(n_rpcs[i]) == len(normal) == int(d["n_rpcs"]), tag
            mine = rpcs[rpcs["msg_index"] == i]
            assert sorted(int(p) for p in mine["peer"]) == normal, tag
            for r in mine:                                            # make_rpcs_for/2: batch of 1 from next_index - 1
                ni = int(row0["next_index"][int(r["peer"])])
                if int(r["kind"]) == abi.RPC_AER:
                    assert int(r["prev_log_index"]) == ni - 1 and int(r["n_entries"]) == min(1, max(0, s.last[0] - ni + 1)), tag
                    assert int(r["term"]) == cur and int(r["leader_commit"]) == int(row0["commit_index"]), tag
            for f in ("next_index", "match_index", "commit_index_sent"):
                assert np.array_equal(row0[f], row1[f]), (tag, f)     # make_rpcs_for does not advance the peers
            assert int(row1["role"]) == abi.ROLE_LEADER and int(row1["current_term"]) == cur, tag
            seen["enforce"] += 1
            continue
        # plain clauses
        assert int(row1["role"]) == s.role and int(row1["current_term"]) == s.term, tag
        assert int(row1["voted_for"]) == s.voted_for and int(row1["leader_id"]) == s.leader_id, tag
        if s.role == abi.ROLE_FOLLOWER:
            assert int(row1["status_mask"]) == 0xFF and not row1["peer_query_index"].any(), tag
        for f in ("commit_index", "last_applied", "last_index", "last_term", "n_runs"):
            assert int(row0[f]) == int(row1[f]), (tag, f)
        want = s.flags
        if s.reply:
            want |= abi.F_REPLY | {"vote": abi.F_REPLY_VOTE, "hb": abi.F_REPLY_HEARTBEAT, "aer": 0}[s.reply[0]]
            assert int(d["reply_to"]) == s.reply[1] and int(d["reply_term"]) == s.reply[2], tag
            if s.reply[0] == "aer":
                assert (int(d["reply_next_index"]), int(d["reply_last_index"]), int(d["reply_last_term"])) == s.reply[3:], tag
            elif s.reply[0] == "hb":
                assert int(d["reply_next_index"]) == s.reply[3], tag
            seen["reply"] += 1
        else:
            seen["ignored"] += 1
        assert (fl & (ELECTION_FLAGS & ~abi.F_UNHANDLED)) == want, (tag, hex(fl), hex(want))
        assert int(d["n_rpcs"]) == 0, tag
    assert all(v > 3 for v in seen.values()), seen
