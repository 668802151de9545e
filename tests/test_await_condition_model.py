"""Third restatement of handle_await_condition/2 (src/ra_server.erl:1916-1960) with the follower
catch-up predicate follower_catchup_cond/3 (:2202-2230) and has_log_entry_or_snapshot/3 (:3168-3183),
against the checker on random states.  The {next_event, Msg} re-processing is checked as "one decision
equals two steps through the modelled intermediate state" (see tests/test_election_model.py)."""
import numpy as np
import pytest

from ra_amd import abi
import fuzz
from test_election_model import Srv, random_msg, ELECTION_FLAGS
from test_repair_and_vote_models import log_term

KINDS = [abi.MSG_AER, abi.MSG_AER, abi.MSG_AER, abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC, abi.MSG_ELECTION_TIMEOUT,
         abi.MSG_AWAIT_TIMEOUT, abi.MSG_VOTE_RESULT, abi.MSG_AER_REPLY, abi.MSG_HEARTBEAT_RPC, abi.MSG_PRE_VOTE_RESULT]


def has_log_entry_or_snapshot(row, idx, term):                        # :3168-3183
    t = log_term(row, idx)
    if t is None:
        si = int(row["snapshot_index"])
        if si != abi.UNDEF_INT and si == idx:
            return "entry_ok" if int(row["snapshot_term"]) == term else "term_mismatch"
        return "missing"
    return "entry_ok" if t == term else "term_mismatch"


def catchup_pred(row, m):                                             # follower_catchup_cond/3 :2202-2230
    if int(m["kind"]) == abi.MSG_AER and int(m["term"]) >= int(row["current_term"]):
        r = has_log_entry_or_snapshot(row, int(m["a"]), int(m["b"]))
        if r == "entry_ok":
            return True
        if r == "term_mismatch":
            return int(row["cond_reason"]) == abi.COND_MISSING
        return False
    return False


@pytest.mark.parametrize("n,seed", [(3, 1), (5, 2), (7, 3), (2, 4)])
def test_await_condition_clauses_match_the_model(oracle_lib, n, seed):
    rng = np.random.default_rng(5000 + seed)
    G = 200
    st = fuzz.random_states(rng, G, n, max_runs=6)
    S = len(st)
    st["role"] = abi.ROLE_AWAIT_CONDITION
    st["cond_reason"] = rng.choice([abi.COND_MISSING, abi.COND_TERM_MISMATCH], size=S)
    st["cond_reply"] = rng.integers(0, 60, size=(S, 4))
    st["cond_leader"] = rng.integers(0, n, size=S)
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    before = cpu.get_state()
    msgs = []
    for sv in range(S):
        m = random_msg(rng, sv, before[sv], n, KINDS)
        if int(m["kind"][0]) == abi.MSG_AER:                          # prev index around the log's range
            ents = abi.log_entries(before[sv])
            r = rng.random()
            if ents and r < 0.6:
                i, t = ents[int(rng.integers(0, len(ents)))]
                m["a"], m["b"] = i, t if rng.random() < 0.7 else t + 1
            elif r < 0.8 and int(before[sv]["snapshot_index"]) != abi.UNDEF_INT:
                m["a"] = int(before[sv]["snapshot_index"])
                m["b"] = int(before[sv]["snapshot_term"]) + (0 if rng.random() < 0.7 else 1)
            else:
                m["a"], m["b"] = int(before[sv]["last_index"]) + int(rng.integers(1, 4)), int(before[sv]["last_term"])
            m["n_entries"] = 0
        msgs.append(m[0])
    msgs = np.array(msgs, dtype=abi.MSG_DTYPE)
    dec, _ = cpu.step(msgs)
    after = cpu.get_state()
    seen = {"released": 0, "held": 0, "timeout": 0, "vote": 0, "pre": 0, "election": 0, "invariant": 0}
    for m, d in zip(msgs, dec):
        sv = int(m["server"])
        row0, row1 = before[sv], after[sv]
        k, fl = int(m["kind"]), int(d["flags"])
        tag = f"N={n} server {sv} reason {int(row0['cond_reason'])} msg {m}"
        next_event = (k == abi.MSG_REQUEST_VOTE) or (k not in (abi.MSG_PRE_VOTE_RPC, abi.MSG_ELECTION_TIMEOUT,
                                                                  abi.MSG_AWAIT_TIMEOUT) and catchup_pred(row0, m))
        if next_event:
            # :1918-1919 / :1950-1955: transition to follower, then the message again
            mid = row0.copy()
            mid["role"], mid["cond_reason"] = abi.ROLE_FOLLOWER, abi.COND_NONE
            mid["status_mask"] = 0xFF                                # become(follower, ..) :2182-2192
            two = oracle_lib.Oracle(1, n)
            base = (sv // n) * n
            grp = before[base:base + n].copy()
            grp[sv - base] = mid
            two.set_state(0, grp)
            m2 = m.copy(); m2["server"] = sv - base
            d2, _ = two.step(np.array([m2], dtype=abi.MSG_DTYPE))
            got = two.get_state()[sv - base]
            if fl & abi.F_INVARIANT:
                # the reference would crash in the second half (e.g. ?assertNot(PLIdx < LastApplied)): the
                # engine reports it and leaves the server exactly as it was, await_condition included
                assert int(d2["flags"][0]) & abi.F_INVARIANT and int(d2["invariant"][0]) == int(d["invariant"]), tag
                assert row1.tobytes() == row0.tobytes(), tag
                seen["invariant"] += 1
                continue
            diff = [f for f in abi.SERVER_STATE_DTYPE.names if got[f].tobytes() != row1[f].tobytes()]
            assert not diff, (tag, diff, [(got[f], row1[f]) for f in diff])
            same = ~(abi.F_REPROCESSED | abi.F_ROLE_CHANGED | abi.F_LEADER_CHANGED)
            assert (fl & same) == (int(d2["flags"][0]) & same), (tag, hex(fl), hex(int(d2["flags"][0])))
            assert fl & abi.F_REPROCESSED, tag
            for f in ("reply_to", "reply_term", "reply_next_index", "reply_last_index", "reply_last_term"):
                assert int(d[f]) == int(d2[f][0]), (tag, f)
            seen["vote" if k == abi.MSG_REQUEST_VOTE else "released"] += 1
            continue
        s = Srv(row0)
        if k == abi.MSG_PRE_VOTE_RPC:                                 # :1920-1921
            s.process_pre_vote(abi.ROLE_AWAIT_CONDITION, m)
            seen["pre"] += 1
        elif k == abi.MSG_ELECTION_TIMEOUT:                           # :1922-1931
            if s.voter:
                s.call_for_election_pre_vote(int(m["c"]))
            seen["election"] += 1
        elif k == abi.MSG_AWAIT_TIMEOUT:                              # :1932-1945: the stored failed reply goes out
            s.role = abi.ROLE_FOLLOWER
            cr = [int(x) for x in row0["cond_reply"]]
            s.reply = ("aer", int(row0["cond_leader"]), cr[0], cr[1], cr[2], cr[3])
            seen["timeout"] += 1
        else:
            seen["held"] += 1                                         # predicate false: stay, no effects (:1956-1959)
        assert int(row1["role"]) == s.role, tag
        assert int(row1["current_term"]) == s.term and int(row1["voted_for"]) == s.voted_for, tag
        assert int(row1["votes"]) == s.votes and int(row1["pre_vote_token"]) == s.token, tag
        for f in ("commit_index", "last_applied", "last_index", "last_term", "last_written_index", "n_runs", "first_index"):
            assert int(row0[f]) == int(row1[f]), (tag, f)
        if s.role == abi.ROLE_AWAIT_CONDITION:
            for f in ("cond_reason", "cond_leader"):
                assert int(row0[f]) == int(row1[f]), (tag, f)
            assert np.array_equal(row0["cond_reply"], row1["cond_reply"]), tag
        else:
            assert int(row1["cond_reason"]) == abi.COND_NONE, tag
        want = s.flags
        if s.reply:
            want |= abi.F_REPLY | (abi.F_REPLY_PRE_VOTE if s.reply[0] == "pre" else 0)
            assert int(d["reply_to"]) == s.reply[1] and int(d["reply_term"]) == s.reply[2], tag
            if s.reply[0] == "pre":
                assert int(d["reply_next_index"]) == s.reply[3], tag
                want |= abi.F_REPLY_SUCCESS if s.reply[4] else 0
            else:
                assert (int(d["reply_next_index"]), int(d["reply_last_index"]), int(d["reply_last_term"])) == s.reply[3:], tag
        if s.requests:
            pre, term, token, li, lt = s.requests
            want |= abi.F_SEND_VOTE_REQUESTS | (abi.F_PRE_VOTE_REQS if pre else 0)
            assert int(d["reply_term"]) == term and int(d["reply_last_index"]) == li, tag
        assert (fl & ELECTION_FLAGS) == want, (tag, hex(fl & ELECTION_FLAGS), hex(want))
    assert seen["released"] > 10 and seen["held"] > 10 and seen["timeout"] > 5 and seen["vote"] > 5, seen


# ------------------------------------------------------------------------------------------------------------------
# The two wal_down conditions, written the way the reference writes them: a condition MAP with a predicate fun, an
# optional transition_to and an optional timeout map, and handle_await_condition/2 reading them with maps:get/3.

LEADER, FOLLOWER = abi.ROLE_LEADER, abi.ROLE_FOLLOWER


def wal_down_condition(msg, row, can_write):                          # :2232-2233  {ra_log:can_write(Log), State}
    return can_write


def condition_map(row, n):
    """#{predicate_fun => fun wal_down_condition/2} of a follower whose write was refused (:1377-1385); a leader whose
    append raised wal_down adds transition_to => leader and timeout => #{duration, effects, transition_to} with
    CondEffs = the first key of maps:to_list(maps:remove(Self, Cluster)), or [] (:660-668)."""
    if int(row["cond_reason"]) == abi.COND_WAL_DOWN:
        return {"predicate_fun": wal_down_condition}
    me = int(row["self"])
    others = [i for i in range(8) if (int(row["present_mask"]) >> i) & 1 and i != me]
    cond_effs = [("next_event", "cast", ("transfer_leadership", others[0]))] if others else []
    return {"predicate_fun": wal_down_condition, "transition_to": LEADER,
            "timeout": {"duration": 5000, "effects": cond_effs, "transition_to": LEADER}}


def handle_await_condition(msg, row, n, can_write):
    """:1916-1959, the clauses in their order; returns (next state name, condition removed?, effects) or the name of
    the clause that is somebody else's (request_vote / pre_vote / election_timeout / ra_log_event)."""
    k = int(msg["kind"])
    cond = condition_map(row, n)
    if k == abi.MSG_REQUEST_VOTE:                                     # :1918-1919
        return FOLLOWER, False, [("next_event", msg)]
    if k == abi.MSG_PRE_VOTE_RPC:
        return "process_pre_vote"                                     # :1920-1921
    if k == abi.MSG_ELECTION_TIMEOUT:
        return "election_timeout"                                     # :1922-1931
    if k == abi.MSG_AWAIT_TIMEOUT:                                    # :1932-1945
        if cond["predicate_fun"](msg, row, can_write):
            return cond.get("transition_to", FOLLOWER), True, []
        timeout = cond.get("timeout", {})
        return timeout.get("transition_to", FOLLOWER), True, timeout.get("effects", [])
    if k in (abi.MSG_WRITTEN, abi.MSG_SNAPSHOT_WRITTEN):
        return "ra_log_event"                                         # :1946-1949
    if cond["predicate_fun"](msg, row, can_write):                    # :1950-1955
        return cond.get("transition_to", FOLLOWER), True, [("next_event", msg)]
    return abi.ROLE_AWAIT_CONDITION, False, []                        # :1956-1959


WAL_KINDS = [abi.MSG_AER, abi.MSG_AER_REPLY, abi.MSG_AER_REPLY, abi.MSG_REQUEST_VOTE, abi.MSG_AWAIT_TIMEOUT, abi.MSG_AWAIT_TIMEOUT,
             abi.MSG_VOTE_RESULT, abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY, abi.MSG_PRE_VOTE_RESULT]


@pytest.mark.parametrize("n,seed", [(3, 11), (5, 12), (7, 13), (1, 14)])
def test_wal_down_conditions_match_the_condition_map_model(oracle_lib, n, seed):
    rng = np.random.default_rng(6000 + seed)
    G = 200
    st = fuzz.random_states(rng, G, n, max_runs=6)
    S = len(st)
    st["role"] = abi.ROLE_AWAIT_CONDITION
    st["cond_reason"] = rng.choice([abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER], size=S)
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    before = cpu.get_state()
    msgs = np.array([random_msg(rng, sv, before[sv], n, WAL_KINDS)[0] for sv in range(S)], dtype=abi.MSG_DTYPE)
    msgs["flags"] |= np.where(rng.random(S) < 0.5, abi.MF_CAN_WRITE, 0).astype(msgs["flags"].dtype)
    dec, _ = cpu.step(msgs)
    after = cpu.get_state()
    seen = {"held": 0, "timeout_transfer": 0, "timeout_plain": 0, "to_leader": 0, "to_follower": 0, "vote": 0, "invariant": 0}
    for m, d in zip(msgs, dec):
        sv = int(m["server"])
        row0, row1 = before[sv], after[sv]
        fl = int(d["flags"])
        tag = f"N={n} server {sv} reason {int(row0['cond_reason'])} msg {m}"
        nxt, removed, effects = handle_await_condition(m, row0, n, bool(int(m["flags"]) & abi.MF_CAN_WRITE))
        transfer = any(len(e) == 3 and e[1] == "cast" for e in effects)   # {next_event, cast, {transfer_leadership, _}}
        reprocess = any(len(e) == 2 for e in effects)                      # {next_event, Msg}
        if not reprocess:
            assert bool(fl & abi.F_TRANSFER_LEADERSHIP) == transfer, tag
            assert not fl & abi.F_REPROCESSED and int(row1["role"]) == nxt, tag
            assert int(row1["cond_reason"]) == (abi.COND_NONE if removed else int(row0["cond_reason"])), tag
            want = row0.copy()
            want["role"], want["cond_reason"] = row1["role"], row1["cond_reason"]
            if nxt == FOLLOWER and removed:
                want["status_mask"] = 0xFF; want["backoff_mask"] = 0  # become(follower, ..) :2182-2192
            assert row1.tobytes() == want.tobytes(), tag              # nothing else moves
            assert (fl & ~(abi.F_ROLE_CHANGED | abi.F_TRANSFER_LEADERSHIP)) == 0, (tag, hex(fl))
            if int(m["kind"]) == abi.MSG_AWAIT_TIMEOUT:
                seen["timeout_transfer" if transfer else "timeout_plain"] += 1
            else:
                seen["held"] += 1
            continue
        # {next_event, Msg}: one decision = the transition, then the message in the new state (which may itself
        # step down and re-process: handle_leader/2 -> handle_follower/2)
        mid = row0.copy()
        mid["role"], mid["cond_reason"] = nxt, abi.COND_NONE
        if nxt == FOLLOWER:
            mid["status_mask"] = 0xFF; mid["backoff_mask"] = 0
        two = oracle_lib.Oracle(1, n)
        base = (sv // n) * n
        grp = before[base:base + n].copy()
        grp[sv - base] = mid
        two.set_state(0, grp)
        m2 = m.copy(); m2["server"] = sv - base
        d2, _ = two.step(np.array([m2], dtype=abi.MSG_DTYPE))
        got = two.get_state()[sv - base]
        two.close()
        if fl & abi.F_INVARIANT:
            assert int(d2["flags"][0]) & abi.F_INVARIANT and int(d2["invariant"][0]) == int(d["invariant"]), tag
            assert row1.tobytes() == row0.tobytes(), tag
            seen["invariant"] += 1
            continue
        diff = [f for f in abi.SERVER_STATE_DTYPE.names if got[f].tobytes() != row1[f].tobytes()]
        assert not diff, (tag, diff, [(got[f], row1[f]) for f in diff])
        same = ~(abi.F_REPROCESSED | abi.F_ROLE_CHANGED | abi.F_LEADER_CHANGED)
        assert (fl & same) == (int(d2["flags"][0]) & same), (tag, hex(fl), hex(int(d2["flags"][0])))
        assert fl & abi.F_REPROCESSED and not fl & abi.F_TRANSFER_LEADERSHIP, tag
        for f in ("reply_to", "reply_term", "reply_next_index", "reply_last_index", "reply_last_term", "n_rpcs"):
            assert int(d[f]) == int(d2[f][0]), (tag, f)
        seen["vote" if int(m["kind"]) == abi.MSG_REQUEST_VOTE else "to_leader" if nxt == LEADER else "to_follower"] += 1
    cpu.close()
    assert seen["held"] > 10 and seen["timeout_plain"] > 5 and seen["to_follower"] > 10 and seen["to_leader"] > 10, seen
    assert (seen["timeout_transfer"] > 5) == (n > 1), seen
