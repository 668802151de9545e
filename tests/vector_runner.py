"""Runs the transcribed reference vectors (tests/golden/ra_server_suite_vectors.json) against any
engine exposing   set_state(first, states) / get_state(first, n) / step(msgs) -> (dec, rpcs).
Both the CPU checker (oracle.oracle.Oracle) and the HIP engine (ra_amd.engine.RaGpuBatch) do."""
from __future__ import annotations

import json
import os

import numpy as np

from ra_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                      "ra_server_suite_vectors.json")

ROLE = {"follower": abi.ROLE_FOLLOWER, "candidate": abi.ROLE_CANDIDATE, "leader": abi.ROLE_LEADER,
        "pre_vote": abi.ROLE_PRE_VOTE, "await_condition": abi.ROLE_AWAIT_CONDITION}
COND = {"none": abi.COND_NONE, "missing": abi.COND_MISSING, "term_mismatch": abi.COND_TERM_MISMATCH,
        "wal_down": abi.COND_WAL_DOWN, "wal_down_leader": abi.COND_WAL_DOWN_LEADER}
FLAG = {k[2:]: getattr(abi, k) for k in dir(abi) if k.startswith("F_")}


def load():
    with open(GOLDEN) as f:
        return json.load(f)


def slot(name):
    if name is None:
        return abi.NONE
    return int(name[1:]) - 1


def initial_state(v) -> np.ndarray:
    """One-group state array (n_members rows); only row slot(self) is exercised."""
    n = v["n_members"]
    st = abi.empty_server_states(1, n)
    i = slot(v["self"])
    if v["init"] == "base":
        # base_state/2, test/ra_server_SUITE.erl:4151-4192
        st["current_term"][i] = 5
        st["commit_index"][i] = 3
        st["last_applied"][i] = 3
        st["leader_id"][i] = slot("n1")
        abi.set_log(st, i, [(0, 0), (1, 1), (2, 3), (3, 5)], last_written=(3, 5))
        st["next_index"][i, :n] = 4
        st["match_index"][i, :n] = 3
    tw = v.get("tweak") or {}
    for k in ("commit_index", "last_applied", "current_term", "votes", "pre_vote_token", "machine_version",
              "effective_machine_version", "query_index"):
        if k in tw:
            st[k][i] = tw[k]
    if "voted_for" in tw:
        st["voted_for"][i] = slot(tw["voted_for"])
    if "leader_id" in tw:
        st["leader_id"][i] = slot(tw["leader_id"])
    if "role" in tw:
        st["role"][i] = ROLE[tw["role"]]
    if "cond_reason" in tw:
        st["cond_reason"][i] = COND[tw["cond_reason"]]
    if "cond_reply" in tw:
        st["cond_reply"][i] = tw["cond_reply"]
    if "cond_leader" in tw:
        st["cond_leader"][i] = slot(tw["cond_leader"])
    if tw.get("self_nonvoter"):
        st["self_nonvoter"][i] = 1
    if "members_present" in tw:
        st["present_mask"][i] = sum(1 << slot(m) for m in tw["members_present"])
    if "nonvoters" in tw:
        m = int(st["voter_mask"][i])
        for nm in tw["nonvoters"]:
            m &= ~(1 << slot(nm))
        st["voter_mask"][i] = m
    if "peers_not_normal" in tw:                      # status =/= normal: sending_snapshot, disconnected, ...
        m = int(st["status_mask"][i])
        for nm in tw["peers_not_normal"]:
            m &= ~(1 << slot(nm))
        st["status_mask"][i] = m
    if "peers_backoff" in tw:                         # status = {snapshot_backoff, _}
        for nm in tw["peers_backoff"]:
            st["status_mask"][i] = int(st["status_mask"][i]) & ~(1 << slot(nm))
            st["backoff_mask"][i] = int(st["backoff_mask"][i]) | (1 << slot(nm))
    if "log" in tw:
        abi.set_log(st, i, [tuple(e) for e in tw["log"]],
                    last_written=tuple(tw["last_written"]) if "last_written" in tw else None)
    if "install_snapshot" in tw:
        # ra_log_memory:install_snapshot/4 (test/ra_log_memory.erl:236-247): entries <= Index
        # dropped, last_index = Index, last_written = snapshot = {Index, Term}
        si, stm = tw["install_snapshot"]
        keep = [e for e in abi.log_entries(st[i]) if e[0] > si]
        abi.set_log(st, i, keep, last_written=(si, stm), snapshot=(si, stm))
        if not keep:
            st["last_index"][i], st["last_term"][i] = si, stm
    for name, p in (tw.get("peers") or {}).items():
        j = slot(name)
        for k, val in p.items():
            st[k][i, j] = val
    return st


def make_msg(v, m) -> np.ndarray:
    out = np.zeros(1, dtype=abi.MSG_DTYPE)
    out["server"] = slot(v["self"])
    out["from"] = slot(m["from"]) if "from" in m else abi.NONE
    k = m["kind"]
    if k == "aer":
        out["kind"] = abi.MSG_AER
        out["term"] = m["term"]
        out["a"], out["b"] = m["prev"]
        out["c"] = m["commit"]
        ents = m["entries"]
        out["n_entries"] = len(ents)
        if ents:
            first = ents[0][0]
            for q, (idx, _t) in enumerate(ents):
                assert idx == first + q
            gap = first - (m["prev"][0] + 1)
            assert 0 <= gap < 256
            out["gap"] = gap
            terms = [t for _, t in ents]
            n0 = 0
            while n0 < len(terms) and terms[n0] == terms[0]:
                n0 += 1
            rest = terms[n0:]
            assert all(t == rest[0] for t in rest), "at most two term runs per AER"
            out["n_run0"] = n0
            out["run0_term"] = terms[0]
            out["run1_term"] = rest[0] if rest else 0
    elif k == "aer_reply":
        out["kind"] = abi.MSG_AER_REPLY
        out["term"] = m["term"]
        out["flags"] = abi.MF_SUCCESS if m["success"] else 0
        out["a"], out["b"], out["c"] = m["next_index"], m["last_index"], m["last_term"]
    elif k == "request_vote":
        out["kind"] = abi.MSG_REQUEST_VOTE
        out["term"] = m["term"]
        out["a"], out["b"] = m["last"]
    elif k == "vote_result":
        out["kind"] = abi.MSG_VOTE_RESULT
        out["term"] = m["term"]
        out["flags"] = abi.MF_SUCCESS if m["granted"] else 0
    elif k == "written":
        out["kind"] = abi.MSG_WRITTEN
        out["term"] = m["term"]
        out["a"], out["b"] = m["range"]
    elif k == "pipeline_rpcs":
        out["kind"] = abi.MSG_PIPELINE_RPCS
    elif k == "tick":                                  # leader tick_timeout -> make_rpcs/1
        out["kind"] = abi.MSG_PIPELINE_RPCS
        out["flags"] = abi.MF_TICK
    elif k == "append":
        out["kind"] = abi.MSG_APPEND
        out["n_entries"] = m["n"]
        out["flags"] = abi.MF_FORCE if m.get("force") else 0
    elif k == "heartbeat_rpc":
        out["kind"] = abi.MSG_HEARTBEAT_RPC
        out["term"] = m["term"]
        out["a"] = m["query_index"]
    elif k == "heartbeat_reply":
        out["kind"] = abi.MSG_HEARTBEAT_REPLY
        out["term"] = m["term"]
        out["a"] = m["query_index"]
    elif k == "consistent_query":
        out["kind"] = abi.MSG_CONSISTENT_QUERY
    elif k == "await_timeout":
        out["kind"] = abi.MSG_AWAIT_TIMEOUT
    elif k == "snapshot_written":
        out["kind"] = abi.MSG_SNAPSHOT_WRITTEN
        out["a"], out["b"] = m["index"], m["term"]
    elif k == "election_timeout":
        out["kind"] = abi.MSG_ELECTION_TIMEOUT
        out["c"] = m["token"]
    elif k == "pre_vote_rpc":
        out["kind"] = abi.MSG_PRE_VOTE_RPC
        out["term"] = m["term"]
        out["a"], out["b"] = m["last"]
        out["c"] = m["token"]
        out["n_entries"] = m["machine_version"]
        out["gap"] = m["version"]
    elif k == "pre_vote_result":
        out["kind"] = abi.MSG_PRE_VOTE_RESULT
        out["term"] = m["term"]
        out["flags"] = abi.MF_SUCCESS if m["granted"] else 0
        out["c"] = m["token"]
    else:
        raise ValueError(k)
    if m.get("can_write"):                            # ra_log:can_write/1 is true (wal_down_condition/2)
        out["flags"] |= abi.MF_CAN_WRITE
    return out


def _check_state(row, exp, where):
    for k, val in exp.items():
        if k in ("leader_id", "voted_for"):
            assert int(row[k]) == slot(val), f"{where}: {k}={int(row[k])} expected {val}"
        elif k == "last_written":
            got = [int(row["last_written_index"]), int(row["last_written_term"])]
            assert got == list(val), f"{where}: last_written={got} expected {val}"
        elif k == "snapshot":
            got = [int(row["snapshot_index"]), int(row["snapshot_term"])]
            assert got == list(val), f"{where}: snapshot={got} expected {val}"
        elif k == "log":
            got = [list(e) for e in abi.log_entries(row)]
            assert got == [list(e) for e in val], f"{where}: log={got} expected {val}"
        else:
            assert int(row[k]) == val, f"{where}: {k}={int(row[k])} expected {val}"


def states_equal(a, b) -> bool:
    return a.tobytes() == b.tobytes()


def canonical(row: np.ndarray) -> np.ndarray:
    """Zero the parts of a state row that carry no meaning (slots beyond n_members / n_runs)."""
    r = row.copy()
    n, k = int(r["n_members"]), int(r["n_runs"])
    for f in ("match_index", "next_index", "commit_index_sent"):
        r[f][n:] = 0
    r["run_start"][k:] = 0
    r["run_term"][k:] = 0
    r["_pad"][:] = 0
    return r


def run_vector(engine_factory, v):
    """engine_factory(n_groups, n_members) -> engine.  Raises AssertionError on any mismatch."""
    n = v["n_members"]
    eng = engine_factory(1, n)
    i = slot(v["self"])
    init = initial_state(v)
    eng.set_state(0, init)
    for sn, s in enumerate(v["steps"]):
        where = f"{v['id']} step {sn} ({v['source']})"
        if s.get("reset"):
            eng.set_state(0, init)
        before = eng.get_state(0, n)
        cur = before.copy()
        want_role = ROLE[s["as"]]
        patched = False
        if int(cur["role"][i]) != want_role:
            # the reference test calls handle_<as>/2 directly on this state
            cur["role"][i] = want_role
            patched = True
        if v.get("log_model") == "mem" and s["msg"].get("kind") == "written":
            # vectors authored against the reference's fake log (test/ra_log_memory.erl): it keeps no
            # `pending` seq, every written event applies -> nothing pending before the event
            empty = int(cur["last_index"][i]) + 1
            if int(cur["pending_first"][i]) != empty:
                cur["pending_first"][i] = empty
                patched = True
        if patched:
            eng.set_state(0, cur)
        dec, rpcs = eng.step(make_msg(v, s["msg"]))
        d = dec[0]
        after = eng.get_state(0, n)
        row = after[i]
        exp = s["expect"]
        flags = int(d["flags"])
        if "invariant" in exp:
            assert flags & abi.F_INVARIANT, f"{where}: expected the reference to exit"
            assert int(d["invariant"]) == exp["invariant"], f"{where}: invariant {int(d['invariant'])}"
            assert states_equal(canonical(cur[i]), canonical(row)), f"{where}: state changed by a crash"
            continue
        assert not (flags & abi.F_INVARIANT), f"{where}: invariant {int(d['invariant'])}"
        if "role" in exp:
            assert int(d["role"]) == ROLE[exp["role"]], \
                f"{where}: role={abi.ROLE_NAMES[int(d['role'])]} expected {exp['role']}"
            assert int(row["role"]) == ROLE[exp["role"]], f"{where}: state role"
        if "state" in exp:
            _check_state(row, exp["state"], where)
            if "commit_index" in exp["state"]:
                assert int(d["commit_index"]) == exp["state"]["commit_index"], where
            if "last_applied" in exp["state"]:
                assert int(d["last_applied"]) == exp["state"]["last_applied"], where
        for name, p in (exp.get("peers") or {}).items():
            for k, val in p.items():
                got = int(row[k][slot(name)])
                assert got == val, f"{where}: peer {name}.{k}={got} expected {val}"
        if exp.get("state_unchanged"):
            assert states_equal(canonical(cur[i]), canonical(row)), f"{where}: state changed"
        if exp.get("no_reply"):
            assert not (flags & abi.F_REPLY), f"{where}: unexpected reply"
        if "reply" in exp:
            r = exp["reply"]
            assert flags & abi.F_REPLY, f"{where}: no reply"
            assert bool(flags & abi.F_REPLY_VOTE) == bool(r.get("vote", False)), f"{where}: reply kind"
            assert bool(flags & abi.F_REPLY_PRE_VOTE) == bool(r.get("pre_vote", False)), f"{where}: reply kind"
            assert bool(flags & abi.F_REPLY_HEARTBEAT) == bool(r.get("heartbeat", False)), f"{where}: reply kind"
            if "query_index" in r:
                assert int(d["reply_next_index"]) == r["query_index"], f"{where}: reply query_index"
            if "token" in r:
                assert int(d["reply_next_index"]) == r["token"], f"{where}: reply token"
            if "to" in r:
                assert int(d["reply_to"]) == slot(r["to"]), f"{where}: reply_to={int(d['reply_to'])}"
            if "success" in r:
                assert bool(flags & abi.F_REPLY_SUCCESS) == r["success"], f"{where}: reply success"
            for k in ("term", "next_index", "last_index", "last_term"):
                if k in r:
                    got = int(d["reply_" + k])
                    assert got == r[k], f"{where}: reply.{k}={got} expected {r[k]}"
        if "vote_requests" in exp:
            q = exp["vote_requests"]
            assert flags & abi.F_SEND_VOTE_REQUESTS, f"{where}: no send_vote_requests"
            assert bool(flags & abi.F_PRE_VOTE_REQS) == q["pre_vote"], f"{where}: request kind"
            assert int(d["reply_term"]) == q["term"], f"{where}: request term {int(d['reply_term'])}"
            assert [int(d["reply_last_index"]), int(d["reply_last_term"])] == q["last"], f"{where}: last log"
            if "token" in q:
                assert int(d["reply_next_index"]) == q["token"], f"{where}: request token"
        if "heartbeats" in exp:
            hb = exp["heartbeats"]
            want = sum(1 << slot(x) for x in hb["to"])
            assert bool(flags & abi.F_SEND_HEARTBEATS) == bool(want), f"{where}: send heartbeats flag"
            assert int(d["heartbeat_to"]) == want, f"{where}: heartbeat_to={int(d['heartbeat_to']):#x}"
            assert int(d["reply_term"]) == hb["term"], f"{where}: heartbeat term"
            assert int(d["reply_last_term"]) == hb["query_index"], f"{where}: heartbeat query_index"
        if "cancel_backoff" in exp:
            want = sum(1 << slot(x) for x in exp["cancel_backoff"])
            assert bool(flags & abi.F_CANCEL_SNAPSHOT_RETRY) == bool(want), f"{where}: cancel_snapshot_retry_timer flag"
            assert int(d["cancel_backoff"]) == want, f"{where}: cancel_backoff={int(d['cancel_backoff']):#x}"
        if "query_quorum" in exp:
            assert flags & abi.F_QUERY_QUORUM, f"{where}: no query quorum"
            assert int(d["reply_next_index"]) == exp["query_quorum"], \
                f"{where}: consensus query index {int(d['reply_next_index'])}"
        if exp.get("effects_only_reply"):
            other = flags & ~(abi.F_REPLY | abi.F_REPLY_SUCCESS | abi.F_REPLY_VOTE | abi.F_REPLY_PRE_VOTE |
                              abi.F_REPLY_HEARTBEAT | abi.F_PERSIST |
                              abi.F_LEADER_CHANGED | abi.F_ROLE_CHANGED | abi.F_REPROCESSED)
            assert other == 0, f"{where}: extra effects flags {other:#x}"
        for fn in exp.get("flags_set", []):
            assert flags & FLAG[fn], f"{where}: flag {fn} not set ({flags:#x})"
        for fn in exp.get("flags_clear", []):
            assert not (flags & FLAG[fn]), f"{where}: flag {fn} set ({flags:#x})"
        if "rpcs" in exp:
            got = {int(r["peer"]): r for r in rpcs}
            assert int(d["n_rpcs"]) == len(rpcs), f"{where}: n_rpcs"
            if exp.get("rpcs_exact"):
                assert sorted(got) == sorted(slot(e["peer"]) for e in exp["rpcs"]), \
                    f"{where}: rpc peers {sorted(got)}"
            for e in exp["rpcs"]:
                r = got.get(slot(e["peer"]))
                assert r is not None, f"{where}: no rpc for {e['peer']}"
                assert int(r["kind"]) == abi.RPC_AER, where
                if "term" in e:
                    assert int(r["term"]) == e["term"], where
                if "prev" in e:
                    assert [int(r["prev_log_index"]), int(r["prev_log_term"])] == e["prev"], \
                        f"{where}: rpc prev {int(r['prev_log_index'])}:{int(r['prev_log_term'])}"
                if "commit" in e:
                    assert int(r["leader_commit"]) == e["commit"], where
                if "entries" in e:
                    lo = int(r["prev_log_index"]) + 1
                    hi = int(r["prev_log_index"]) + int(r["n_entries"])
                    assert [lo, hi] == e["entries"], f"{where}: rpc entries {lo}..{hi}"
        if s.get("fork"):
            eng.set_state(0, before)
    if hasattr(eng, "close"):
        eng.close()
