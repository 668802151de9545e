"""Third restatements, against the checker on random states, of
 * handle_follower(#heartbeat_rpc{}) (src/ra_server.erl:1441-1456),
 * handle_follower({ra_log_event, {written, _, _}}) (:1457-1474): the reply goes out only when last_written
   moved and a leader is known,
 * handle_leader({ra_log_event, {written, _, _}}) (:739-744): the log event, then evaluate_quorum with the
   leader's own last_written in the median, then {next_event, info, pipeline_rpcs}.
The log side is tests/ra_log_model.py (pending as a real ra_seq), seeded from the state row."""
import numpy as np
import pytest

from ra_amd import abi
import fuzz
from ra_log_model import LogModel


def seed_log_model(row) -> LogModel:
    m = LogModel()
    ents = abi.log_entries(row)
    m.terms = dict(ents)
    m.range = (int(row["first_index"]), int(row["last_index"])) if ents else None
    m.last_term = int(row["last_term"])
    m.lw = (int(row["last_written_index"]), int(row["last_written_term"]))
    si = int(row["snapshot_index"])
    m.snap = None if si == abi.UNDEF_INT else (si, int(row["snapshot_term"]))
    pf, li = int(row["pending_first"]), int(row["last_index"])
    m.pending = [(pf, li)] if pf < li else ([pf] if pf == li else [])
    return m


def _msgs(rows, kind_fn):
    out = []
    for sv, r in enumerate(rows):
        m = np.zeros(1, dtype=abi.MSG_DTYPE)
        m["server"] = sv
        kind_fn(m, r)
        out.append(m[0])
    return np.array(out, dtype=abi.MSG_DTYPE)


@pytest.mark.parametrize("n,seed", [(3, 1), (5, 2)])
def test_follower_heartbeat_rpc(oracle_lib, n, seed):
    rng = np.random.default_rng(7000 + seed)
    G = 150
    st = fuzz.random_states(rng, G, n, max_runs=6)
    st["role"] = abi.ROLE_FOLLOWER
    st["cond_reason"] = 0
    cpu = oracle_lib.Oracle(G, n); cpu.set_state(0, st)
    before = cpu.get_state()

    def hb(m, r):
        m["kind"] = abi.MSG_HEARTBEAT_RPC
        m["term"] = max(0, int(r["current_term"]) + int(rng.choice([-1, 0, 0, 1, 2])))
        m["from"] = int(rng.choice([i for i in range(n) if i != int(r["self"])]))
        m["a"] = int(rng.integers(0, 50))
    msgs = _msgs(before, hb)
    dec, _ = cpu.step(msgs)
    after = cpu.get_state()
    older = newer = 0
    for m, d, r0, r1 in zip(msgs, dec, before, after):
        cur, term, frm, fl = int(r0["current_term"]), int(m["term"]), int(m["from"]), int(d["flags"])
        tag = f"server {int(m['server'])} cur {cur} msg {m}"
        assert fl & abi.F_REPLY and fl & abi.F_REPLY_HEARTBEAT and int(d["reply_to"]) == frm, tag
        assert int(d["reply_next_index"]) == int(m["a"]), tag
        assert int(r1["role"]) == abi.ROLE_FOLLOWER, tag
        if term >= cur:                                               # :1441-1450
            assert int(d["reply_term"]) == term == int(r1["current_term"]), tag
            assert int(r1["leader_id"]) == frm, tag
            assert int(r1["voted_for"]) == (abi.NONE if term > cur else int(r0["voted_for"])), tag
            assert bool(fl & abi.F_PERSIST) == (term > cur), tag
            newer += term > cur
        else:                                                         # :1451-1456
            assert int(d["reply_term"]) == cur and r1.tobytes() == r0.tobytes(), tag
            older += 1
        for f in ("commit_index", "last_applied", "last_index", "last_written_index", "pending_first"):
            assert int(r0[f]) == int(r1[f]), (tag, f)
    assert older > 10 and newer > 10


def written_msg(rng, m, r):
    """A written event around the server's pending range: exact, partial, stale term, already confirmed."""
    m["kind"] = abi.MSG_WRITTEN
    pf, li, lt = int(r["pending_first"]), int(r["last_index"]), int(r["last_term"])
    x = rng.random()
    if pf <= li and x < 0.6:
        lo, hi = pf, int(rng.integers(pf, li + 1))                    # a prefix of what is pending
    elif pf <= li and x < 0.75:
        lo = int(rng.integers(pf, li + 1)); hi = int(rng.integers(lo, li + 1))   # maybe not a prefix
    else:
        hi = max(1, int(rng.integers(max(1, int(r["first_index"])), li + 2)))
        lo = max(1, hi - int(rng.integers(0, 3)))
    ents = dict(abi.log_entries(r))
    t = ents.get(hi, lt)
    m["term"] = t if rng.random() < 0.8 else t + 1
    m["a"], m["b"] = lo, hi


def apply_written(model: LogModel, m):
    lo, hi = int(m["a"]), int(m["b"])
    model.written(int(m["term"]), [(lo, hi)] if lo < hi else [hi])


@pytest.mark.parametrize("n,seed", [(3, 3), (5, 4)])
def test_follower_written_event(oracle_lib, n, seed):
    rng = np.random.default_rng(7000 + seed)
    G = 200
    st = fuzz.random_states(rng, G, n, max_runs=6)
    st["role"] = abi.ROLE_FOLLOWER
    st["cond_reason"] = 0
    unknown = rng.random(len(st)) < 0.2
    st["leader_id"][unknown] = abi.NONE
    cpu = oracle_lib.Oracle(G, n); cpu.set_state(0, st)
    before = cpu.get_state()
    msgs = _msgs(before, lambda m, r: written_msg(rng, m, r))
    dec, _ = cpu.step(msgs)
    after = cpu.get_state()
    seen = {"moved_reply": 0, "moved_no_leader": 0, "unchanged": 0, "resend": 0, "crash": 0}
    for m, d, r0, r1 in zip(msgs, dec, before, after):
        tag = f"server {int(m['server'])} msg {m}"
        fl = int(d["flags"])
        model = seed_log_model(r0)
        lw0 = model.lw
        try:
            apply_written(model, m)
        except AssertionError:                                        # {ok, Pend} = ra_seq:remove_prefix(..) badmatch
            assert fl & abi.F_INVARIANT and int(d["invariant"]) == abi.INV_WRITTEN_NOT_PREFIX, tag
            assert r1.tobytes() == r0.tobytes(), tag
            seen["crash"] += 1
            continue
        assert not fl & abi.F_INVARIANT, (tag, int(d["invariant"]))
        assert (int(r1["last_written_index"]), int(r1["last_written_term"])) == model.lw, tag
        pend = model.pending
        want_pf = (pend[-1][0] if isinstance(pend[-1], tuple) else pend[-1]) if pend else int(r1["last_index"]) + 1
        assert int(r1["pending_first"]) == want_pf, (tag, pend)
        assert bool(fl & abi.F_RESEND_PENDING) == model.resend, tag
        seen["resend"] += model.resend
        moved = model.lw != lw0
        leader = int(r0["leader_id"])
        if moved and leader != abi.NONE:                              # :1466-1470
            assert fl & abi.F_REPLY and fl & abi.F_REPLY_SUCCESS and int(d["reply_to"]) == leader, tag
            assert (int(d["reply_term"]), int(d["reply_next_index"]), int(d["reply_last_index"]),
                    int(d["reply_last_term"])) == (int(r0["current_term"]), int(r0["last_index"]) + 1, *model.lw), tag
            seen["moved_reply"] += 1
        else:
            assert not fl & abi.F_REPLY, tag
            seen["moved_no_leader" if moved else "unchanged"] += 1
        for f in ("current_term", "commit_index", "last_applied", "last_index", "role"):
            assert int(r0[f]) == int(r1[f]), (tag, f)
    assert seen["moved_reply"] > 30 and seen["moved_no_leader"] > 3 and seen["unchanged"] > 10, seen


@pytest.mark.parametrize("n,seed", [(3, 5), (5, 6), (7, 7)])
def test_leader_written_event(oracle_lib, n, seed):
    rng = np.random.default_rng(7000 + seed)
    G = 200
    st = fuzz.random_states(rng, G, n, max_runs=6)
    st["role"] = abi.ROLE_LEADER
    st["cond_reason"] = 0
    st["leader_id"] = st["self"]
    cpu = oracle_lib.Oracle(G, n); cpu.set_state(0, st)
    before = cpu.get_state()
    msgs = _msgs(before, lambda m, r: written_msg(rng, m, r))
    dec, _ = cpu.step(msgs)
    after = cpu.get_state()
    advanced = same = lower = 0
    for m, d, r0, r1 in zip(msgs, dec, before, after):
        tag = f"server {int(m['server'])} msg {m}"
        fl = int(d["flags"])
        model = seed_log_model(r0)
        try:
            apply_written(model, m)
        except AssertionError:
            assert fl & abi.F_INVARIANT and r1.tobytes() == r0.tobytes(), tag
            continue
        if fl & abi.F_INVARIANT:
            continue                                                  # a later step of the clause (pipelining) asserted
        assert (int(r1["last_written_index"]), int(r1["last_written_term"])) == model.lw, tag
        # evaluate_quorum/2 :3633-3657 with match_indexes/1 :3671-3682 and agreed_commit/1 :3684-3688
        me = int(r0["self"])
        idxs = [model.lw[0]]
        for p in range(n):
            if p == me or not (int(r0["present_mask"]) >> p) & 1 or not (int(r0["voter_mask"]) >> p) & 1:
                continue
            idxs.append(int(r0["match_index"][p]))
        idxs.sort(reverse=True)
        agreed = idxs[len(idxs) // 2]
        ci0 = int(r0["commit_index"])
        term_at = dict(abi.log_entries(r1)).get(agreed)
        if term_at is None and model.snap and model.snap[0] == agreed:
            term_at = model.snap[1]
        # increment_commit_index/1 :3648-3657: no max() -- the potential index is taken whenever its term is
        # the current term (5.4.2), even if it is below the old commit index
        want_ci = agreed if term_at == int(r0["current_term"]) else ci0
        assert int(r1["commit_index"]) == want_ci, (tag, idxs, agreed, term_at)
        assert fl & abi.F_PIPELINE, tag                              # [{next_event, info, pipeline_rpcs} | _]
        assert int(r1["role"]) == abi.ROLE_LEADER and int(r1["current_term"]) == int(r0["current_term"]), tag
        advanced += want_ci > ci0
        same += want_ci == ci0
        lower += want_ci < ci0
    assert advanced > 10 and same > 10, (advanced, same, lower)
