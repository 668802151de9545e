"""Random-but-structurally-valid server states and messages for differential testing of the HIP
path against the CPU checker.  Values are drawn close to each other (indexes near last_index,
terms near current_term) so that every clause of the transition is reached often: log-matching
ok / missing / mismatch, drop_existing overlap, truncation, overwrite, quorum with term gate,
all a8 repair branches, vote clause order, role demotion + re-processing, await_condition
predicate, pipelining with the in-flight clamp, invariant breaches."""
from __future__ import annotations

import numpy as np

from ra_amd import abi


def random_states(rng: np.random.Generator, n_groups: int, n_members: int, max_runs: int = 8,
                  backlog: int = 24) -> np.ndarray:
    S = n_groups * n_members
    st = abi.empty_server_states(n_groups, n_members)
    for s in range(S):
        has_snap = rng.random() < 0.4
        if has_snap:
            si = int(rng.integers(0, 50))
            stm = int(rng.integers(0, 4))
            first = si + 1
            base_term = stm
        else:
            si, stm = None, None
            first = 0
            base_term = 0
        empty_range = has_snap and rng.random() < 0.15
        if empty_range:
            abi.set_log(st, s, [], snapshot=(si, stm))
        else:
            n_runs = int(rng.integers(1, max_runs - 2))
            entries = []
            idx = first
            term = base_term if first > 0 else 0
            for r in range(n_runs):
                ln = int(rng.integers(1, max(2, backlog // n_runs)))
                if r > 0 or first > 0:
                    term += int(rng.integers(1, 3)) if r > 0 else int(rng.integers(0, 2))
                for _ in range(ln):
                    entries.append((idx, term))
                    idx += 1
            li = entries[-1][0]
            lwi = int(rng.integers(max(first - 1, 0), li + 1)) if rng.random() < 0.7 else li
            if lwi >= first:
                lwt = dict(entries)[lwi]
            else:
                lwt = stm if has_snap else 0
            abi.set_log(st, s, entries, last_written=(lwi, lwt),
                        snapshot=(si, stm) if has_snap else None)
        li, lt = int(st["last_index"][s]), int(st["last_term"][s])
        first = int(st["first_index"][s])
        if first <= li and rng.random() < 0.4:
            # `pending` need not start right after last_written (resends, segment flushes)
            st["pending_first"][s] = int(rng.integers(first, li + 2))
        st["current_term"][s] = lt + int(rng.integers(0, 3))
        lo = max(first - 1, 0) if first > 0 else 0
        la = int(rng.integers(lo, li + 1))
        st["last_applied"][s] = la
        st["commit_index"][s] = la + int(rng.integers(0, 3))
        role = rng.choice([abi.ROLE_FOLLOWER, abi.ROLE_LEADER, abi.ROLE_CANDIDATE, abi.ROLE_PRE_VOTE,
                           abi.ROLE_AWAIT_CONDITION], p=[0.4, 0.35, 0.08, 0.05, 0.12])
        st["role"][s] = role
        if role == abi.ROLE_AWAIT_CONDITION:
            st["cond_reason"][s] = rng.choice([abi.COND_MISSING, abi.COND_TERM_MISMATCH])
            st["cond_reply"][s] = rng.integers(0, 60, size=4)
            st["cond_leader"][s] = int(rng.integers(0, n_members))
        st["voted_for"][s] = abi.NONE if rng.random() < 0.5 else int(rng.integers(0, n_members))
        st["leader_id"][s] = abi.NONE if rng.random() < 0.3 else int(rng.integers(0, n_members))
        st["votes"][s] = int(rng.integers(0, n_members))
        full = (1 << n_members) - 1
        st["present_mask"][s] = full if rng.random() < 0.85 else (int(rng.integers(0, full + 1)) | (1 << (s % n_members)))
        st["voter_mask"][s] = full if rng.random() < 0.85 else (int(rng.integers(0, full + 1)) | (1 << (s % n_members)))
        st["status_mask"][s] = 0xFF if rng.random() < 0.85 else int(rng.integers(0, 256))
        # some of the peers that are not normal are in {snapshot_backoff, _} (never self, only members)
        if int(st["status_mask"][s]) != 0xFF and rng.random() < 0.6:
            st["backoff_mask"][s] = (int(rng.integers(0, 256)) & ~int(st["status_mask"][s]) &
                                     int(st["present_mask"][s]) & ~(1 << (s % n_members)) & 0xFF)
        st["self_nonvoter"][s] = 1 if rng.random() < 0.05 else 0
        st["pre_vote_token"][s] = int(rng.integers(0, 3))
        if rng.random() < 0.35:
            st["query_index"][s] = int(rng.integers(0, 6))
        if rng.random() < 0.3:
            st["peer_query_index"][s, :n_members] = rng.integers(0, 6, size=n_members)
        st["machine_version"][s] = int(rng.integers(0, 3))
        st["effective_machine_version"][s] = int(rng.integers(0, 3))
        for j in range(n_members):
            mi = max(0, li - int(rng.integers(0, 12)))
            if rng.random() < 0.1:
                mi = 0
            ni = mi + 1 + int(rng.integers(0, 6))
            if rng.random() < 0.1:
                ni = max(0, mi - int(rng.integers(0, 3)))       # next_index <= match_index
            if rng.random() < 0.05:
                ni = li + 1 + int(rng.integers(0, 3))           # may point past the log
            st["match_index"][s, j] = mi
            st["next_index"][s, j] = ni
            st["commit_index_sent"][s, j] = max(0, int(st["commit_index"][s]) - int(rng.integers(0, 3)))
    return st


def _term_at(row, idx):
    for i, t in abi.log_entries(row):
        if i == idx:
            return t
    if int(row["snapshot_index"]) == idx:
        return int(row["snapshot_term"])
    return None


def random_msgs(rng: np.random.Generator, st: np.ndarray, n_members: int, frac: float = 0.9) -> np.ndarray:
    """At most one message per server (a tick), for a random `frac` of the servers, shuffled."""
    S = len(st)
    targets = np.flatnonzero(rng.random(S) < frac)
    rng.shuffle(targets)
    m = np.zeros(len(targets), dtype=abi.MSG_DTYPE)
    for q, s in enumerate(targets):
        row = st[s]
        li, lt = int(row["last_index"]), int(row["last_term"])
        ct = int(row["current_term"])
        first = int(row["first_index"])
        self_ = int(row["self"])
        m["server"][q] = s
        frm = int(rng.integers(0, n_members))
        if rng.random() < 0.03:
            frm = 7 if n_members < 8 else 0
        m["from"][q] = frm
        role = int(row["role"])
        kinds = [abi.MSG_AER, abi.MSG_AER_REPLY, abi.MSG_REQUEST_VOTE, abi.MSG_VOTE_RESULT,
                 abi.MSG_WRITTEN, abi.MSG_PIPELINE_RPCS, abi.MSG_APPEND, abi.MSG_AWAIT_TIMEOUT,
                 abi.MSG_NOP, abi.MSG_ELECTION_TIMEOUT, abi.MSG_PRE_VOTE_RPC, abi.MSG_PRE_VOTE_RESULT,
                 abi.MSG_SNAPSHOT_WRITTEN, abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY,
                 abi.MSG_CONSISTENT_QUERY]
        if role == abi.ROLE_LEADER:
            p = [0.082, 0.3116, 0.0656, 0.0164, 0.082, 0.082, 0.0984, 0.0082, 0.0164, 0.0082, 0.0164, 0.0082, 0.0246, 0.02, 0.1, 0.06]
        elif role == abi.ROLE_CANDIDATE:
            p = [0.1602, 0.0712, 0.1157, 0.2937, 0.089, 0.0089, 0.0089, 0.0089, 0.0178, 0.0356, 0.0356, 0.0178, 0.0267, 0.06, 0.04, 0.01]
        elif role == abi.ROLE_AWAIT_CONDITION:
            p = [0.4371, 0.0465, 0.093, 0.0186, 0.093, 0.0093, 0.0093, 0.1116, 0.0186, 0.0279, 0.0279, 0.0093, 0.0279, 0.04, 0.02, 0.01]
        elif role == abi.ROLE_PRE_VOTE:
            p = [0.2225, 0.0356, 0.089, 0.0178, 0.089, 0.0089, 0.0089, 0.0089, 0.0178, 0.0534, 0.0712, 0.2403, 0.0267, 0.06, 0.04, 0.01]
        else:
            p = [0.3696, 0.0528, 0.1584, 0.0264, 0.1144, 0.0088, 0.0176, 0.0088, 0.0176, 0.0352, 0.0264, 0.0088, 0.0352, 0.08, 0.03, 0.01]
        kind = int(rng.choice(kinds, p=p))
        m["kind"][q] = kind
        term = ct + int(rng.choice([-1, 0, 0, 0, 0, 1, 2], p=[0.1, 0.2, 0.2, 0.2, 0.1, 0.15, 0.05]))
        m["term"][q] = max(term, 0)
        if kind == abi.MSG_AER:
            mode = rng.random()
            if mode < 0.45:
                prev = li
            elif mode < 0.75:
                prev = max(li - int(rng.integers(1, 6)), 0)
            elif mode < 0.9:
                prev = li + int(rng.integers(1, 4))
            else:
                prev = max(first - 1, 0)
            pt = _term_at(row, prev)
            if pt is None or rng.random() < 0.15:
                pt = lt + int(rng.integers(-1, 2))
            m["a"][q] = prev
            m["b"][q] = max(pt, 0)
            m["c"][q] = max(0, int(row["commit_index"]) + int(rng.integers(-2, 5)))
            n_ent = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
            gap = 0 if rng.random() < 0.95 else int(rng.integers(1, 3))
            m["gap"][q] = gap
            m["n_entries"][q] = n_ent
            if n_ent:
                # entry terms: mostly copy what the local log has (overlap), then switch to a
                # newer term (append / overwrite); at most two runs
                base = prev + 1 + gap
                t0 = _term_at(row, base)
                if t0 is None or rng.random() < 0.3:
                    t0 = max(int(m["b"][q]), lt) + int(rng.integers(0, 2))
                n0 = int(rng.integers(1, n_ent + 1))
                t1 = t0 + int(rng.integers(0, 3))
                m["n_run0"][q] = n0
                m["run0_term"][q] = t0
                m["run1_term"][q] = t1
        elif kind == abi.MSG_AER_REPLY:
            ok = rng.random() < 0.6
            m["flags"][q] = abi.MF_SUCCESS if ok else 0
            mi = int(row["match_index"][frm]) if frm < n_members else 0
            last = max(0, mi + int(rng.integers(-4, 6)))
            if rng.random() < 0.2:
                last = max(0, li - int(rng.integers(0, 4)))
            m["b"][q] = last
            m["a"][q] = last + 1 + int(rng.integers(0, 3))
            lt_ = _term_at(row, last)
            if lt_ is None or rng.random() < 0.35:
                lt_ = max(0, lt + int(rng.integers(-2, 2)))
            m["c"][q] = lt_
        elif kind == abi.MSG_REQUEST_VOTE:
            m["a"][q] = max(0, li + int(rng.integers(-2, 3)))
            m["b"][q] = max(0, lt + int(rng.integers(-1, 2)))
        elif kind == abi.MSG_VOTE_RESULT:
            m["flags"][q] = abi.MF_SUCCESS if rng.random() < 0.7 else 0
        elif kind == abi.MSG_WRITTEN:
            hi = max(0, li + int(rng.integers(-3, 3)))
            lo = max(0, hi - int(rng.integers(0, 6)))
            if rng.random() < 0.5:
                lo = min(int(row["pending_first"]), hi)          # the in-order case: a prefix of pending
            m["a"][q], m["b"][q] = lo, hi
            tw = _term_at(row, hi)
            if tw is None or rng.random() < 0.3:
                tw = max(0, lt + int(rng.integers(-2, 1)))
            m["term"][q] = tw
        elif kind == abi.MSG_SNAPSHOT_WRITTEN:
            # around last_applied / the range start / past the end of the log
            si = max(0, int(row["last_applied"]) + int(rng.integers(-3, 3)))
            if rng.random() < 0.15:
                si = li + int(rng.integers(0, 3))
            if rng.random() < 0.1:
                si = max(0, first - int(rng.integers(0, 3)))
            m["a"][q] = si
            tt = _term_at(row, si)
            m["b"][q] = tt if tt is not None else max(0, lt - int(rng.integers(0, 2)))
        elif kind in (abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY):
            m["a"][q] = int(rng.integers(0, 8))
            if rng.random() < 0.1:
                m["from"][q] = abi.NONE
        elif kind == abi.MSG_ELECTION_TIMEOUT:
            m["c"][q] = int(rng.integers(0, 3))
        elif kind == abi.MSG_PRE_VOTE_RPC:
            m["a"][q] = max(0, li + int(rng.integers(-2, 3)))
            m["b"][q] = max(0, lt + int(rng.integers(-1, 2)))
            m["c"][q] = int(rng.integers(0, 1000))
            m["n_entries"][q] = int(rng.integers(0, 4))
            m["gap"][q] = int(rng.choice([0, 1, 1, 1, 2]))
        elif kind == abi.MSG_PRE_VOTE_RESULT:
            m["flags"][q] = abi.MF_SUCCESS if rng.random() < 0.75 else 0
            m["c"][q] = int(rng.integers(0, 3))
        elif kind == abi.MSG_APPEND:
            m["n_entries"][q] = int(rng.integers(0, 5))
            m["flags"][q] = abi.MF_FORCE if rng.random() < 0.2 else 0
        elif kind == abi.MSG_PIPELINE_RPCS:
            m["flags"][q] = abi.MF_TICK if rng.random() < 0.4 else 0      # tick_timeout: make_rpcs/1
        _ = self_
    return m


def sort_rpcs(r: np.ndarray) -> np.ndarray:
    return r[np.lexsort((r["peer"], r["msg_index"]))]
