"""handle_follower(#append_entries_rpc{}) (src/ra_server.erl:1283-1440) restated clause by clause on
top of tests/ra_log_model.py (entry lists, ra_log:exists/2 per entry in drop_existing/3, a real
ra_seq for pending) against the checker, on random follower histories: tail appends, resends that
overlap the log (same and different terms), empty rpcs that truncate, gaps (missing), wrong
prev_log_term (term_mismatch), stale terms, commit indexes ahead of the log, written events."""
import numpy as np
import pytest

from ra_amd import abi
from ra_log_model import LogModel


class Follower:
    def __init__(self):
        self.log = LogModel()
        self.ct, self.voted_for, self.leader = 0, None, None
        self.ci = self.la = 0

    # ra_server:fetch_term/2 with the snapshot fallback (:3185-3196)
    def term_of(self, idx):
        t = self.log.fetch_term(idx)
        if t is None and self.log.snap and self.log.snap[0] == idx:
            return self.log.snap[1]
        return t

    def has_log_entry_or_snapshot(self, idx, term):            # :3168-3183
        t = self.log.fetch_term(idx)
        if t is None:
            if self.log.snap and self.log.snap[0] == idx:
                return "entry_ok" if self.log.snap[1] == term else "term_mismatch"
            return "missing"
        return "entry_ok" if t == term else "term_mismatch"

    def reply(self, term, success):                            # append_entries_reply/3 :3624-3631
        li, _ = self.log.last_index_term()
        return (term, success, li + 1, self.log.lw[0], self.log.lw[1])

    def evaluate_commit_index_follower(self):                  # :2246-2280 + apply_to
        if self.leader is None:
            return
        li, _ = self.log.last_index_term()
        apply_to = min(li, self.ci)
        if apply_to > self.la:
            self.la = apply_to

    def aer(self, term, leader, commit, pli, plt, entries):
        """-> (next role, reply tuple or None)"""
        cur_term, last_applied = self.ct, self.la
        if not term >= cur_term:
            return "follower", self.reply(cur_term, False)     # :1431-1440
        if term > self.ct:                                     # update_term/2
            self.ct, self.voted_for = term, None
        self.leader = leader
        h = self.has_log_entry_or_snapshot(pli, plt)
        if h == "entry_ok":
            rest, last_valid = list(entries), pli              # drop_existing/3 :3700-3708
            while rest and self.log.fetch_term(rest[0][0]) == rest[0][1]:
                last_valid = rest[0][0]
                rest.pop(0)
            if not rest:
                local_last, _ = self.log.last_index_term()
                if not entries and local_last > pli:
                    assert not (pli < last_applied)            # ?assertNot(PLIdx < LastApplied)
                    assert self.log.set_last_index(pli)
                    validated = True
                else:
                    validated = local_last <= last_valid
                if validated:
                    self.ci = commit                           # not clamped
                    self.evaluate_commit_index_follower()
                    return "follower", self.reply(term, True)
                v = max(last_applied, last_valid)
                return "follower", (cur_term, True, v + 1, v, self.term_of(v))   # pre-update CurTerm
            self.ci = commit
            assert not (rest[0][0] < last_applied)             # ?assertNot(FstIdx < LastApplied)
            self.log.write(rest)
            self.evaluate_commit_index_follower()
            return "follower", None                            # the reply comes with the written event
        if h == "missing":
            return "await_condition", self.reply(term, False)
        lat = self.term_of(last_applied)                       # mismatch_append_entries_reply/3
        assert lat is not None
        return "await_condition", (term, False, last_applied + 1, last_applied, lat)


def _msg(kind, **kw):
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"] = 1
    m["kind"] = kind
    m["from"] = kw.pop("frm", abi.NONE)
    for k, v in kw.items():
        m[k] = v
    return m


def run_history(oracle_lib, seed):
    rng = np.random.default_rng(4000 + seed)
    cpu = oracle_lib.Oracle(1, 3)
    f = Follower()
    leader_term = 1
    seen = set()
    for step in range(250):
        st = cpu.get_state()[1]
        if int(st["role"]) != abi.ROLE_FOLLOWER:               # the model covers the follower role
            patch = cpu.get_state()
            patch["role"][1] = abi.ROLE_FOLLOWER
            patch["cond_reason"][1] = abi.COND_NONE
            patch["cond_reply"][1] = 0
            patch["cond_leader"][1] = abi.NONE
            cpu.set_state(0, patch)
        li, lt = f.log.last_index_term()
        r = rng.random()
        if r < 0.08:                                           # a snapshot at last_applied keeps the term structure short
            t = f.log.fetch_term(f.la)
            if f.la == 0 or t is None:
                continue
            cpu.step(_msg(abi.MSG_SNAPSHOT_WRITTEN, a=f.la, b=t))
            f.log.snapshot_written(f.la, t)
        elif r < 0.30:                                         # a written event for the unwritten tail
            a, b = f.log.lw[0] + 1, li
            if a > b:
                continue
            d, _ = cpu.step(_msg(abi.MSG_WRITTEN, term=f.log.fetch_term(b) or lt, a=a, b=b))
            f.log.written(f.log.fetch_term(b) or lt, [(a, b)] if b > a else [a])
            got_reply = bool(int(d["flags"][0]) & abi.F_REPLY)
        else:
            if rng.random() < 0.05:
                leader_term += 1
            term = leader_term if rng.random() < 0.9 else max(0, f.ct - 1)
            mode = rng.random()
            if mode < 0.45:
                pli, plt = li, lt                              # at the tail
            elif mode < 0.70:
                pli = max(f.la, li - int(rng.integers(1, 5)))  # behind the tail (resend / truncate)
                plt = f.term_of(pli)
                if plt is None:
                    continue
            elif mode < 0.80:
                pli, plt = li + int(rng.integers(1, 3)), lt    # a gap: missing
            else:
                pli = max(0, li - int(rng.integers(0, 3)))     # wrong prev_log_term
                plt = (f.term_of(pli) or 0) + 1
            n = int(rng.integers(0, 5))
            # entries: a resend of what is there (same terms) and/or new entries in the leader's term
            ents = []
            for k in range(n):
                idx = pli + 1 + k
                have = f.log.fetch_term(idx)
                ents.append((idx, have if (have is not None and rng.random() < 0.6 and
                                           (not ents or ents[-1][1] <= have)) else max(term, ents[-1][1] if ents else 0)))
            terms = [t for _, t in ents]
            runs = sorted(set(terms), key=terms.index)
            if len(runs) > 2 or terms != sorted(terms):
                continue                                       # the message format carries <= 2 ascending term runs
            n0 = terms.count(runs[0]) if runs else 0
            commit = int(rng.integers(0, li + 3))
            d, _ = cpu.step(_msg(abi.MSG_AER, frm=0, term=term, a=pli, b=plt, c=commit, n_entries=n, n_run0=n0,
                                 run0_term=runs[0] if runs else 0, run1_term=runs[1] if len(runs) > 1 else 0))
            flags = int(d["flags"][0])
            try:
                role, reply = f.aer(term, 0, commit, pli, plt, ents)
            except AssertionError:
                assert flags & abi.F_INVARIANT, f"step {step}: the reference would have crashed"
                seen.add("crash")
                break
            assert not (flags & abi.F_INVARIANT), f"step {step}: invariant {int(d['invariant'][0])}"
            assert abi.ROLE_NAMES[int(d["role"][0])] == role, f"step {step}"
            seen.add(role if reply is None or reply[1] else role + "/fail")
            if reply is None:
                assert not (flags & abi.F_REPLY), f"step {step}"
            else:
                assert flags & abi.F_REPLY, f"step {step}"
                got = (int(d["reply_term"][0]), bool(flags & abi.F_REPLY_SUCCESS), int(d["reply_next_index"][0]),
                       int(d["reply_last_index"][0]), int(d["reply_last_term"][0]))
                assert got == reply, f"step {step}: reply {got} model {reply}"
        st = cpu.get_state()[1]
        assert (int(st["current_term"]), int(st["commit_index"]), int(st["last_applied"])) == (f.ct, f.ci, f.la), f"step {step}"
        assert (int(st["last_index"]), int(st["last_term"])) == f.log.last_index_term(), f"step {step}"
        assert (int(st["last_written_index"]), int(st["last_written_term"])) == f.log.lw, f"step {step}"
        assert [tuple(e) for e in abi.log_entries(st)] == sorted(f.log.terms.items()), f"step {step}"
    cpu.close()
    return seen


@pytest.mark.parametrize("seed", list(range(20)))
def test_follower_aer_matches_clause_model(oracle_lib, seed):
    run_history(oracle_lib, seed)


def test_follower_aer_model_reaches_every_outcome(oracle_lib):
    seen = set()
    for seed in range(20, 32):
        seen |= run_history(oracle_lib, seed)
    assert {"follower", "follower/fail", "await_condition/fail"} <= seen, seen
