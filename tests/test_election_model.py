"""Third restatement of the election family, clause by clause in the reference's order, against the
checker: handle_candidate/2 (src/ra_server.erl:1045-1186), handle_pre_vote/2 (:1188-1278),
call_for_election/2 (:2877-2924), process_pre_vote/3 (:2926-2983), update_term/2 and
update_term_and_voted_for/3 (:3041-3066), plus the follower's election_timeout and pre_vote_rpc
clauses (:1475-1482, :1618-1626).

Where the reference answers with {next_event, Msg} (step down, then handle Msg again as a follower)
the engine does both halves in one decision.  The model restates the first half; the second half is
checked by consistency: the one-step result must equal feeding the same message to a server that
already is in the modelled intermediate state."""
import numpy as np
import pytest

from ra_amd import abi
import fuzz

ELECTION_FLAGS = (abi.F_REPLY | abi.F_REPLY_SUCCESS | abi.F_REPLY_VOTE | abi.F_REPLY_PRE_VOTE | abi.F_REPLY_HEARTBEAT |
                  abi.F_START_ELECTION_TIMEOUT | abi.F_SEND_VOTE_REQUESTS | abi.F_PRE_VOTE_REQS | abi.F_BECAME_LEADER |
                  abi.F_PERSIST | abi.F_UNHANDLED)


class Srv:
    """The integer part of ra_server_state() the election clauses touch."""

    def __init__(self, row):
        self.row = row
        self.role = int(row["role"])
        self.term = int(row["current_term"])
        self.voted_for = int(row["voted_for"])
        self.votes = int(row["votes"])
        self.leader_id = int(row["leader_id"])
        self.token = int(row["pre_vote_token"])
        self.me = int(row["self"])
        self.voter = not int(row["self_nonvoter"])
        present, voters = int(row["present_mask"]), int(row["voter_mask"])
        self.quorum = bin(present & voters).count("1") // 2 + 1          # required_quorum/1 :3996-3999
        self.last = (int(row["last_index"]), int(row["last_term"]))      # last_idx_term/1
        self.last_written = (int(row["last_written_index"]), int(row["last_written_term"]))
        self.flags = 0
        self.reply = None            # (kind, to, term, a, b, c)
        self.requests = None         # (pre?, term, token, last_idx, last_term)
        self.next_event = False
        self.reset_query_index = False
        self.initialise_peers = False

    # ---- :3041-3066
    def update_term_and_voted_for(self, term, voted_for):
        if term == self.term and voted_for == self.voted_for:
            return
        self.term, self.voted_for = term, voted_for
        self.flags |= abi.F_PERSIST
        self.reset_query_index = True

    def update_term(self, term):
        if term > self.term:
            self.update_term_and_voted_for(term, abi.NONE)

    # ---- :2877-2924; the {next_event, cast, VoteForSelf} is the next message handled
    def call_for_election_candidate(self):
        new_term = self.term + 1
        self.requests = (False, new_term, 0) + self.last
        self.update_term_and_voted_for(new_term, self.me)
        self.role, self.leader_id, self.votes = abi.ROLE_CANDIDATE, abi.NONE, 0
        self.candidate_vote_result(new_term, True)

    def call_for_election_pre_vote(self, token):
        self.requests = (True, self.term, token) + self.last
        self.update_term_and_voted_for(self.term, self.me)
        self.role, self.leader_id, self.votes, self.token = abi.ROLE_PRE_VOTE, abi.NONE, 0, token
        self.pre_vote_result(self.term, True, token)

    # ---- handle_candidate(#request_vote_result{}) :1045-1068, :1131-1133
    def candidate_vote_result(self, term, granted):
        if granted and term == self.term:
            votes = self.votes + 1
            if votes == self.quorum:
                self.role, self.leader_id, self.votes = abi.ROLE_LEADER, self.me, 0
                self.initialise_peers = True
                self.flags |= abi.F_BECAME_LEADER
            else:
                self.votes = votes
        elif term > self.term:
            self.update_term_and_voted_for(term, abi.NONE)
            self.role = abi.ROLE_FOLLOWER

    # ---- handle_pre_vote(#pre_vote_result{}) :1218-1246
    def pre_vote_result(self, term, granted, token):
        if term > self.term:
            self.update_term(term)
            self.role, self.votes = abi.ROLE_FOLLOWER, 0
        elif granted and term == self.term and token == self.token and self.voter:
            votes = self.votes + 1
            if votes == self.quorum:
                self.call_for_election_candidate()
            else:
                self.votes = votes

    # ---- process_pre_vote/3 :2926-2983
    def process_pre_vote(self, fsm, m):
        term, token, frm = int(m["term"]), int(m["c"]), int(m["from"])
        if term >= self.term:
            self.update_term(term)
            up_to_date = (int(m["b"]), int(m["a"])) >= (self.last[1], self.last[0])   # :3000-3004
            theirs, ours, eff = int(m["n_entries"]), int(self.row["machine_version"]), \
                int(self.row["effective_machine_version"])
            if up_to_date and int(m["gap"]) > abi.PROTO_VERSION:
                self.reply = ("pre", frm, term, token, False)
            elif up_to_date and (theirs == eff or eff <= theirs <= ours):
                self.reply = ("pre", frm, term, token, True)
            elif up_to_date:
                self.reply = ("pre", frm, term, token, False)
                self.flags |= abi.F_START_ELECTION_TIMEOUT
            elif fsm == abi.ROLE_FOLLOWER:
                self.flags |= abi.F_START_ELECTION_TIMEOUT
            else:
                self.reply = ("pre", frm, term, token, False)
        else:
            self.reply = ("pre", frm, self.term, token, False)

    def aer_reply_false(self, to):                                    # append_entries_reply/3 :3624-3631
        self.reply = ("aer", to, self.term, self.last[0] + 1, self.last_written[0], self.last_written[1])

    def unhandled(self):
        self.flags |= abi.F_UNHANDLED


def handle_candidate(s: Srv, m):
    k, term, frm = int(m["kind"]), int(m["term"]), int(m["from"])
    if k == abi.MSG_VOTE_RESULT:
        s.candidate_vote_result(term, bool(int(m["flags"]) & abi.MF_SUCCESS))
    elif k in (abi.MSG_AER, abi.MSG_HEARTBEAT_RPC):
        if term >= s.term:                                            # :1069-1073, :1079-1082
            s.update_term_and_voted_for(term, abi.NONE)
            s.role, s.next_event = abi.ROLE_FOLLOWER, True
        elif k == abi.MSG_AER:                                        # :1074-1078
            s.aer_reply_false(frm)
        else:                                                         # :1083-1088
            s.reply = ("hb", frm, s.term, int(m["a"]))
    elif k in (abi.MSG_HEARTBEAT_REPLY, abi.MSG_AER_REPLY):           # :1089-1107
        if term > s.term:
            s.update_term_and_voted_for(term, abi.NONE)
            s.role = abi.ROLE_FOLLOWER
        else:
            s.unhandled()
    elif k in (abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC):
        if term > s.term:                                             # :1108-1123
            s.update_term_and_voted_for(term, abi.NONE)
            s.role, s.next_event = abi.ROLE_FOLLOWER, True
        elif k == abi.MSG_REQUEST_VOTE:                               # :1124-1126
            s.reply = ("vote", frm, s.term, False)
        else:                                                         # :1127-1131
            s.process_pre_vote(abi.ROLE_CANDIDATE, m)
    elif k == abi.MSG_PRE_VOTE_RESULT:                                # :1134-1136
        pass
    elif k == abi.MSG_ELECTION_TIMEOUT:                               # :1161-1162
        s.call_for_election_candidate()


def handle_pre_vote(s: Srv, m):
    k, term, frm = int(m["kind"]), int(m["term"]), int(m["from"])
    if k in (abi.MSG_AER, abi.MSG_HEARTBEAT_RPC):
        if term >= s.term:                                            # :1190-1201
            s.update_term(term)
            s.role, s.votes, s.next_event = abi.ROLE_FOLLOWER, 0, True
        elif k == abi.MSG_HEARTBEAT_RPC:                              # :1202-1206
            s.reply = ("hb", frm, s.term, int(m["a"]))
        else:
            s.unhandled()                                             # no clause for an older append_entries_rpc
    elif k == abi.MSG_HEARTBEAT_REPLY:
        if term > s.term:                                             # :1207-1210
            s.update_term(term)
            s.role, s.votes = abi.ROLE_FOLLOWER, 0
        else:
            s.unhandled()
    elif k == abi.MSG_REQUEST_VOTE:
        if term > s.term:                                             # :1211-1216
            s.update_term(term)
            s.role, s.votes, s.next_event = abi.ROLE_FOLLOWER, 0, True
        else:
            s.unhandled()
    elif k == abi.MSG_PRE_VOTE_RESULT:
        s.pre_vote_result(term, bool(int(m["flags"]) & abi.MF_SUCCESS), int(m["c"]))
    elif k == abi.MSG_PRE_VOTE_RPC:                                   # :1250-1251
        s.process_pre_vote(abi.ROLE_PRE_VOTE, m)
    elif k == abi.MSG_VOTE_RESULT:                                    # :1252-1254
        pass
    elif k == abi.MSG_ELECTION_TIMEOUT:                               # :1255-1256
        s.call_for_election_pre_vote(int(m["c"]))
    elif k == abi.MSG_AER_REPLY:
        s.unhandled()


def handle_follower_election(s: Srv, m):
    k = int(m["kind"])
    if k == abi.MSG_ELECTION_TIMEOUT:                                 # :1618-1626
        if s.voter:
            s.call_for_election_pre_vote(int(m["c"]))
    elif k == abi.MSG_PRE_VOTE_RPC:                                   # :1475-1482
        if s.voter:
            s.process_pre_vote(abi.ROLE_FOLLOWER, m)


def random_msg(rng, server, row, n, kinds):
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"] = server
    k = int(rng.choice(kinds))
    m["kind"] = k
    cur = int(row["current_term"])
    m["term"] = max(0, cur + int(rng.choice([-1, 0, 0, 0, 1, 2])))
    others = [i for i in range(n) if i != int(row["self"])] or [0]
    m["from"] = int(rng.choice(others))
    li, lt = int(row["last_index"]), int(row["last_term"])
    if k in (abi.MSG_VOTE_RESULT, abi.MSG_PRE_VOTE_RESULT):
        m["flags"] = abi.MF_SUCCESS if rng.random() < 0.7 else 0
        m["c"] = int(row["pre_vote_token"]) if rng.random() < 0.8 else int(rng.integers(0, 5))
    elif k in (abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC):
        m["a"] = max(0, li + int(rng.integers(-2, 3)))
        m["b"] = max(0, lt + int(rng.integers(-1, 2)))
        m["c"] = int(rng.integers(1, 1000))
        m["n_entries"] = int(rng.choice([0, int(row["effective_machine_version"]), int(row["machine_version"]),
                                         int(row["machine_version"]) + 1]))
        m["gap"] = abi.PROTO_VERSION + (1 if rng.random() < 0.1 else 0)
    elif k == abi.MSG_ELECTION_TIMEOUT:
        m["c"] = int(rng.integers(1, 1 << 30))
    elif k in (abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY):
        m["a"] = int(rng.integers(0, 9))
    elif k == abi.MSG_AER:
        m["a"], m["b"], m["c"] = li, lt, int(row["commit_index"])
    elif k == abi.MSG_AER_REPLY:
        m["flags"] = abi.MF_SUCCESS if rng.random() < 0.5 else 0
        m["a"], m["b"], m["c"] = li + 1, li, lt
    return m


KINDS = [abi.MSG_VOTE_RESULT, abi.MSG_PRE_VOTE_RESULT, abi.MSG_REQUEST_VOTE, abi.MSG_PRE_VOTE_RPC,
         abi.MSG_ELECTION_TIMEOUT, abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY, abi.MSG_AER, abi.MSG_AER_REPLY]


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (7, 6), (8, 7)])
def test_election_clauses_match_the_model(oracle_lib, n, seed):
    rng = np.random.default_rng(4000 + seed)
    G = 120
    st = fuzz.random_states(rng, G, n, max_runs=6)
    # only the three roles under test; votes below the quorum so a single result can complete it
    roles = rng.choice([abi.ROLE_CANDIDATE, abi.ROLE_PRE_VOTE, abi.ROLE_FOLLOWER], size=len(st), p=[0.4, 0.4, 0.2])
    st["role"] = roles
    st["cond_reason"] = 0
    st["votes"] = rng.integers(0, n // 2 + 1, size=len(st))
    st["pre_vote_token"] = rng.integers(0, 5, size=len(st))
    cand = roles != abi.ROLE_FOLLOWER
    st["voted_for"][cand] = st["self"][cand]                # a candidate has voted for itself
    st["leader_id"][cand] = abi.NONE
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    seen = {"leader": 0, "next_event": 0, "unhandled": 0, "candidate_from_pre": 0, "requests": 0, "replies": 0}
    for rnd in range(6):
        before = cpu.get_state()
        msgs = []
        for sv in range(len(before)):
            role = int(before[sv]["role"])
            if role == abi.ROLE_FOLLOWER:
                kinds = [abi.MSG_ELECTION_TIMEOUT, abi.MSG_PRE_VOTE_RPC]
            elif role in (abi.ROLE_CANDIDATE, abi.ROLE_PRE_VOTE):
                kinds = KINDS
            else:
                continue
            msgs.append(random_msg(rng, sv, before[sv], n, kinds)[0])
        if not msgs:
            break
        msgs = np.array(msgs, dtype=abi.MSG_DTYPE)
        dec, _ = cpu.step(msgs)
        after = cpu.get_state()
        for m, d in zip(msgs, dec):
            sv = int(m["server"])
            row0, row1 = before[sv], after[sv]
            s = Srv(row0)
            {abi.ROLE_CANDIDATE: handle_candidate, abi.ROLE_PRE_VOTE: handle_pre_vote,
             abi.ROLE_FOLLOWER: handle_follower_election}[s.role](s, m)
            tag = f"N={n} round {rnd} server {sv} role {abi.ROLE_NAMES[int(row0['role'])]} msg {m}"
            fl = int(d["flags"])
            if s.next_event:
                # first half modelled; second half by consistency with a server already in that state
                seen["next_event"] += 1
                mid = row0.copy()
                mid["role"], mid["current_term"], mid["voted_for"], mid["votes"] = s.role, s.term, s.voted_for, s.votes
                if s.reset_query_index:
                    mid["peer_query_index"] = 0
                mid["status_mask"] = 0xFF                            # become(follower, ..) :2182-2192
                two = oracle_lib.Oracle(1, n)
                base = (sv // n) * n
                grp = before[base:base + n].copy()
                grp[sv - base] = mid
                two.set_state(0, grp)
                m2 = m.copy(); m2["server"] = sv - base
                d2, _ = two.step(np.array([m2], dtype=abi.MSG_DTYPE))
                got = two.get_state()[sv - base]
                assert got.tobytes() == row1.tobytes(), tag
                assert fl & abi.F_REPROCESSED and fl & abi.F_ROLE_CHANGED, tag
                same = ~(abi.F_REPROCESSED | abi.F_ROLE_CHANGED | abi.F_PERSIST | abi.F_LEADER_CHANGED)
                assert (fl & same) == (int(d2["flags"][0]) & same), tag
                assert bool(fl & abi.F_PERSIST) == bool((s.flags | int(d2["flags"][0])) & abi.F_PERSIST), tag
                for f in ("reply_to", "reply_term", "reply_next_index", "reply_last_index", "reply_last_term"):
                    assert int(d[f]) == int(d2[f][0]), (tag, f)
                continue
            # ---- state
            assert int(row1["role"]) == s.role, tag
            assert int(row1["current_term"]) == s.term, tag
            assert int(row1["voted_for"]) == s.voted_for, tag
            assert int(row1["votes"]) == s.votes, tag
            assert int(row1["leader_id"]) == s.leader_id, tag
            assert int(row1["pre_vote_token"]) == s.token, tag
            if s.reset_query_index:
                assert not row1["peer_query_index"].any(), tag
            if s.role == abi.ROLE_FOLLOWER and int(row0["role"]) != abi.ROLE_FOLLOWER:
                assert int(row1["status_mask"]) == 0xFF, tag         # become(follower, ..) :2182-2192
            if s.initialise_peers:                                    # initialise_peers/1 :3234-3242
                for i in range(n):
                    if (int(row1["present_mask"]) >> i) & 1:
                        assert int(row1["next_index"][i]) == s.last[0] + 1 and int(row1["match_index"][i]) == 0, tag
                seen["leader"] += 1
            else:
                for f in ("match_index", "next_index", "commit_index_sent"):
                    assert np.array_equal(row0[f], row1[f]), (tag, f)
            for f in ("commit_index", "last_applied", "last_index", "last_term", "last_written_index", "n_runs"):
                assert int(row0[f]) == int(row1[f]), (tag, f)
            # ---- effects
            want = s.flags
            if s.reply:
                kind = s.reply[0]
                want |= abi.F_REPLY | {"pre": abi.F_REPLY_PRE_VOTE, "vote": abi.F_REPLY_VOTE, "hb": abi.F_REPLY_HEARTBEAT,
                                       "aer": 0}[kind]
                assert int(d["reply_to"]) == s.reply[1] and int(d["reply_term"]) == s.reply[2], tag
                if kind == "pre":
                    assert int(d["reply_next_index"]) == s.reply[3], tag
                    want |= abi.F_REPLY_SUCCESS if s.reply[4] else 0
                elif kind == "vote":
                    want |= abi.F_REPLY_SUCCESS if s.reply[3] else 0
                elif kind == "hb":
                    assert int(d["reply_next_index"]) == s.reply[3], tag
                else:
                    assert (int(d["reply_next_index"]), int(d["reply_last_index"]), int(d["reply_last_term"])) == s.reply[3:], tag
                seen["replies"] += 1
            if s.requests:
                pre, term, token, li, lt = s.requests
                want |= abi.F_SEND_VOTE_REQUESTS | (abi.F_PRE_VOTE_REQS if pre else 0)
                assert int(d["reply_term"]) == term and int(d["reply_last_index"]) == li and \
                    int(d["reply_last_term"]) == lt, tag
                if pre:
                    assert int(d["reply_next_index"]) == token, tag
                seen["requests"] += 1
                if int(row0["role"]) == abi.ROLE_PRE_VOTE and s.role in (abi.ROLE_CANDIDATE, abi.ROLE_LEADER) \
                        and int(m["kind"]) == abi.MSG_PRE_VOTE_RESULT:
                    seen["candidate_from_pre"] += 1
            assert (fl & ELECTION_FLAGS) == want, (tag, hex(fl & ELECTION_FLAGS), hex(want))
            assert bool(fl & abi.F_ROLE_CHANGED) == (s.role != int(row0["role"])), tag
            seen["unhandled"] += bool(fl & abi.F_UNHANDLED)
    assert seen["next_event"] > 20 and seen["replies"] > 20 and seen["requests"] > 10 and seen["unhandled"] > 5, seen
    assert seen["leader"] > 0, seen
    if n > 1:
        assert seen["candidate_from_pre"] > 0, seen
