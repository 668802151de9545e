"""Closed-loop driver for the batched engine: the part of ra_server_proc that turns the effects of
one transition into the next messages (TEST TOOLING; the record/effect vocabulary and the encoding are
ra_amd/effects.py, the Python twin of erlang/ra_gpu_batch.erl).

Every member of every group is a row of the engine (or of the checker).  A tick hands each server at
most one message; the decisions and rpc records that come back are routed exactly the way the owning
gen_statem would route the reference's effects:

  {send_rpc, Peer, #append_entries_rpc{}}      rgb_rpc record   -> AER to the peer (entries' terms read
                                                                  from the leader's log, like ra_log does)
  {cast, Leader, {Id, #append_entries_reply{}}} RGB_F_REPLY      -> AER_REPLY
  {reply, #request_vote_result{}} / pre_vote    RGB_F_REPLY_VOTE / _PRE_VOTE -> VOTE_RESULT / PRE_VOTE_RESULT
  {send_vote_requests, _}                       RGB_F_SEND_VOTE_REQUESTS     -> REQUEST_VOTE / PRE_VOTE_RPC
  {next_event, info, pipeline_rpcs}             RGB_F_PIPELINE   -> PIPELINE_RPCS to itself
  post_election_effects: noop command           RGB_F_BECAME_LEADER -> APPEND(1, force)
  ra_log:write / ra_log:append -> WAL           RGB_F_WROTE / APPEND -> {written, Term, Seq} later, in
                                                order, one event per term (ra_log_wal batch writers,
                                                src/ra_log_wal.erl:596-635, 785-807)
  heartbeat_rpc / heartbeat_reply               RGB_F_SEND_HEARTBEATS / RGB_F_REPLY_HEARTBEAT
  leader tick_timeout -> ra_server:make_rpcs/1  PIPELINE_RPCS with RGB_MF_TICK (re-sends to stale peers)

The network between members drops and delays, and reorders between different senders (never between
the same two members, never duplicating: Erlang distribution); local events (WAL, next_event) are
reliable and ordered.  Timers (election_timeout, await_condition_timeout) and client commands
fire at random.  With p_wal_down > 0 a member's WAL goes down for a few ticks now and then: a follower's
ra_log:write/2 then answers {error, wal_down} and a leader's ra_log:append/2 raises wal_down -- host I/O, so the host
recipes of INTEGRATION.md put the server into await_condition (RGB_COND_WAL_DOWN / RGB_COND_WAL_DOWN_LEADER) and
RGB_MF_CAN_WRITE says when ra_log:can_write/1 is true again.  With p_snapshot > 0 members also take snapshots (SNAPSHOT_WRITTEN truncates their logs) and a
leader whose peer fell behind its snapshot sends it: that transfer is ra_server's business, emulated
here on the rows and written back with set_state.  The Raft safety properties are checked on what the
logs still hold after every tick (check_safety)."""
from __future__ import annotations

from collections import deque

import numpy as np

from ra_amd import abi, effects


class ClusterSim:
    def __init__(self, eng, n_groups, n_members, seed, drop=0.1, max_delay=3,
                 p_election=0.02, p_command=0.3, p_query=0.05, p_tick=0.3, max_leaders=9, p_snapshot=0.0,
                 p_wal_down=0.0):
        self.eng, self.G, self.N = eng, n_groups, n_members
        self.S = n_groups * n_members
        self.rng = np.random.default_rng(seed)
        self.drop, self.max_delay = drop, max_delay
        self.p_election, self.p_command, self.p_query, self.p_tick = p_election, p_command, p_query, p_tick
        self.max_leaders = max_leaders
        self.p_snapshot = p_snapshot                     # > 0: members take snapshots at last_applied
        self.p_wal_down = p_wal_down                     # > 0: per member and tick, its WAL goes down for 2..7 ticks
        self.wal_injection = p_wal_down > 0              # (stays set after heal(): RGB_MF_CAN_WRITE keeps being passed)
        self.wal_up_at = np.zeros(self.S, dtype=np.int64)   # first tick at which ra_log:can_write/1 is true again
        self.host = []                                   # host-level snapshot transfers: (deliver_at, what, args)
        self.tick = 0
        self.net = [[] for _ in range(self.S)]          # [(deliver_at, msg)]
        self.local = [deque() for _ in range(self.S)]   # reliable, ordered
        self.wal = [[] for _ in range(self.S)]          # writes the WAL has not confirmed: (first, last, term)
        self.token = 1
        self.leaders_of_term = [dict() for _ in range(n_groups)]   # term -> member slot
        self.committed = [dict() for _ in range(n_groups)]         # index -> term, once any member committed it
        self.elections = np.zeros(n_groups, dtype=np.int64)
        self.stats = {"msgs": 0, "dropped": 0, "invariants": 0, "commands": 0, "queries_answered": 0,
                      "snapshots": 0, "installs": 0, "install_refused": 0,
                      "wal_down_follower": 0, "wal_down_leader": 0, "transfer_leadership": 0, "wal_down_reprocessed": 0}
        self.history = []                                # the batches fed to the engine, for replay
        self.leader_contact = np.full(self.S, -10**9, dtype=np.int64)   # tick of the last {record_leader_msg, _}
        self.election_silence = 0                        # ticks without a leader message before a timeout may fire
        self.quiet = 0                                   # consecutive ticks that produced no effect to route
        self.state = eng.get_state()

    # ------------------------------------------------------------------ network
    def send(self, to_server, msg):
        """Erlang distribution between two processes: messages may be lost (a dropped connection) but are
        never duplicated and never overtake each other.  ra_server relies on both -- granted votes are
        counted with a plain counter (src/ra_server.erl:1045-1061), and an old empty append_entries_rpc
        overtaken by newer ones would trip ?assertNot(PLIdx < LastApplied) (:1317) -- so the simulated
        network drops and delays, and reorders only between different senders."""
        if self.rng.random() < self.drop:
            self.stats["dropped"] += 1
            return
        self.net[to_server].append((self.tick + 1 + int(self.rng.integers(0, self.max_delay)), msg))

    def heal(self):
        """A reliable network from now on, and election timers that behave: they only fire after a
        silence from the leader (liveness checks)."""
        self.drop = 0.0
        self.p_wal_down = 0.0
        self.election_silence = 40

    def idle(self):
        return not any(self.net) and not any(self.local) and not any(self.wal) and not self.host

    # ------------------------------------------------------------------ one tick
    def choose(self, s):
        st = self.state[s]
        role = int(st["role"])
        r = self.rng.random()
        if self.local[s] and r < 0.8:
            return self.local[s].popleft()
        # the oldest message of every sender (FIFO per pair), once its delay has passed
        heads, ready = set(), []
        for k, (at, msg) in enumerate(self.net[s]):
            frm = int(msg["from"])
            if frm in heads:
                continue
            heads.add(frm)
            if at <= self.tick:
                ready.append(k)
        if ready and r < 0.97:
            k = ready[int(self.rng.integers(0, len(ready)))]
            return self.net[s].pop(k)[1]
        g = s // self.N
        if self.p_snapshot and self.rng.random() < self.p_snapshot:
            # ra_snapshot finished writing a snapshot of the machine at last_applied
            la = int(st["last_applied"])
            si = 0 if int(st["snapshot_index"]) == abi.UNDEF_INT else int(st["snapshot_index"])
            if la >= si + 3 and la >= int(st["first_index"]):
                term = dict(abi.log_entries(st))[la]
                self.stats["snapshots"] += 1
                return effects.encode(s, effects.SnapshotWritten(la, term))
        if role == abi.ROLE_LEADER:
            if self.rng.random() < self.p_command:
                self.stats["commands"] += 1
                return effects.encode(s, effects.Commands(int(self.rng.integers(1, 4))))
            if self.rng.random() < self.p_query:
                return effects.encode(s, effects.CONSISTENT_QUERY)
            if self.rng.random() < self.p_tick:                      # tick_timeout -> make_rpcs/1
                return effects.encode(s, effects.TICK_TIMEOUT)
        elif role == abi.ROLE_AWAIT_CONDITION:
            if self.rng.random() < 0.2:
                return effects.encode(s, effects.AWAIT_CONDITION_TIMEOUT)
        elif (self.elections[g] < self.max_leaders or
              int(self.state["n_runs"][g * self.N:(g + 1) * self.N].max()) <= 5) and \
                self.rng.random() < self.p_election and \
                self.tick - self.leader_contact[s] >= self.election_silence:
            self.token += 1
            return effects.encode(s, effects.ElectionTimeout(self.token))
        return None

    def edit(self, s, row):
        """A host-side change of one server's integer state (rgb_upload_state): recorded for replay."""
        self.eng.set_state(s, row.reshape(1))
        self.history.append(("set", s, row.copy()))
        self.state = self.eng.get_state()

    def step(self):
        self.run_host_events()
        if self.p_wal_down:
            hit = (self.rng.random(self.S) < self.p_wal_down) & (self.wal_up_at <= self.tick)
            self.wal_up_at[hit] = self.tick + 2 + self.rng.integers(0, 6, size=int(hit.sum()))
        batch = [m for m in (self.choose(s) for s in range(self.S)) if m is not None]
        self.now = self.tick                             # the tick these messages are processed in
        self.tick += 1
        if not batch:
            self.flush_wals()
            return None
        msgs = np.array(batch, dtype=abi.MSG_DTYPE)
        # ra_log:can_write/1 as the owning process sees it when it hands the message over (wal_down_condition/2)
        if self.wal_injection:
            msgs["flags"] |= np.where(self.wal_up_at[msgs["server"]] <= self.now, abi.MF_CAN_WRITE, 0).astype(msgs["flags"].dtype)
        before = self.state
        dec, rpcs = self.eng.step(msgs)
        self.state = after = self.eng.get_state()
        self.history.append(msgs)
        self.stats["msgs"] += len(msgs)
        by_msg = {}
        for r in rpcs:
            by_msg.setdefault(int(r["msg_index"]), []).append(r)
        for i, (m, d) in enumerate(zip(msgs, dec)):
            self.route(m, d, by_msg.get(i, []), before[int(m["server"])], after[int(m["server"])])
        self.flush_wals()
        return msgs, dec, rpcs

    # ------------------------------------------------------------------ effects -> messages
    def route(self, m, d, rpcs, st0, st1):
        s = int(m["server"]); g = s // self.N; me = s % self.N
        fl = int(d["flags"]); kind = int(m["kind"])
        if fl & abi.F_LEADER_MSG:
            self.leader_contact[s] = self.tick
        if fl & abi.F_TRANSFER_LEADERSHIP:
            # {next_event, cast, {transfer_leadership, Peer}}: ra_server's own clause (not on the batched path); a
            # leader that declines the hint simply goes on leading, which is what happens here
            self.stats["transfer_leadership"] += 1
        if int(st0["role"]) == abi.ROLE_AWAIT_CONDITION and int(st0["cond_reason"]) in (abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER) \
                and fl & abi.F_REPROCESSED:
            self.stats["wal_down_reprocessed"] += 1
        if self.wal_up_at[s] > self.now and not fl & (abi.F_INVARIANT | abi.F_UNHANDLED):
            if fl & abi.F_WROTE:
                # follower: ra_log:write/2 -> {error, wal_down} (src/ra_server.erl:1377-1385): State1 (term, leader_id,
                # commit_index := LeaderCommit) with the log as it was, Effects0 = [{record_leader_msg, _}] only
                row = st0.copy()
                for k in ("current_term", "voted_for", "leader_id"):
                    row[k] = st1[k]
                row["commit_index"] = m["c"]
                row["role"] = abi.ROLE_AWAIT_CONDITION
                row["cond_reason"] = abi.COND_WAL_DOWN
                self.edit(s, row)
                self.stats["wal_down_follower"] += 1
                return
            if kind == abi.MSG_APPEND and int(st0["role"]) == abi.ROLE_LEADER and \
                    int(st1["last_index"]) > int(st0["last_index"]):
                # leader: ra_log:append/2 raised wal_down (:655-672): the state as it was, nothing pipelined, the
                # command answered with an error (a new leader's noop is simply lost: nothing commits in its term
                # until the next command)
                row = st0.copy()
                row["role"] = abi.ROLE_AWAIT_CONDITION
                row["cond_reason"] = abi.COND_WAL_DOWN_LEADER
                self.edit(s, row)
                self.stats["wal_down_leader"] += 1
                return
        peer = lambda slot: g * self.N + int(slot)
        for e in effects.decode(m, d, rpcs, st1, self.N):
            tag = e if isinstance(e, str) else e[0]
            if tag == "exit":
                self.stats["invariants"] += 1
                raise AssertionError(f"tick {self.tick}: server {s} hit reference invariant {e[1]} "
                                     f"on {m} in state {st0}")
            elif tag == "reply":                                     # {reply, _}: back to the caller
                to = int(m["from"])
                self.send(peer(to), effects.encode(peer(to), e[1], from_slot=me))
            elif tag == "cast":                                      # {cast, To, {Id, Reply}}
                to, (frm, rec) = e[1], e[2]
                self.send(peer(to), effects.encode(peer(to), rec, from_slot=frm))
            elif tag == "send_vote_requests":
                for slot, rec in e[1]:
                    self.send(peer(slot), effects.encode(peer(slot), rec))
            elif tag == "send_rpc":
                slot, rec = e[1], e[2]
                if isinstance(rec, effects.AppendEntriesRpc):
                    for piece in effects.split_entries(rec):         # at most two term runs per message
                        self.send(peer(slot), effects.encode(peer(slot), piece))
                else:
                    self.send(peer(slot), effects.encode(peer(slot), rec))
            elif tag == "send_snapshot":
                self.send_snapshot(s, e[1], e[2][0], e[2][1], int(st1["current_term"]))
            elif tag == "next_event" and e[1] == "info":
                self.local[s].append(effects.encode(s, effects.PIPELINE_RPCS))
            elif tag == "next_event":                                # the noop command of a new leader
                self.elections[g] += 1
                self.local[s].append(effects.encode(s, effects.Commands(1, noop=True)))
            elif tag in ("query_quorum", "query_apply"):
                self.stats["queries_answered"] += 1
        # the log writes of this transition go to the WAL
        if kind == abi.MSG_APPEND and not fl & abi.F_UNHANDLED and int(st1["last_index"]) > int(st0["last_index"]) \
                and int(st1["role"]) == abi.ROLE_LEADER and int(st0["role"]) == abi.ROLE_LEADER:
            self.wal[s].append((int(st0["last_index"]) + 1, int(st1["last_index"]), int(st1["current_term"])))
        if fl & abi.F_WROTE:
            base = int(m["a"]) + 1 + int(m["gap"])
            split = base + int(m["n_run0"])
            lo, hi = int(d["reply_next_index"]), int(d["reply_last_index"])
            if lo < split:
                self.wal[s].append((lo, min(hi, split - 1), int(m["run0_term"])))
            if hi >= split:
                self.wal[s].append((max(lo, split), hi, int(m["run1_term"])))

    # ------------------------------------------------------------------ snapshot transfer (host side)
    # {send_snapshot, Peer, _} is not part of the batched path: ra_server_proc spawns a sender, marks the
    # peer {sending_snapshot, _} (no rpcs to it meanwhile), the follower goes through receive_snapshot
    # and ra_log:install_snapshot, and the leader handles #install_snapshot_result{}
    # (src/ra_server.erl:764-792, 1556-1590, 1744-1806; src/ra_log.erl:1216-1260).  Those transitions
    # stay in ra_server; the shell re-uploads the integer state they produce.  Here they are emulated
    # on the rows and written back with set_state.
    def send_snapshot(self, leader, slot, idx, term, leader_term):
        row = self.state[leader].copy()
        row["status_mask"] &= np.uint8(~(1 << slot) & 0xFF)          # {sending_snapshot, Pid}
        self.edit(leader, row)
        g = leader // self.N
        self.host.append((self.tick + 1 + int(self.rng.integers(0, 4)), "install",
                          (leader, g * self.N + slot, idx, term, leader_term)))

    def run_host_events(self):
        due = [e for e in self.host if e[0] <= self.tick]
        self.host = [e for e in self.host if e[0] > self.tick]
        for _, what, args in due:
            getattr(self, "host_" + what)(*args)

    def host_install(self, leader, follower, idx, term, leader_term):
        st = self.state[follower]
        ok = (int(st["role"]) == abi.ROLE_FOLLOWER and leader_term >= int(st["current_term"])
              and idx > int(st["last_applied"]))
        if ok and (self.wal[follower] or any(int(m["kind"]) == abi.MSG_WRITTEN for m in self.local[follower])):
            # {awaiting_pending, ..}: the last chunk waits until the WAL has confirmed everything
            self.host.append((self.tick + 2, "install", (leader, follower, idx, term, leader_term)))
            return
        if not ok:
            self.stats["install_refused"] += 1
            self.host.append((self.tick + 1, "sender_down", (leader, follower % self.N, leader_term)))
            return
        row = st.copy()
        if leader_term > int(row["current_term"]):
            row["current_term"] = leader_term
            row["voted_for"] = abi.NONE
        row["leader_id"] = leader % self.N
        # ra_log:install_snapshot: range = undefined, last_term = SnapTerm, last_written = {SnapIdx, SnapTerm}
        row["snapshot_index"], row["snapshot_term"] = idx, term
        row["last_index"], row["last_term"] = idx, term
        row["last_written_index"], row["last_written_term"] = idx, term
        row["first_index"] = idx + 1
        row["pending_first"] = idx + 1
        row["n_runs"] = 0
        row["run_start"] = 0
        row["run_term"] = 0
        row["commit_index"] = row["last_applied"] = idx
        self.edit(follower, row)
        self.leader_contact[follower] = self.tick
        self.stats["installs"] += 1
        self.host.append((self.tick + 1 + int(self.rng.integers(0, 3)), "result",
                          (leader, follower % self.N, idx, leader_term)))

    def host_result(self, leader, slot, idx, leader_term):
        st = self.state[leader]
        if int(st["role"]) != abi.ROLE_LEADER or int(st["current_term"]) != leader_term:
            return                                                  # no longer that leader: the peers were reset
        row = st.copy()
        row["match_index"][slot] = idx
        row["next_index"][slot] = idx + 1
        row["commit_index_sent"][slot] = idx
        row["status_mask"] |= np.uint8(1 << slot)
        self.edit(leader, row)
        self.local[leader].append(effects.encode(leader, effects.PIPELINE_RPCS))

    def host_sender_down(self, leader, slot, leader_term):
        st = self.state[leader]
        if int(st["role"]) != abi.ROLE_LEADER or int(st["current_term"]) != leader_term:
            return
        row = st.copy()
        row["status_mask"] |= np.uint8(1 << slot)                   # 'DOWN' of the sender: back to normal
        self.edit(leader, row)

    def flush_wals(self):
        """complete_batch: one {written, Term, Seq} per writer and term, oldest first."""
        for s in range(self.S):
            if not self.wal[s] or self.rng.random() < 0.5:
                continue
            cur = None                                   # (term, lo, hi) of the open batch writer
            for lo, hi, term in self.wal[s]:
                if cur and cur[0] == term:
                    # ra_seq:append, or limit(Idx-1) then append on a rewrite (:607-616)
                    clo, chi = cur[1], cur[2]
                    if lo > chi:
                        assert lo == chi + 1
                        cur = (term, clo, hi)
                    else:
                        cur = (term, min(clo, lo), hi)
                else:
                    if cur:
                        self.local[s].append(effects.encode(s, effects.Written(cur[0], cur[1], cur[2])))
                    cur = (term, lo, hi)
            self.local[s].append(effects.encode(s, effects.Written(cur[0], cur[1], cur[2])))
            self.wal[s] = []

    # ------------------------------------------------------------------ Raft safety
    def check_safety(self):
        st = self.state
        for g in range(self.G):
            rows = st[g * self.N:(g + 1) * self.N]
            logs = [dict(abi.log_entries(r)) for r in rows]
            # election safety: at most one leader per term
            for slot, r in enumerate(rows):
                if int(r["role"]) == abi.ROLE_LEADER:
                    t = int(r["current_term"])
                    prev = self.leaders_of_term[g].setdefault(t, slot)
                    assert prev == slot, f"group {g}: members {prev} and {slot} both led term {t}"
            # log matching: same (index, term) => identical prefixes (as far as both logs still hold them)
            for a in range(self.N):
                for b in range(a + 1, self.N):
                    common = [i for i in logs[a] if i in logs[b] and logs[a][i] == logs[b][i]]
                    if common:
                        top = max(common)
                        lo = max(min(logs[a]), min(logs[b]), 1)
                        for i in range(lo, top + 1):
                            assert logs[a].get(i) == logs[b].get(i), \
                                f"group {g}: logs of {a} and {b} match at {top} but differ at {i}"
            # state machine safety: a committed index never changes its term, on any member
            for slot, r in enumerate(rows):
                # (a follower takes LeaderCommit as it comes, src/ra_server.erl:1331-1332, 1366: its
                # commit_index may step back below last_applied under a new leader; both bound what is known
                # committed)
                ci = max(int(r["commit_index"]), int(r["last_applied"]))
                if self.wal_injection:
                    # a refused write leaves the follower with commit_index := LeaderCommit over a log that still
                    # holds its stale suffix (src/ra_server.erl:1377-1385 keeps State1 with log => Log1) until the
                    # leader's resend overwrites it.  Harmless in the reference -- a follower only evaluates its
                    # commit index inside an append_entries_rpc clause that has validated or overwritten the suffix
                    # (:1331-1376, 2246-2259; written events do not, :1457-1474) -- so with WAL outages only what is
                    # APPLIED counts as known committed here
                    ci = int(r["last_applied"])
                assert int(r["last_applied"]) <= int(r["last_index"])
                for i in range(max(1, int(r["first_index"])), min(ci, int(r["last_index"])) + 1):
                    t = logs[slot][i]
                    was = self.committed[g].setdefault(i, t)
                    assert was == t, f"group {g} member {slot}: committed index {i} had term {was}, now {t}"
            for slot, r in enumerate(rows):
                si = int(r["snapshot_index"])
                if si != abi.UNDEF_INT and si in self.committed[g]:
                    assert self.committed[g][si] == int(r["snapshot_term"]), \
                        f"group {g} member {slot}: snapshot {si}:{int(r['snapshot_term'])} vs committed {self.committed[g][si]}"
            # leader completeness: a leader holds every entry committed so far
            for slot, r in enumerate(rows):
                if int(r["role"]) == abi.ROLE_LEADER and int(r["current_term"]) == max(self.leaders_of_term[g]):
                    for i, t in self.committed[g].items():
                        if i < int(r["first_index"]):
                            continue                                 # below its snapshot: committed by definition
                        assert logs[slot].get(i) == t, \
                            f"group {g}: leader {slot} of term {int(r['current_term'])} lacks committed {i}:{t}"
