"""A minimal host shell around the batched engine: what ra_server_proc does with the effects of a
transition, for every member of every group at once, with entry payloads and the state machines kept
on the host (they never cross the C ABI).

    shell = RaShell(engine, n_groups, n_members, machine=lambda: KvMachine())
    shell.trigger_election(group, member)
    shell.run(20)
    shell.command(group, ("put", "k", 1))
    shell.run_until_quiet()
    shell.machines[group * n_members + member].state

`engine` is ra_amd.engine.RaGpuBatch on an MI355X (the shell itself has no compute path and no
fallback: it only calls step(msgs) -> (decisions, rpcs) and get_state(), which is also how the CPU tests
put it in front of their reference checker).  One `tick()` hands every server at most
one pending message (mailbox order), submits the batch, and turns the decisions into the next
messages with ra_amd.effects -- the same routing erlang/ra_gpu_batch.erl does for a real Ra node:

  * {send_rpc, Peer, #append_entries_rpc{}}: the entries' payloads are read from the leader's
    in-memory log and travel with the message on the host side;
  * RGB_F_WROTE / commands: payloads go into the member's log, a `written` event follows (the WAL
    stand-in confirms on the next tick);
  * RGB_F_APPLIED: entries last_applied_before+1 .. last_applied are applied to the member's machine;
  * replies, vote requests, heartbeats, pipeline_rpcs, the post-election noop: as in the reference.

Loopback transport only (all members live in this process), in-memory logs, no snapshots: this is the
caller of the hot path, small enough to read, not a storage engine.  Delivery is reliable and FIFO per
pair, which is what ra_server assumes of Erlang distribution."""
from __future__ import annotations

from collections import deque
from typing import Callable, Dict, List, Optional

import numpy as np

from . import abi, effects as fx

NOOP = ("$ra", "noop")


class KvMachine:
    """A ra_machine in miniature: apply/2 over ("put", K, V) / ("delete", K)."""

    def __init__(self):
        self.state: Dict = {}
        self.applied = 0

    def apply(self, index: int, command):
        self.applied = index
        if command == NOOP:
            return
        if command[0] == "put":
            self.state[command[1]] = command[2]
        elif command[0] == "delete":
            self.state.pop(command[1], None)


class RaShell:
    def __init__(self, engine, n_groups: int, n_members: int, machine: Callable[[], object] = KvMachine):
        self.eng, self.G, self.N = engine, n_groups, n_members
        self.S = n_groups * n_members
        self.mailbox: List[deque] = [deque() for _ in range(self.S)]     # (rgb_msg, payloads or None)
        self.logs: List[Dict[int, object]] = [dict() for _ in range(self.S)]   # index -> payload
        self.machines = [machine() for _ in range(self.S)]
        self.pending_commands: List[deque] = [deque() for _ in range(self.S)]  # payloads of queued Commands
        self.token = 0
        self.down: set = set()                        # partitioned members: nothing in, nothing out
        self.state = engine.get_state()
        self.ticks = 0

    # ---------------------------------------------------------------- client side
    def leader_of(self, group: int) -> Optional[int]:
        rows = self.state[group * self.N:(group + 1) * self.N]
        best = None
        for slot, r in enumerate(rows):
            if int(r["role"]) == abi.ROLE_LEADER and (best is None or int(r["current_term"]) > int(rows[best]["current_term"])):
                best = slot
        return best

    def trigger_election(self, group: int, member: int):
        self.token += 1
        self._post(group * self.N + member, fx.encode(group * self.N + member, fx.ElectionTimeout(self.token)))

    def command(self, group: int, payload) -> bool:
        """{command, _} to the group's leader; False when the group has none right now."""
        lead = self.leader_of(group)
        if lead is None:
            return False
        s = group * self.N + lead
        self.pending_commands[s].append(payload)
        self._post(s, fx.encode(s, fx.Commands(1)))
        return True

    def tick_leaders(self):
        """tick_timeout on every leader: make_rpcs/1 re-sends to stale peers."""
        for s in range(self.S):
            if int(self.state[s]["role"]) == abi.ROLE_LEADER:
                self._post(s, fx.encode(s, fx.TICK_TIMEOUT))

    # ---------------------------------------------------------------- the loop
    def partition(self, group: int, member: int):
        """Cut one member off: its mailbox is discarded and nothing reaches it until heal()."""
        s = group * self.N + member
        self.down.add(s)
        self.mailbox[s].clear()

    def heal(self, group: int, member: int):
        self.down.discard(group * self.N + member)

    def _post(self, server: int, msg, payloads=None):
        if server not in self.down:
            self.mailbox[server].append((msg, payloads))

    def quiet(self) -> bool:
        return not any(self.mailbox)

    def tick(self) -> int:
        batch = [(s, *self.mailbox[s].popleft()) for s in range(self.S) if self.mailbox[s]]
        self.ticks += 1
        if not batch:
            return 0
        msgs = np.array([m for _, m, _ in batch], dtype=abi.MSG_DTYPE)
        before = self.state
        dec, rpcs = self.eng.step(msgs)
        self.state = after = self.eng.get_state()
        by_msg: Dict[int, list] = {}
        for r in rpcs:
            by_msg.setdefault(int(r["msg_index"]), []).append(r)
        for i, ((s, m, payloads), d) in enumerate(zip(batch, dec)):
            self._handle_effects(s, m, payloads, d, by_msg.get(i, []), before[s], after[s])
        return len(batch)

    def run(self, ticks: int):
        for _ in range(ticks):
            self.tick()

    def run_until_quiet(self, max_ticks: int = 10_000) -> int:
        n = 0
        while not self.quiet():
            self.tick()
            n += 1
            if n >= max_ticks:
                raise RuntimeError("the cluster did not settle")
        return n

    # ---------------------------------------------------------------- effects
    def _handle_effects(self, s, m, payloads, d, rpcs, st0, st1):
        g, me = s // self.N, s % self.N
        peer = lambda slot: g * self.N + int(slot)
        fl, kind = int(d["flags"]), int(m["kind"])
        log = self.logs[s]
        # the log writes of this transition (payloads stay here; the engine moved the cursors)
        if kind == abi.MSG_APPEND and int(st1["last_index"]) > int(st0["last_index"]) and not fl & abi.F_INVARIANT:
            first, last = int(st0["last_index"]) + 1, int(st1["last_index"])
            for idx in range(first, last + 1):
                noop = bool(int(m["flags"]) & abi.MF_FORCE)
                log[idx] = NOOP if noop else self.pending_commands[s].popleft()
            self._post(s, fx.encode(s, fx.Written(int(st1["current_term"]), first, last)))
        elif kind == abi.MSG_APPEND and not int(m["flags"]) & abi.MF_FORCE and self.pending_commands[s]:
            self.pending_commands[s].popleft()                       # not the leader any more: the command is lost
        if fl & abi.F_WROTE:
            first, last = int(d["reply_next_index"]), int(d["reply_last_index"])
            base = int(m["a"]) + 1 + int(m["gap"])
            for idx in [k for k in log if k >= first]:               # an overwrite drops the old tail
                del log[idx]
            for idx in range(first, last + 1):
                log[idx] = payloads[idx - base]
            split = base + int(m["n_run0"])
            if first < split:
                self._post(s, fx.encode(s, fx.Written(int(m["run0_term"]), first, min(last, split - 1))))
            if last >= split:
                self._post(s, fx.encode(s, fx.Written(int(m["run1_term"]), max(first, split), last)))
        if fl & abi.F_TRUNCATED:
            for idx in [k for k in log if k > int(st1["last_index"])]:
                del log[idx]
        if fl & abi.F_APPLIED:
            mac = self.machines[s]
            for idx in range(int(st0["last_applied"]) + 1, int(st1["last_applied"]) + 1):
                mac.apply(idx, log[idx])
        for e in fx.decode(m, d, rpcs, st1, self.N):
            tag = e if isinstance(e, str) else e[0]
            if tag == "exit":
                raise RuntimeError(f"server {s}: the reference would exit with invariant {e[1]} on {m}")
            if tag == "reply":
                to = peer(int(m["from"]))
                self._post(to, fx.encode(to, e[1], from_slot=me))
            elif tag == "cast":
                to = peer(e[1])
                self._post(to, fx.encode(to, e[2][1], from_slot=e[2][0]))
            elif tag == "send_vote_requests":
                for slot, rec in e[1]:
                    self._post(peer(slot), fx.encode(peer(slot), rec))
            elif tag == "send_rpc":
                to = peer(e[1])
                if isinstance(e[2], fx.AppendEntriesRpc):
                    for piece in fx.split_entries(e[2]):
                        self._post(to, fx.encode(to, piece), [log[i] for i, _ in piece.entries])
                else:
                    self._post(to, fx.encode(to, e[2]))
            elif tag == "send_snapshot":
                raise NotImplementedError("this shell takes no snapshots, so none can be needed")
            elif tag == "next_event" and e[1] == "info":             # next_event: ahead of the mailbox
                self.mailbox[s].appendleft((fx.encode(s, fx.PIPELINE_RPCS), None))
            elif tag == "next_event":
                self.mailbox[s].appendleft((fx.encode(s, fx.Commands(1, noop=True)), None))
