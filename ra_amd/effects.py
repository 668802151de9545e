"""The reference's wire records and effects, in Python, on either side of the batched path.

`encode(server, record)` turns one of the records of src/ra.hrl:123-200 (or a local event) into the
rgb_msg the engine takes; `decode(msg, decision, rpcs, state_after)` turns what comes back into the
effects list the owning ra_server_proc would have received from ra_server:handle_<role>/2, with the
reference's names and field meanings.  This is the Python twin of erlang/ra_gpu_batch.erl
(encode_msg/3, decode_effects/3) -- the shell a host written in Python needs around the C ABI, and
what tests/cluster_sim.py routes with.

Member ids are member slots (0..7) of the server's group; `server = group * n_members + slot`.
Entries are (index, term) pairs: payloads never cross the boundary.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import abi


# ---------------------------------------------------------------- records (src/ra.hrl)
@dataclass(frozen=True)
class AppendEntriesRpc:                      # :123-129
    term: int
    leader_id: int
    leader_commit: int
    prev_log_index: int
    prev_log_term: int
    entries: Tuple[Tuple[int, int], ...] = ()


@dataclass(frozen=True)
class AppendEntriesReply:                    # :131-142
    term: int
    success: bool
    next_index: int
    last_index: int
    last_term: int


@dataclass(frozen=True)
class RequestVoteRpc:                        # :145-149
    term: int
    candidate_id: int
    last_log_index: int
    last_log_term: int


@dataclass(frozen=True)
class RequestVoteResult:                     # :152-154
    term: int
    vote_granted: bool


@dataclass(frozen=True)
class PreVoteRpc:                            # :157-166
    term: int
    token: int
    candidate_id: int
    last_log_index: int
    last_log_term: int
    machine_version: int = 0
    version: int = abi.PROTO_VERSION


@dataclass(frozen=True)
class PreVoteResult:                         # :168-171
    term: int
    token: int
    vote_granted: bool


@dataclass(frozen=True)
class HeartbeatRpc:                          # :193-196
    query_index: int
    term: int
    leader_id: int


@dataclass(frozen=True)
class HeartbeatReply:                        # :198-200
    query_index: int
    term: int


# local events of the owning process
@dataclass(frozen=True)
class Written:                               # {ra_log_event, {written, Term, [From..To]}}
    term: int
    first: int
    last: int


@dataclass(frozen=True)
class SnapshotWritten:                       # {ra_log_event, {snapshot_written, {Idx, Term}, _, snapshot, _, _}}
    index: int
    term: int


@dataclass(frozen=True)
class Commands:                              # {command, _} / {commands, _}: n entries appended by the leader
    n: int
    noop: bool = False                       # the post-election noop forces pipelining (src/ra_server.erl:682-689)


@dataclass(frozen=True)
class ElectionTimeout:
    token: int = 0                           # the make_ref() of call_for_election(pre_vote, _)


PIPELINE_RPCS = "pipeline_rpcs"
TICK_TIMEOUT = "tick_timeout"
AWAIT_CONDITION_TIMEOUT = "await_condition_timeout"
CONSISTENT_QUERY = "consistent_query"


def _blank(server: int, kind: int, frm: int = abi.NONE) -> np.ndarray:
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"], m["kind"], m["from"] = server, kind, frm
    return m


def encode(server: int, record, from_slot: int = abi.NONE) -> np.void:
    """One inbound message for `server`.  `from_slot` is the sending peer for the {Peer, Reply} forms."""
    r = record
    if isinstance(r, AppendEntriesRpc):
        m = _blank(server, abi.MSG_AER, r.leader_id)
        m["term"], m["a"], m["b"], m["c"] = r.term, r.prev_log_index, r.prev_log_term, r.leader_commit
        ents = list(r.entries)
        if ents:
            first = ents[0][0]
            if first <= r.prev_log_index or first - r.prev_log_index - 1 > 255:
                raise ValueError("entries must start above prev_log_index (gap <= 255)")
            for k, (i, _) in enumerate(ents):
                if i != first + k:
                    raise ValueError("entries must be contiguous")
            t0 = ents[0][1]
            n0 = next((k for k, (_, t) in enumerate(ents) if t != t0), len(ents))
            t1 = ents[n0][1] if n0 < len(ents) else 0
            if any(t != t1 for _, t in ents[n0:]):
                raise ValueError("at most two term runs ride in one message: split the rpc (split_entries)")
            m["gap"] = first - r.prev_log_index - 1
            m["n_entries"], m["n_run0"], m["run0_term"], m["run1_term"] = len(ents), n0, t0, t1
    elif isinstance(r, AppendEntriesReply):
        m = _blank(server, abi.MSG_AER_REPLY, from_slot)
        m["term"], m["flags"] = r.term, abi.MF_SUCCESS if r.success else 0
        m["a"], m["b"], m["c"] = r.next_index, r.last_index, r.last_term
    elif isinstance(r, RequestVoteRpc):
        m = _blank(server, abi.MSG_REQUEST_VOTE, r.candidate_id)
        m["term"], m["a"], m["b"] = r.term, r.last_log_index, r.last_log_term
    elif isinstance(r, RequestVoteResult):
        m = _blank(server, abi.MSG_VOTE_RESULT, from_slot)
        m["term"], m["flags"] = r.term, abi.MF_SUCCESS if r.vote_granted else 0
    elif isinstance(r, PreVoteRpc):
        m = _blank(server, abi.MSG_PRE_VOTE_RPC, r.candidate_id)
        m["term"], m["a"], m["b"], m["c"] = r.term, r.last_log_index, r.last_log_term, r.token
        m["n_entries"], m["gap"] = r.machine_version, r.version
    elif isinstance(r, PreVoteResult):
        m = _blank(server, abi.MSG_PRE_VOTE_RESULT, from_slot)
        m["term"], m["c"], m["flags"] = r.term, r.token, abi.MF_SUCCESS if r.vote_granted else 0
    elif isinstance(r, HeartbeatRpc):
        m = _blank(server, abi.MSG_HEARTBEAT_RPC, r.leader_id)
        m["term"], m["a"] = r.term, r.query_index
    elif isinstance(r, HeartbeatReply):
        m = _blank(server, abi.MSG_HEARTBEAT_REPLY, from_slot)
        m["term"], m["a"] = r.term, r.query_index
    elif isinstance(r, Written):
        m = _blank(server, abi.MSG_WRITTEN)
        m["term"], m["a"], m["b"] = r.term, r.first, r.last
    elif isinstance(r, SnapshotWritten):
        m = _blank(server, abi.MSG_SNAPSHOT_WRITTEN)
        m["a"], m["b"] = r.index, r.term
    elif isinstance(r, Commands):
        m = _blank(server, abi.MSG_APPEND)
        m["n_entries"], m["flags"] = r.n, abi.MF_FORCE if r.noop else 0
    elif isinstance(r, ElectionTimeout):
        m = _blank(server, abi.MSG_ELECTION_TIMEOUT)
        m["c"] = r.token
    elif r == PIPELINE_RPCS:
        m = _blank(server, abi.MSG_PIPELINE_RPCS)
    elif r == TICK_TIMEOUT:
        m = _blank(server, abi.MSG_PIPELINE_RPCS)
        m["flags"] = abi.MF_TICK
    elif r == AWAIT_CONDITION_TIMEOUT:
        m = _blank(server, abi.MSG_AWAIT_TIMEOUT)
    elif r == CONSISTENT_QUERY:
        m = _blank(server, abi.MSG_CONSISTENT_QUERY)
    else:
        raise TypeError(f"not a message of the batched path: {record!r}")
    return m[0]


def split_entries(rpc: AppendEntriesRpc) -> List[AppendEntriesRpc]:
    """An append_entries_rpc whose entries span more than two terms, cut into consecutive rpcs of at
    most two term runs each (what one rgb_msg carries); the follower sees them as a pipelined series."""
    ents = list(rpc.entries)
    out, prev_i, prev_t, k = [], rpc.prev_log_index, rpc.prev_log_term, 0
    if not ents:
        return [rpc]
    while k < len(ents):
        j, runs = k, 0
        while j < len(ents):
            if j == k or ents[j][1] != ents[j - 1][1]:
                runs += 1
                if runs == 3:
                    break
            j += 1
        out.append(AppendEntriesRpc(rpc.term, rpc.leader_id, rpc.leader_commit, prev_i, prev_t, tuple(ents[k:j])))
        prev_i, prev_t = ents[j - 1]
        k = j
    return out


def decode(msg, dec, rpcs, state_after, n_members: int) -> list:
    """effects() of one transition, in the reference's vocabulary.  `rpcs` = the rgb_rpc records whose
    msg_index is this message; `state_after` = the server's row after the transition (the leader's log
    supplies the entries of its rpcs, as ra_log does for make_append_entries_rpc/6)."""
    fl = int(dec["flags"])
    me = int(dec["server"]) % n_members
    to = int(dec["reply_to"])
    ok = bool(fl & abi.F_REPLY_SUCCESS)
    fx: list = []
    if fl & abi.F_INVARIANT:
        return [("exit", int(dec["invariant"]))]
    if fl & abi.F_REPLY:
        if fl & abi.F_REPLY_VOTE:
            fx.append(("reply", RequestVoteResult(int(dec["reply_term"]), ok)))
        elif fl & abi.F_REPLY_PRE_VOTE:
            fx.append(("reply", PreVoteResult(int(dec["reply_term"]), int(dec["reply_next_index"]), ok)))
        elif fl & abi.F_REPLY_HEARTBEAT:
            fx.append(("cast", to, (me, HeartbeatReply(int(dec["reply_next_index"]), int(dec["reply_term"])))))
        else:
            fx.append(("cast", to, (me, AppendEntriesReply(int(dec["reply_term"]), ok, int(dec["reply_next_index"]),
                                                          int(dec["reply_last_index"]), int(dec["reply_last_term"])))))
    if fl & abi.F_SEND_VOTE_REQUESTS:
        present = int(state_after["present_mask"])
        reqs = []
        for slot in range(n_members):
            if slot == me or not (present >> slot) & 1:
                continue
            if fl & abi.F_PRE_VOTE_REQS:
                reqs.append((slot, PreVoteRpc(int(dec["reply_term"]), int(dec["reply_next_index"]), me,
                                              int(dec["reply_last_index"]), int(dec["reply_last_term"]),
                                              int(state_after["machine_version"]))))
            else:
                reqs.append((slot, RequestVoteRpc(int(dec["reply_term"]), me, int(dec["reply_last_index"]),
                                                  int(dec["reply_last_term"]))))
        fx.append(("send_vote_requests", reqs))
    if len(rpcs):
        log = dict(abi.log_entries(state_after))
        for r in rpcs:
            if int(r["kind"]) == abi.RPC_SNAPSHOT:
                fx.append(("send_snapshot", int(r["peer"]), (int(r["prev_log_index"]), int(r["prev_log_term"]))))
            else:
                prev, n = int(r["prev_log_index"]), int(r["n_entries"])
                ents = tuple((prev + 1 + k, log[prev + 1 + k]) for k in range(n))
                fx.append(("send_rpc", int(r["peer"]),
                           AppendEntriesRpc(int(r["term"]), me, int(r["leader_commit"]), prev,
                                            int(r["prev_log_term"]), ents)))
    if fl & abi.F_CANCEL_SNAPSHOT_RETRY:                               # make_all_rpcs/1 :2356-2363
        for slot in range(n_members):
            if (int(dec["cancel_backoff"]) >> slot) & 1:
                fx.append(("cancel_snapshot_retry_timer", slot))
    if fl & abi.F_SEND_HEARTBEATS:
        for slot in range(n_members):
            if (int(dec["heartbeat_to"]) >> slot) & 1:
                fx.append(("send_rpc", slot, HeartbeatRpc(int(dec["reply_last_term"]), int(dec["reply_term"]), me)))
    if fl & abi.F_LEADER_MSG:
        fx.append(("record_leader_msg", int(state_after["leader_id"])))
    if fl & abi.F_PIPELINE:
        fx.append(("next_event", "info", PIPELINE_RPCS))
    if fl & abi.F_AUX_EVAL:
        fx.append(("aux", "eval"))
    if fl & abi.F_START_ELECTION_TIMEOUT:
        fx.append("start_election_timeout")
    if fl & abi.F_BECAME_LEADER:
        fx.append(("next_event", "cast", ("command", "noop")))      # post_election_effects/1
    if fl & abi.F_QUERY_QUORUM:
        fx.append(("query_quorum", int(dec["reply_next_index"])))   # release waiting queries up to this index
    if fl & abi.F_QUERY_APPLY:
        fx.append(("query_apply",))
    if fl & abi.F_RESEND_PENDING:
        fx.append(("resend_pending",))
    return fx
