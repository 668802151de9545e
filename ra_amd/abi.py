"""numpy mirrors of the C-ABI structs and constants in include/ra_gpu_batch.h.

Every array that crosses the boundary is a numpy structured array whose dtype below has the
exact C layout (verified against rgb_struct_size() when the library loads).  Field meanings
follow the reference's wire records (reference src/ra.hrl:123-154) -- see the header.
"""
from __future__ import annotations

import numpy as np

ABI_VERSION = 9
CFG_ROUNDS_PER_LAUNCH = 1      # rgb_config.flags: rgb_submit launches one kernel per sub-tick round (the default since round 5)
CFG_SUBMIT_TRAINS = 16         # rgb_config.flags: opt-in -- rgb_submit fuses the sub-tick rounds of a batch into one train launch
CFG_FUSE_PIPELINE = 4          # rgb_config.flags: a leader's success reply / written event carries its pipeline_rpcs event's rpcs (opt-in)
CFG_TRAIN_PERSISTENT = 2       # rgb_config.flags: trains always in the persistent form (placement by construction)
UNDEF = np.uint64(0xFFFFFFFFFFFFFFFF)  # Erlang 'undefined'
UNDEF_INT = 0xFFFFFFFFFFFFFFFF
NONE = 0xFF                            # undefined member slot
MAX_MEMBERS = 8
MAX_RUNS = 16
AER_CHUNK_SIZE = 128
DEFAULT_MAX_PIPELINE_COUNT = 4096

# ra_state()
ROLE_FOLLOWER, ROLE_CANDIDATE, ROLE_LEADER, ROLE_PRE_VOTE, ROLE_AWAIT_CONDITION = range(5)
ROLE_NAMES = ["follower", "candidate", "leader", "pre_vote", "await_condition"]
COND_NONE, COND_MISSING, COND_TERM_MISMATCH, COND_WAL_DOWN, COND_WAL_DOWN_LEADER = range(5)

(MSG_NOP, MSG_AER, MSG_AER_REPLY, MSG_REQUEST_VOTE, MSG_VOTE_RESULT, MSG_WRITTEN,
 MSG_PIPELINE_RPCS, MSG_APPEND, MSG_AWAIT_TIMEOUT, MSG_ELECTION_TIMEOUT, MSG_PRE_VOTE_RPC,
 MSG_PRE_VOTE_RESULT, MSG_SNAPSHOT_WRITTEN, MSG_HEARTBEAT_RPC, MSG_HEARTBEAT_REPLY,
 MSG_CONSISTENT_QUERY) = range(16)
N_KINDS = 16
PROTO_VERSION = 1
# device order of a tick: clause family = (class rank of the kind, success flag); the four hot
# kinds first (each has a specialised kernel), the rest after (ra_amd/csrc/rgb_internal.h)
KIND_RANK = np.array([15, 0, 1, 5, 6, 2, 4, 3, 7, 8, 9, 10, 11, 12, 13, 14], dtype=np.int64)
MF_SUCCESS = 0x01
MF_FORCE = 0x02
MF_TICK = 0x04
MF_SEQ2 = 0x08
MF_CAN_WRITE = 0x10
MF_SEQX = 0x20       # WRITTEN (with MF_SEQ2): more than two ranges, the lower ones in the batch's range list (c, n_entries)

F_REPLY = 1 << 0
F_REPLY_SUCCESS = 1 << 1
F_REPLY_VOTE = 1 << 2
F_PERSIST = 1 << 3
F_LEADER_MSG = 1 << 4
F_LEADER_CHANGED = 1 << 5
F_APPLIED = 1 << 6
F_AUX_EVAL = 1 << 7
F_WROTE = 1 << 8
F_TRUNCATED = 1 << 9
F_PIPELINE = 1 << 10
F_REPROCESSED = 1 << 11
F_ROLE_CHANGED = 1 << 12
F_BECAME_LEADER = 1 << 13
F_UNHANDLED = 1 << 14
F_INVARIANT = 1 << 15
F_RUNS_OVERFLOW = 1 << 16
F_SEND_SNAPSHOT = 1 << 17
F_REPLY_PRE_VOTE = 1 << 18
F_START_ELECTION_TIMEOUT = 1 << 19
F_SEND_VOTE_REQUESTS = 1 << 20
F_PRE_VOTE_REQS = 1 << 21
F_RESEND_PENDING = 1 << 22
F_REPLY_HEARTBEAT = 1 << 23
F_SEND_HEARTBEATS = 1 << 24
F_QUERY_QUORUM = 1 << 25
F_QUERY_APPLY = 1 << 26
F_CANCEL_SNAPSHOT_RETRY = 1 << 27
F_TRANSFER_LEADERSHIP = 1 << 29   # leader's wal_down condition timed out: {transfer_leadership, Peer} (src/ra_server.erl:660-668)
F_COMPACT = 1 << 28   # device-resident decision streams: the 32-byte compact form (expand_decisions)

(INV_NONE, INV_LEADER_SAW_AER_SAME_TERM, INV_TRUNCATE_BELOW_APPLIED, INV_WRITE_BELOW_APPLIED,
 INV_MISMATCH_TERM_UNDEFINED, INV_WRITE_INTEGRITY, INV_SET_LAST_INDEX_NOT_FOUND,
 INV_LAST_WRITTEN_TERM, INV_NEXT_INDEX_REGRESSED, INV_PIPELINE_PREV_UNDEFINED,
 INV_WRITTEN_NOT_PREFIX, INV_LEADER_SAW_HEARTBEAT_SAME_TERM) = range(12)

RPC_AER, RPC_SNAPSHOT = 1, 2

OK, E_INVAL, E_NOMEM, E_HIP, E_STATE, E_FULL, E_EMPTY, E_UNSUPPORTED, E_NODEVICE, E_COMM = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9)
COMM_ID_BYTES = 128

u8, u16, u32, u64, i32 = np.uint8, np.uint16, np.uint32, np.uint64, np.int32

MSG_DTYPE = np.dtype([
    ("server", u32), ("kind", u8), ("from", u8), ("flags", u8), ("gap", u8),
    ("term", u64), ("a", u64), ("b", u64), ("c", u64),
    ("n_entries", u32), ("n_run0", u32), ("run0_term", u64), ("run1_term", u64),
])

DECISION_DTYPE = np.dtype([
    ("server", u32), ("role", u8), ("reply_to", u8), ("n_rpcs", u8), ("kind", u8),
    ("flags", u32), ("invariant", u16), ("heartbeat_to", u8), ("cancel_backoff", u8),
    ("reply_term", u64), ("reply_next_index", u64), ("reply_last_index", u64),
    ("reply_last_term", u64), ("commit_index", u64), ("last_applied", u64),
])

RPC_DTYPE = np.dtype([
    ("msg_index", u32), ("server", u32), ("peer", u8), ("kind", u8), ("n_entries", u16),
    ("_pad", u32),
    ("term", u64), ("prev_log_index", u64), ("prev_log_term", u64), ("leader_commit", u64),
    ("next_index", u64),
])

SERVER_STATE_DTYPE = np.dtype([
    ("current_term", u64), ("commit_index", u64), ("last_applied", u64),
    ("last_index", u64), ("last_term", u64),
    ("last_written_index", u64), ("last_written_term", u64),
    ("snapshot_index", u64), ("snapshot_term", u64), ("first_index", u64),
    ("cond_reply", u64, (4,)),
    ("match_index", u64, (MAX_MEMBERS,)), ("next_index", u64, (MAX_MEMBERS,)),
    ("commit_index_sent", u64, (MAX_MEMBERS,)),
    ("run_start", u64, (MAX_RUNS,)), ("run_term", u64, (MAX_RUNS,)),
    ("role", u8), ("cond_reason", u8), ("self", u8), ("n_members", u8),
    ("voted_for", u8), ("leader_id", u8), ("votes", u8), ("n_runs", u8),
    ("present_mask", u8), ("voter_mask", u8), ("status_mask", u8), ("self_nonvoter", u8),
    ("cond_leader", u8), ("backoff_mask", u8), ("n_pending_old", u8), ("_pad", u8, (1,)),
    ("pre_vote_token", u64), ("query_index", u64), ("peer_query_index", u64, (MAX_MEMBERS,)),
    ("pending_first", u64),
    ("machine_version", u32), ("effective_machine_version", u32),
    ("pending_old", u64, (2, 2)),
])

LEADERBOARD_DTYPE = np.dtype([
    ("leader", u32), ("n_leaders", u32), ("term", u64), ("commit_index", u64),
    ("last_applied", u64),
])

CONFIG_DTYPE = np.dtype([
    ("abi_version", u32), ("device", i32), ("max_runs", u32), ("ring_slots", u32),
    ("ring_capacity", u32), ("max_pipeline_count", u32), ("max_aer_batch", u32), ("flags", u32),
])

# include/ra_gpu_wal.h
WAL_ENTRY_DTYPE = np.dtype([("index", u64), ("term", u64), ("data_offset", u64), ("data_len", u32),
                            ("_pad", u32)])
assert WAL_ENTRY_DTYPE.itemsize == 32
WAL_RECORD_DTYPE = np.dtype([("index", u64), ("term", u64), ("data_offset", u64), ("data_len", u32),
                             ("hdr_len", u32), ("hdr_offset", u64), ("out_offset", u64)])
assert WAL_RECORD_DTYPE.itemsize == 48
WAL_SCANNED_DTYPE = np.dtype([("index", u64), ("term", u64), ("data_offset", u64), ("data_len", u32),
                              ("checksum", u32), ("uid_offset", u64), ("id_ref", u32), ("uid_len", np.uint16),
                              ("trunc", np.uint8), ("flags", np.uint8), ("next_offset", u64)])
assert WAL_SCANNED_DTYPE.itemsize == 56
WAL_NO_CHECKSUMS = 1
WAL_REC_FIRST, WAL_REC_VALIDATE, WAL_REC_UNKNOWN = 1, 2, 4
WAL_END_ZEROS, WAL_END_DATA, WAL_END_CAP = 0, 1, 2
WAL_CLEAN, WAL_DROPPED_LAST, WAL_CORRUPT = 0, 1, 2
WAL_FILE_HEADER = b"RAWA\x01"

# rgb_view (ABI v9, rgb_collect_view): pointers into the pinned slot the device wrote
VIEW_DTYPE = np.dtype([("decisions", "<u8"), ("rpcs", "<u8"), ("tick", "<u8"), ("n", "<u4"), ("n_rpcs", "<u4"),
                       ("slot", "<u4"), ("_pad", "<u4")])
STRUCT_DTYPES = [MSG_DTYPE, DECISION_DTYPE, RPC_DTYPE, SERVER_STATE_DTYPE, LEADERBOARD_DTYPE,
                 CONFIG_DTYPE, VIEW_DTYPE]
EXPECTED_SIZES = [64, 64, 56, 704, 32, 32, 40]
for _dt, _sz in zip(STRUCT_DTYPES, EXPECTED_SIZES):
    assert _dt.itemsize == _sz, (_dt, _dt.itemsize, _sz)


def family(msgs: np.ndarray) -> np.ndarray:
    return KIND_RANK[msgs["kind"]] * 2 + (msgs["flags"] & MF_SUCCESS)


def empty_server_states(n_groups: int, n_members: int) -> np.ndarray:
    """ra_server:init/1 on an empty log for every member of every group: the `empty_state/2`
    fixture of the reference's tests (test/ra_server_SUITE.erl:4139-4149): term 0, log [0:0]
    written, every peer new_peer/0 (src/ra_server.erl:2990-2995)."""
    n = n_groups * n_members
    st = np.zeros(n, dtype=SERVER_STATE_DTYPE)
    st["snapshot_index"] = UNDEF
    st["snapshot_term"] = UNDEF
    st["next_index"][:, :n_members] = 1   # slots beyond n_members stay 0 (canonical form)
    st["pending_first"] = 1               # nothing pending: last_index + 1
    st["role"] = ROLE_FOLLOWER
    st["self"] = np.arange(n, dtype=np.uint32) % n_members
    st["n_members"] = n_members
    st["voted_for"] = NONE
    st["leader_id"] = NONE
    st["cond_leader"] = NONE
    st["n_runs"] = 1  # run 0 = (start 0, term 0)
    st["present_mask"] = (1 << n_members) - 1
    st["voter_mask"] = (1 << n_members) - 1
    st["status_mask"] = 0xFF
    return st


def set_log(st: np.ndarray, i: int, entries, last_written=None, snapshot=None, first_index=None):
    """Fill server i's log cursors from an explicit [(index, term), ...] list (ascending,
    contiguous), the notation of the reference's tests: `[{1,1},{2,3},{3,5}]`."""
    s = st[i:i + 1]
    if snapshot is not None:
        s["snapshot_index"], s["snapshot_term"] = snapshot
    entries = list(entries)
    if entries:
        fi = entries[0][0] if first_index is None else first_index
        s["first_index"] = fi
        s["last_index"], s["last_term"] = entries[-1]
        runs = []
        prev = None
        for k, (idx, term) in enumerate(entries):
            assert idx == entries[0][0] + k, "log entries must be contiguous"
            if prev is None or term != prev:
                runs.append((idx, term))
                prev = term
        assert len(runs) <= MAX_RUNS
        s["n_runs"] = len(runs)
        s["run_start"] = 0
        s["run_term"] = 0
        for r, (a, t) in enumerate(runs):
            st["run_start"][i, r] = a
            st["run_term"][i, r] = t
    else:
        assert snapshot is not None, "an empty range needs a snapshot"
        s["last_index"], s["last_term"] = snapshot
        s["first_index"] = snapshot[0] + 1
        s["n_runs"] = 0
    if last_written is None:
        last_written = (int(s["last_index"][0]), int(s["last_term"][0]))
    s["last_written_index"], s["last_written_term"] = last_written
    # ra_log `pending`: the unwritten tail is what the WAL still owes (empty = last_index + 1)
    s["pending_first"] = min(int(last_written[0]), int(s["last_index"][0])) + 1


def log_entries(st_row) -> list:
    """Inverse of set_log for one state row: [(index, term), ...] of the ra_log range."""
    out = []
    fi, li = int(st_row["first_index"]), int(st_row["last_index"])
    if fi > li:
        return out
    nr = int(st_row["n_runs"])
    for r in range(nr):
        a = int(st_row["run_start"][r])
        b = int(st_row["run_start"][r + 1]) - 1 if r + 1 < nr else li
        t = int(st_row["run_term"][r])
        out.extend((i, t) for i in range(a, b + 1))
    return out


def expand_decisions(dec: np.ndarray) -> np.ndarray:
    """rgb_decision_expand (include/ra_gpu_batch.h) over an array read from a DEVICE-RESIDENT decision stream: records
    with F_COMPACT were written as 32 bytes; the full records come back (a copy), the flag cleared."""
    d = np.array(dec, dtype=DECISION_DTYPE, copy=True)
    c = (d["flags"] & F_COMPACT) != 0
    if not c.any():
        return d
    w = d.view(np.uint64).reshape(-1, 8)
    aux = (w[:, 1] >> np.uint64(32)).astype(np.uint64)
    A, B = w[:, 2].copy(), w[:, 3].copy()
    flags = d["flags"] & ~np.uint32(F_COMPACT)
    reply = c & ((flags & F_REPLY) != 0)
    wrote = c & ~reply & ((flags & F_WROTE) != 0)
    counted = c & ~reply & ~wrote
    with np.errstate(over="ignore"):
        d["flags"] = flags
        for f in ("invariant", "heartbeat_to", "cancel_backoff"):
            d[f][c] = 0
        for f in ("reply_term", "reply_next_index", "reply_last_index", "reply_last_term"):
            d[f][c] = 0
        u = np.uint64
        d["reply_next_index"][reply] = (A + u(1))[reply]
        d["reply_term"][reply] = B[reply]
        d["reply_last_index"][reply] = (A - (aux & u(0xFF)))[reply]
        d["reply_last_term"][reply] = (B - ((aux >> u(8)) & u(0xF)))[reply]
        d["commit_index"][reply] = (A + ((aux >> u(12)) & u(0x3FF)) - u(512))[reply]
        d["last_applied"][reply] = (A + u(1) - ((aux >> u(22)) & u(0x3FF)))[reply]
        d["reply_last_index"][wrote] = A[wrote]
        d["reply_next_index"][wrote] = (A - (aux & u(0xFFFF)))[wrote]
        d["commit_index"][wrote] = B[wrote]
        d["last_applied"][wrote] = (A - (aux >> u(16)))[wrote]
        d["commit_index"][counted] = A[counted]
        d["last_applied"][counted] = B[counted]
    return d
