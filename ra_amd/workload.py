"""Synthetic multi-Raft workload for ra_gpu_batch (the role ra_bench plays for Ra,
reference src/ra_bench.erl): seeded initial states for the BASELINE.json configurations and a
per-tick message synthesiser, vectorised numpy, deterministic (splitmix64, seeds
0x5EED0002/3/5 as in SURVEY.md section 8d).

A tick carries at most one message per server.  Each group gets one PRIMARY message drawn from
the configured mix (append_entries_reply ok / append_entries_rpc / append_entries_reply failed /
request_vote_rpc with term+1) built from the RECEIVER's current cursors, so the stream stays in
a healthy steady state, plus optional housekeeping messages to other members of the group
({commands,_} appends on the leader, {written,..} log events) that keep the logs moving.
Between ticks `heal()` plays the host: groups whose leader was deposed by term churn hold an
election (new leader = most up-to-date member, noop appended in the new term, peers
re-initialised, cf. src/ra_server.erl:1045-1061, 3234-3242) and long run tables are compacted
behind a snapshot at last_applied (release_cursor).  Both are plain state uploads.
"""
from __future__ import annotations

import numpy as np

from . import abi

U64 = np.uint64
MASK = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


class Draw:
    """Counter-based random stream: value(tick, stream, i) depends on nothing else."""

    def __init__(self, seed: int):
        self.seed = U64(seed & MASK)

    def u64(self, tick: int, stream: int, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            base = splitmix64(np.array([(int(self.seed) ^ (tick * 0x9E3779B1 + stream * 0x85EBCA77)) & MASK],
                                       dtype=np.uint64))[0]
            return splitmix64(base + np.arange(n, dtype=np.uint64) * U64(0xD1B54A32D192ED03))

    def ints(self, tick: int, stream: int, n: int, mod: int) -> np.ndarray:
        return (self.u64(tick, stream, n) % U64(mod)).astype(np.int64)

    def unit(self, tick: int, stream: int, n: int) -> np.ndarray:
        return (self.u64(tick, stream, n) >> U64(11)).astype(np.float64) * (1.0 / (1 << 53))


# ------------------------------------------------------------------ state helpers ----

def term_at(st: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """ra_server:fetch_term/2 (with snapshot fallback) for one index per server; -1 = undefined."""
    idx = np.asarray(idx, dtype=np.int64)
    first = st["first_index"].astype(np.int64)
    li = st["last_index"].astype(np.int64)
    nr = st["n_runs"].astype(np.int64)
    rs = st["run_start"].astype(np.int64)
    r = np.arange(abi.MAX_RUNS)[None, :]
    le = (rs <= idx[:, None]) & (r < nr[:, None])
    k = le.sum(axis=1) - 1
    valid = (first <= li) & (idx >= first) & (idx <= li) & (k >= 0)
    t = np.take_along_axis(st["run_term"].astype(np.int64), np.maximum(k, 0)[:, None], axis=1)[:, 0]
    si = st["snapshot_index"]
    snap = (si != abi.UNDEF) & (si.astype(np.int64) == idx)
    out = np.where(valid, t, np.where(snap, st["snapshot_term"].astype(np.int64), -1))
    return out


def agreed_commit_rows(vals: np.ndarray) -> np.ndarray:
    """agreed_commit/1 (src/ra_server.erl:3684-3688) per row of a [G, n] array."""
    s = -np.sort(-vals, axis=1)
    return s[:, vals.shape[1] // 2]


def initial_states(n_groups: int, n_members: int, seed: int, backlog: int = 0,
                   boundaries: tuple = (0, 0), window: int = 64) -> np.ndarray:
    """Steady-state groups: one leader per group at term t in [1,8], leader last index in
    [1000,2000), match_index = LI - U{0..16}, next_index = match+1+U{0..8}, commit = quorum
    median.  With backlog > 0 (config 5) every follower holds
    `backlog` uncommitted entries (last_index = last_applied + backlog) crossing
    boundaries[0]..boundaries[1] term changes."""
    G, N = n_groups, n_members
    d = Draw(seed)
    st = abi.empty_server_states(G, N)
    g = np.arange(G)
    leader = d.ints(0, 1, G, N)
    nb = boundaries[0] + (d.ints(0, 2, G, boundaries[1] - boundaries[0] + 1) if boundaries[1] > boundaries[0] else 0)
    nb = np.broadcast_to(nb, (G,)).astype(np.int64)
    term = 1 + nb + d.ints(0, 3, G, 8)
    li_l = 1000 + backlog + d.ints(0, 4, G, 1000)
    span = window + backlog                     # entries kept behind the leader's last index
    si = li_l - span
    first = si + 1
    # run table shared by the group: nb+1 older runs spread over the backlog, current term last
    cur_start = li_l - 40                       # the newest 40 entries carry the current term
    n_runs = nb + 2
    run_start = np.zeros((G, abi.MAX_RUNS), dtype=np.int64)
    run_term = np.zeros((G, abi.MAX_RUNS), dtype=np.int64)
    older = nb + 1
    for r in range(int(older.max()) if G else 0):
        use = r < older
        seg = (cur_start - first) // np.maximum(older, 1)
        run_start[:, r] = np.where(use, first + r * seg, 0)
        run_term[:, r] = np.where(use, np.maximum(term - older + r, 0), 0)
    rows = np.arange(G)
    run_start[rows, older] = cur_start
    run_term[rows, older] = term
    snap_term = run_term[:, 0]
    mi = li_l[:, None] - d.ints(0, 5, G * N, 17).reshape(G, N)
    ni = np.minimum(mi + 1 + d.ints(0, 6, G * N, 9).reshape(G, N), li_l[:, None] + 1)
    is_l = np.arange(N)[None, :] == leader[:, None]
    vals = np.where(is_l, li_l[:, None], mi)
    ci = agreed_commit_rows(vals)
    li = np.where(is_l, li_l[:, None], ni - 1)
    lwi = np.where(is_l, li_l[:, None], mi)
    fci = np.maximum(ci[:, None] - d.ints(0, 7, G * N, 3).reshape(G, N), 0)
    ci_m = np.where(is_l, ci[:, None], fci)
    la = np.minimum(ci_m, li)
    if backlog:
        la = np.where(is_l, la, li - backlog)
        ci_m = np.where(is_l, ci_m, la)
    S = G * N
    rep = lambda a: np.repeat(a, N)  # noqa: E731
    st["current_term"] = rep(term)
    st["commit_index"] = ci_m.reshape(S)
    st["last_applied"] = la.reshape(S)
    st["last_index"] = li.reshape(S)
    st["last_term"] = rep(term)
    st["last_written_index"] = lwi.reshape(S)
    st["pending_first"] = lwi.reshape(S) + 1      # the unwritten tail is what the WAL still owes
    st["snapshot_index"] = rep(si)
    st["snapshot_term"] = rep(snap_term)
    st["first_index"] = rep(first)
    st["run_start"] = np.repeat(run_start, N, axis=0)
    st["run_term"] = np.repeat(run_term, N, axis=0)
    st["n_runs"] = rep(n_runs)
    st["last_written_term"] = term_at(st, st["last_written_index"].astype(np.int64))
    st["role"] = np.where(is_l, abi.ROLE_LEADER, abi.ROLE_FOLLOWER).reshape(S)
    st["voted_for"] = rep(leader)
    st["leader_id"] = rep(leader)
    st["match_index"][:, :N] = np.repeat(np.where(is_l, li_l[:, None], mi), N, axis=0)
    st["next_index"][:, :N] = np.repeat(np.where(is_l, li_l[:, None] + 1, ni), N, axis=0)
    st["commit_index_sent"][:, :N] = rep(ci)[:, None]
    # only the leader's peer arrays mean anything; followers keep new_peer/0 defaults
    fl = ~is_l.reshape(S)
    st["match_index"][fl] = 0
    st["next_index"][fl] = 0
    st["next_index"][fl, :N] = 1
    st["commit_index_sent"][fl] = 0
    return st


# ------------------------------------------------------------------ host surgery ----

def _compact(st: np.ndarray, mask: np.ndarray):
    """release_cursor-style compaction: snapshot at last_applied, drop the runs behind it."""
    idx = np.flatnonzero(mask)
    if len(idx) == 0:
        return
    s = st[idx]
    la = s["last_applied"].astype(np.int64)
    li = s["last_index"].astype(np.int64)
    la = np.minimum(la, li)
    lat = term_at(s, la)
    ok = (lat >= 0) & (la + 1 > s["first_index"].astype(np.int64))
    idx, s, la, li, lat = idx[ok], s[ok], la[ok], li[ok], lat[ok]
    if len(idx) == 0:
        return
    nr = s["n_runs"].astype(np.int64)
    rs = s["run_start"].astype(np.int64)
    rt = s["run_term"].astype(np.int64)
    r = np.arange(abi.MAX_RUNS)[None, :]
    nxt = np.concatenate([rs[:, 1:], np.zeros((len(idx), 1), dtype=np.int64)], axis=1)
    run_end = np.where(r + 1 < nr[:, None], nxt - 1, li[:, None])
    first_new = la + 1
    keep = (r < nr[:, None]) & (run_end >= first_new[:, None])
    new_start = np.maximum(rs, first_new[:, None])
    order = np.argsort(~keep, axis=1, kind="stable")
    ks = np.take_along_axis(np.where(keep, new_start, 0), order, axis=1)
    kt = np.take_along_axis(np.where(keep, rt, 0), order, axis=1)
    s["run_start"] = ks
    s["run_term"] = kt
    s["n_runs"] = keep.sum(axis=1)
    s["snapshot_index"] = la
    s["snapshot_term"] = lat
    s["first_index"] = first_new
    low = s["last_written_index"].astype(np.int64) < la
    s["last_written_index"] = np.where(low, la, s["last_written_index"].astype(np.int64))
    s["last_written_term"] = np.where(low, lat, s["last_written_term"].astype(np.int64))
    s["pending_first"] = np.maximum(s["pending_first"].astype(np.int64), la + 1)   # ra_seq:floor(SnapIdx+1, Pend)
    st[idx] = s


def heal(st: np.ndarray, n_members: int, max_runs: int = 8) -> int:
    """Host-side maintenance between ticks (in place).  Returns the number of servers changed."""
    N = n_members
    G = len(st) // N
    before = st.copy()
    _compact(st, st["n_runs"] >= max_runs - 2)
    ct = st["current_term"].astype(np.int64).reshape(G, N)
    role = st["role"].reshape(G, N)
    is_l = role == abi.ROLE_LEADER
    lead_ct = np.where(is_l, ct, -1).max(axis=1)
    n_lead = is_l.sum(axis=1)
    churned = (n_lead != 1) | (ct.max(axis=1) > lead_ct)
    e = np.flatnonzero(churned)
    if len(e):
        rows = (e[:, None] * N + np.arange(N)[None, :])
        lt = st["last_term"].astype(np.int64)[rows]
        li = st["last_index"].astype(np.int64)[rows]
        key = lt * (1 << 40) + li
        w = key.argmax(axis=1)
        T = ct[e].max(axis=1) + 1
        flat = rows.reshape(-1)
        st["current_term"][flat] = np.repeat(T, N)
        st["voted_for"][flat] = np.repeat(w, N)
        st["leader_id"][flat] = np.repeat(w, N)
        st["role"][flat] = abi.ROLE_FOLLOWER
        st["cond_reason"][flat] = abi.COND_NONE
        st["votes"][flat] = 0
        st["status_mask"][flat] = 0xFF
        ws = e * N + w
        # make room for the noop's new run first
        _compact(st, np.isin(np.arange(len(st)), ws) & (st["n_runs"] >= max_runs - 1))
        st["role"][ws] = abi.ROLE_LEADER
        nli = st["last_index"][ws].astype(np.int64) + 1
        nr = st["n_runs"][ws].astype(np.int64)
        empty = st["first_index"][ws].astype(np.int64) > st["last_index"][ws].astype(np.int64)
        st["first_index"][ws] = np.where(empty, nli, st["first_index"][ws].astype(np.int64))
        st["run_start"][ws, nr] = nli
        st["run_term"][ws, nr] = T
        st["n_runs"][ws] = nr + 1
        st["last_index"][ws] = nli
        st["last_term"][ws] = T
        st["last_written_index"][ws] = nli
        st["last_written_term"][ws] = T
        st["pending_first"][ws] = nli + 1
        st["match_index"][ws, :N] = 0
        st["next_index"][ws, :N] = (nli + 1)[:, None]
        st["commit_index_sent"][ws, :N] = st["commit_index"][ws][:, None]
    return int((st.view(np.uint8).reshape(len(st), -1) != before.view(np.uint8).reshape(len(st), -1)).any(axis=1).sum())


# --------------------------------------------------------------- message synthesis ----

MIX_CONFIG3 = dict(reply_ok=0.70, aer=0.20, reply_fail=0.05, request_vote=0.05)
MIX_CONFIG2 = dict(reply_ok=0.50, aer=0.50, reply_fail=0.0, request_vote=0.0)
MIX_CONFIG5 = dict(reply_ok=0.10, aer=0.50, reply_fail=0.40, request_vote=0.0)


def gen_tick(st: np.ndarray, n_members: int, tick: int, seed: int, mix: dict = MIX_CONFIG3,
             groups_per_tick: int | None = None, housekeeping: bool = True,
             backlog_mode: bool = False) -> np.ndarray:
    """One tick of messages (at most one per server) synthesised from the current states."""
    N = n_members
    G = len(st) // N
    d = Draw(seed)
    g_all = np.arange(G)
    if groups_per_tick is not None and groups_per_tick < G:
        pick = np.argsort(d.u64(tick, 90, G), kind="stable")[:groups_per_tick]
        pick.sort()
    else:
        pick = g_all
    P = len(pick)
    f = lambda name: st[name].astype(np.int64).reshape(G, N)  # noqa: E731
    role = st["role"].reshape(G, N)
    ct, li, lt = f("current_term"), f("last_index"), f("last_term")
    lwi, la, ci, first = f("last_written_index"), f("last_applied"), f("commit_index"), f("first_index")
    is_l = role == abi.ROLE_LEADER
    lead = np.where(is_l, ct + 1, 0).argmax(axis=1)
    lead = lead[pick]
    rowsP = np.arange(P)
    L = pick * N + lead                                   # leader server ids
    off = 1 + d.ints(tick, 10, P, max(N - 1, 1))
    j = (lead + off) % N if N > 1 else lead               # a non-leader member
    J = pick * N + j
    u = d.unit(tick, 11, P)
    c1 = mix["reply_ok"]
    c2 = c1 + mix["aer"]
    c3 = c2 + mix["reply_fail"]
    k_ok, k_aer = u < c1, (u >= c1) & (u < c2)
    k_fail, k_vote = (u >= c2) & (u < c3), u >= c3
    if N == 1:
        k_aer = np.zeros(P, bool)
        k_vote = ~k_ok & ~k_fail

    m = np.zeros(P, dtype=abi.MSG_DTYPE)
    ct_l, li_l = ct[pick, lead], li[pick, lead]
    ci_l = ci[pick, lead]
    mi_lj = st["match_index"].astype(np.int64)[L, j]
    r1 = d.ints(tick, 12, P, 1 << 30)
    r2 = d.ints(tick, 13, P, 1 << 30)

    # --- append_entries_reply success -> leader (a5/a6)
    last = np.minimum(li_l, mi_lj + r1 % 5)
    nxt = np.minimum(last + 1 + r2 % 3, li_l + 1)
    stL = st[L]
    lt_last = term_at(stL, last)
    sel = k_ok
    m["server"][sel] = L[sel]
    m["kind"][sel] = abi.MSG_AER_REPLY
    m["from"][sel] = j[sel]
    m["flags"][sel] = abi.MF_SUCCESS
    m["term"][sel] = ct_l[sel]
    m["a"][sel], m["b"][sel], m["c"][sel] = nxt[sel], last[sel], np.maximum(lt_last[sel], 0)

    # --- append_entries_reply failure -> leader (a8 repair)
    firstL = first[pick, lead]
    siL = stL["snapshot_index"].astype(np.int64)
    lo = np.maximum(firstL - 1, 0)
    lastf = np.maximum(mi_lj - r1 % 4, lo)
    tf = term_at(stL, lastf)
    wrong = (r2 & 1) == 1
    tf = np.where(wrong | (tf < 0), np.maximum(tf, 0) + 1, tf)
    sel = k_fail
    m["server"][sel] = L[sel]
    m["kind"][sel] = abi.MSG_AER_REPLY
    m["from"][sel] = j[sel]
    m["term"][sel] = ct_l[sel]
    m["a"][sel], m["b"][sel], m["c"][sel] = (lastf + 1)[sel], lastf[sel], tf[sel]
    _ = siL

    # --- append_entries_rpc -> follower (a2/a3)
    stJ = st[J]
    li_j, lt_j, la_j = li[pick, j], lt[pick, j], la[pick, j]
    first_j = first[pick, j]
    v = r1 % 100
    eterm = np.maximum(ct_l, lt_j)
    prev = li_j.copy()
    pterm = lt_j.copy()
    n_ent = 1 + (r2 % 8)
    run0 = eterm.copy()
    if backlog_mode:
        # prev_log_index inside the uncommitted backlog, prev_log_term wrong half of the time
        spanb = np.maximum(li_j - la_j, 1)
        prev = la_j + 1 + (r2 % spanb)
        prev = np.minimum(prev, li_j)
        pterm = term_at(stJ, prev)
        bad = v < 50
        pterm = np.where(bad | (pterm < 0), np.maximum(pterm, 0) + 1, pterm)
        n_ent = np.where(bad, 0, np.minimum(r1 % 5, li_j - prev))
        run0 = term_at(stJ, np.minimum(prev + 1, li_j))
        e_end = term_at(stJ, np.minimum(prev + np.maximum(n_ent, 1), li_j))
        n_ent = np.where(run0 != e_end, 0, n_ent)          # keep the overlap inside one run
        run0 = np.maximum(run0, 0)
    else:
        hb = (v >= 80) & (v < 85)
        miss = (v >= 85) & (v < 90)
        mism = (v >= 90) & (v < 95)
        ovl = v >= 95
        n_ent = np.where(hb | mism, 0, n_ent)
        prev = np.where(miss, li_j + 1 + r2 % 3, prev)
        pterm = np.where(miss, ct_l, pterm)
        n_ent = np.where(miss, r2 % 3, n_ent)
        pterm = np.where(mism, lt_j + 1, pterm)
        # overlap resend inside the last run
        lrs_j = np.take_along_axis(stJ["run_start"].astype(np.int64),
                                   np.maximum(stJ["n_runs"].astype(np.int64) - 1, 0)[:, None], axis=1)[:, 0]
        back = 1 + r2 % 4
        can = ovl & (li_j - back >= np.maximum(lrs_j, np.maximum(la_j, first_j))) & (first_j <= li_j)
        prev = np.where(can, li_j - back, prev)
        pterm = np.where(can, lt_j, pterm)
        n_ent = np.where(can, back, n_ent)
        run0 = np.where(can, lt_j, run0)
    sel = k_aer
    m["server"][sel] = J[sel]
    m["kind"][sel] = abi.MSG_AER
    m["from"][sel] = lead[sel]
    m["term"][sel] = ct_l[sel]
    m["a"][sel], m["b"][sel], m["c"][sel] = prev[sel], np.maximum(pterm[sel], 0), ci_l[sel]
    m["n_entries"][sel] = n_ent[sel]
    m["n_run0"][sel] = n_ent[sel]
    m["run0_term"][sel] = run0[sel]

    # --- request_vote_rpc with term+1 -> any member (a10, term churn)
    k = d.ints(tick, 14, P, N)
    cand = (k + 1 + d.ints(tick, 15, P, max(N - 1, 1))) % N if N > 1 else k
    K = pick * N + k
    sel = k_vote
    m["server"][sel] = K[sel]
    m["kind"][sel] = abi.MSG_REQUEST_VOTE
    m["from"][sel] = cand[sel]
    m["term"][sel] = (ct[pick, k] + 1)[sel]
    m["a"][sel] = np.maximum(li[pick, k] + (r2 % 3) - 1, 0)[sel]
    m["b"][sel] = lt[pick, k][sel]

    out = [m]
    if housekeeping:
        targeted = np.zeros(G * N, dtype=bool)
        targeted[m["server"]] = True
        # {commands,_}: the leader appends 1..4 entries (also pipelines rpcs)
        ap = (~targeted[L]) & (d.unit(tick, 20, P) < 0.5)
        a = np.zeros(int(ap.sum()), dtype=abi.MSG_DTYPE)
        a["server"] = L[ap]
        a["kind"] = abi.MSG_APPEND
        a["n_entries"] = 1 + d.ints(tick, 21, P, 4)[ap]
        targeted[L[ap]] = True
        out.append(a)
        # {ra_log_event,{written,Term,[LW+1..LI]}} for members with unwritten entries
        members = (pick[:, None] * N + np.arange(N)[None, :]).reshape(-1)
        need = (~targeted[members]) & (st["last_written_index"][members] < st["last_index"][members]) & \
               (st["first_index"][members] <= st["last_index"][members]) & \
               (d.unit(tick, 22, len(members)) < 0.5)
        ws = members[need]
        w = np.zeros(len(ws), dtype=abi.MSG_DTYPE)
        w["server"] = ws
        w["kind"] = abi.MSG_WRITTEN
        w["term"] = st["last_term"][ws]
        w["a"] = np.maximum(st["last_written_index"][ws] + U64(1), st["first_index"][ws])
        w["b"] = st["last_index"][ws]
        out.append(w)
    msgs = np.concatenate(out)
    # a tick holds at most one message per server, so its order is free: group by kind so that
    # wavefronts run one clause family = (kind, success flag), as rgb_submit does for host batches
    return msgs[np.argsort(abi.family(msgs), kind="stable")]


def pad_tick(msgs: np.ndarray, width: int) -> np.ndarray:
    """NOP-pad a tick to a fixed width (device-resident tick streams are dense)."""
    assert len(msgs) <= width
    out = np.zeros(width, dtype=abi.MSG_DTYPE)
    out[:len(msgs)] = msgs
    return out


# algorithmic bytes per decision, SURVEY.md section 8(d) (u64 fields)
def algorithmic_bytes(msgs: np.ndarray, n_members: int) -> int:
    k = msgs["kind"]
    n_aer = int((k == abi.MSG_AER).sum())
    n_rep = int((k == abi.MSG_AER_REPLY).sum())
    n_vote = int(np.isin(k, [abi.MSG_REQUEST_VOTE, abi.MSG_VOTE_RESULT, abi.MSG_ELECTION_TIMEOUT,
                             abi.MSG_PRE_VOTE_RPC, abi.MSG_PRE_VOTE_RESULT, abi.MSG_HEARTBEAT_RPC,
                             abi.MSG_HEARTBEAT_REPLY, abi.MSG_CONSISTENT_QUERY]).sum())
    # housekeeping kinds are priced like the class they resemble: written/await_timeout touch the
    # follower cursor like a vote (112 B); append/pipeline_rpcs walk the peer arrays like a reply
    n_small = int(np.isin(k, [abi.MSG_WRITTEN, abi.MSG_AWAIT_TIMEOUT, abi.MSG_SNAPSHOT_WRITTEN]).sum())
    n_peer = int(np.isin(k, [abi.MSG_APPEND, abi.MSG_PIPELINE_RPCS]).sum())
    return 256 * n_aer + (168 + 16 * n_members) * (n_rep + n_peer) + 112 * (n_vote + n_small)


def algorithmic_bytes_from_counts(kind_counts: np.ndarray, n_members: int) -> np.ndarray:
    """Same pricing as algorithmic_bytes() from per-kind message counts [..., n_kinds]."""
    kc = np.asarray(kind_counts, dtype=np.int64)
    rep = 168 + 16 * n_members
    price = np.zeros(kc.shape[-1], dtype=np.int64)
    price[abi.MSG_AER] = 256
    for k in (abi.MSG_AER_REPLY, abi.MSG_APPEND, abi.MSG_PIPELINE_RPCS):
        price[k] = rep
    for k in (abi.MSG_REQUEST_VOTE, abi.MSG_VOTE_RESULT, abi.MSG_WRITTEN, abi.MSG_AWAIT_TIMEOUT,
              abi.MSG_ELECTION_TIMEOUT, abi.MSG_PRE_VOTE_RPC, abi.MSG_PRE_VOTE_RESULT,
              abi.MSG_SNAPSHOT_WRITTEN, abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY,
              abi.MSG_CONSISTENT_QUERY):
        price[k] = 112
    return (kc * price).sum(axis=-1)
