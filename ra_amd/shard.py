"""Sharding of Raft groups over the GPUs of one node.

Groups are independent units: gpu = splitmix64(group_uid) mod n_gpus, all members of a group
that live on this node stay together, and the decision path needs NO collective.  The only
cross-GPU exchange is the ra_leaderboard / key_metrics snapshot (reference
src/ra_leaderboard.erl:18-26, src/ra.erl:1242-1250): every rank produces one 32-byte row per
local group and the rows are all-gathered (RCCL over xGMI via torch.distributed backend
"nccl"; "gloo" on CPU for tests)."""
from __future__ import annotations

import numpy as np

from . import abi
from .workload import splitmix64


def owner(group_uids: np.ndarray, n_shards: int) -> np.ndarray:
    """Shard (GPU / rank) that owns each group uid."""
    return (splitmix64(np.asarray(group_uids, dtype=np.uint64)) % np.uint64(n_shards)).astype(np.int64)


def local_group_ids(n_global_groups: int, n_shards: int, shard: int, per_rank: int | None = None) -> np.ndarray:
    """Global group uids owned by `shard`, ascending.  With per_rank set (weak-scaling bench:
    a fixed number of groups per GPU) the id space is extended until the shard owns exactly
    per_rank groups."""
    if per_rank is None:
        ids = np.arange(n_global_groups, dtype=np.uint64)
        return ids[owner(ids, n_shards) == shard]
    out = np.zeros(0, dtype=np.uint64)
    lo, step = 0, max(n_global_groups, 1024)
    while len(out) < per_rank:
        ids = np.arange(lo, lo + step, dtype=np.uint64)
        out = np.concatenate([out, ids[owner(ids, n_shards) == shard]])
        lo += step
    return out[:per_rank]


def route(msgs: np.ndarray, group_uid_of_msg: np.ndarray, n_shards: int):
    """Split a host batch by owning shard: per-shard index arrays into msgs, submission order kept inside a shard
    (what a multi-context host does with rgb_route() per message before rgb_submit on each context; the server
    field of a routed message is then rewritten to the owner's local numbering by the caller's uid -> local map)."""
    assert len(msgs) == len(group_uid_of_msg)
    own = owner(group_uid_of_msg, n_shards)
    return [np.flatnonzero(own == s) for s in range(n_shards)]


def all_gather_leaderboard(local_rows: np.ndarray, group_uids: np.ndarray, dist=None):
    """All-gather the per-rank leaderboard shards and return (uids, rows) sorted by group uid.
    `dist` is torch.distributed (already initialised) or None for a single process."""
    import torch
    rows = np.ascontiguousarray(local_rows, dtype=abi.LEADERBOARD_DTYPE)
    uids = np.ascontiguousarray(group_uids, dtype=np.uint64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        order = np.argsort(uids, kind="stable")
        return uids[order], rows[order]
    world = dist.get_world_size()
    n = torch.tensor([len(rows)], dtype=torch.int64)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, n)
    m = int(max(c.item() for c in counts))
    pad_r = np.zeros(m, dtype=abi.LEADERBOARD_DTYPE)
    pad_r[:len(rows)] = rows
    pad_u = np.zeros(m, dtype=np.uint64)
    pad_u[:len(uids)] = uids
    tr = torch.from_numpy(pad_r.view(np.uint8).reshape(-1).copy())
    tu = torch.from_numpy(pad_u.view(np.int64).copy())
    gr = [torch.empty_like(tr) for _ in range(world)]
    gu = [torch.empty_like(tu) for _ in range(world)]
    dist.all_gather(gr, tr)
    dist.all_gather(gu, tu)
    all_r = np.concatenate([g.numpy().view(abi.LEADERBOARD_DTYPE)[:int(c.item())] for g, c in zip(gr, counts)])
    all_u = np.concatenate([g.numpy().view(np.uint64)[:int(c.item())] for g, c in zip(gu, counts)])
    order = np.argsort(all_u, kind="stable")
    return all_u[order], all_r[order]


def leaderboard_rows_from_states(st: np.ndarray, n_members: int) -> np.ndarray:
    """Host restatement of the device leaderboard kernel (used by CPU tests of the sharded path)."""
    N = n_members
    G = len(st) // N
    ct = st["current_term"].reshape(G, N)
    ci = st["commit_index"].reshape(G, N)
    la = st["last_applied"].reshape(G, N)
    is_l = st["role"].reshape(G, N) == abi.ROLE_LEADER
    rows = np.zeros(G, dtype=abi.LEADERBOARD_DTYPE)
    rows["n_leaders"] = is_l.sum(axis=1)
    rows["term"] = ct.max(axis=1)
    key = np.where(is_l, ct.astype(np.int64) + 1, 0)
    # first member with the highest term among leaders (the kernel keeps the first on ties)
    lead = key.argmax(axis=1)
    has = is_l.any(axis=1)
    r = np.arange(G)
    rows["leader"] = np.where(has, lead, abi.NONE)
    rows["commit_index"] = np.where(has, ci[r, lead], ci.max(axis=1))
    rows["last_applied"] = np.where(has, la[r, lead], la.max(axis=1))
    return rows
