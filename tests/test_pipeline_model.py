"""make_pipelined_rpc_effects/3 (src/ra_server.erl:2285-2346), make_rpc_effect/5 (:2382-2416) and
make_append_entries_rpc/6 (:2418-2435) restated as the fold the reference writes, against the
checker: the max_pipeline_count clamp (no reference test pins it), batch sizes, commit_index_sent,
the snapshot branch, both assertions, the `pipeline_rpcs` info event and {commands,_} with and
without the noop's Force."""
import numpy as np
import pytest

import fuzz
from ra_amd import abi

MAX_PIPE, MAX_BATCH = 4096, 128          # src/ra_server.hrl:7-8


def log_term(row, idx):
    for i, t in abi.log_entries(row):
        if i == idx:
            return t
    return None


def model(row, n, force, li, ci, max_pipe, max_batch):
    """-> (next_index[], commit_index_sent[], rpcs{peer: (kind, prev, prev_term, n_entries)}, more) or 'crash'"""
    self_slot = int(row["self"])
    present, status = int(row["present_mask"]), int(row["status_mask"])
    ni = [int(x) for x in row["next_index"]]
    mi = [int(x) for x in row["match_index"]]
    cis = [int(x) for x in row["commit_index_sent"]]
    snap = None if int(row["snapshot_index"]) == abi.UNDEF else (int(row["snapshot_index"]), int(row["snapshot_term"]))
    next_log = li + 1
    rpcs, more = {}, False
    for p in range(n):
        if p == self_slot or not (present >> p) & 1 or not (status >> p) & 1:
            continue
        if not (ni[p] < next_log or cis[p] < ci):
            continue
        inflight = ni[p] - mi[p] - 1
        if not (inflight < max_pipe or force):
            continue
        batch = max(1, min(max_batch, max_pipe - inflight))
        prev = ni[p] - 1
        pt = log_term(row, prev) if prev <= li else None
        if pt is None and snap and snap[0] == prev:
            pt = snap[1]
        if pt is not None:
            to = min(li, prev + batch)
            new_ni = to + 1
            rpcs[p] = (abi.RPC_AER, prev, pt, max(0, to - prev))
        else:
            if snap is None or not (prev < snap[0]):
                return "crash"                                  # case_clause / ?assert(PrevIdx < SnapIdx)
            new_ni = snap[0]
            rpcs[p] = (abi.RPC_SNAPSHOT, snap[0], snap[1], 0)
        if not (new_ni >= ni[p]):
            return "crash"                                      # ?assert(NewNextIdx >= NextIdx)
        ni[p], cis[p] = new_ni, ci
        more = more or (new_ni < next_log and (new_ni - mi[p] - 1) < max_pipe)
    return ni, cis, rpcs, more


@pytest.mark.parametrize("n,max_pipe", [(3, MAX_PIPE), (5, MAX_PIPE), (7, MAX_PIPE), (5, 6), (3, 3)])
def test_pipelining_matches_fold_model(oracle_lib, n, max_pipe):
    """max_pipe 6 / 3: a small max_pipeline_count so that the clamp actually binds."""
    rng = np.random.default_rng(1200 + n + max_pipe)
    G = 400
    st = fuzz.random_states(rng, G, n, max_runs=6)
    lead = np.arange(G) * n + rng.integers(0, n, size=G)
    st["role"][lead] = abi.ROLE_LEADER
    cpu = oracle_lib.Oracle(G, n, max_pipeline_count=max_pipe, max_aer_batch=MAX_BATCH)
    cpu.set_state(0, st)
    checked = crashes = clamped = snaps = mores = 0
    for rep in range(5):
        cur = cpu.get_state()
        msgs = np.zeros(G, dtype=abi.MSG_DTYPE)
        msgs["server"] = lead
        kinds = rng.choice([abi.MSG_PIPELINE_RPCS, abi.MSG_APPEND], size=G)
        msgs["kind"] = kinds
        msgs["n_entries"] = np.where(kinds == abi.MSG_APPEND, rng.integers(1, 5, size=G), 0)
        msgs["flags"] = np.where((kinds == abi.MSG_APPEND) & (rng.random(G) < 0.3), abi.MF_FORCE, 0)
        dec, rpcs = cpu.step(msgs)
        after = cpu.get_state()
        by_msg = {}
        for r in rpcs:
            by_msg.setdefault(int(r["msg_index"]), {})[int(r["peer"])] = r
        for k, s in enumerate(lead):
            row = cur[s]
            if int(row["role"]) != abi.ROLE_LEADER:
                continue
            appended = int(msgs["n_entries"][k]) if kinds[k] == abi.MSG_APPEND else 0
            li = int(row["last_index"]) + appended
            probe = row.copy()
            if appended:                                        # the appended entries carry the current term
                ents = abi.log_entries(row) + [(int(row["last_index"]) + 1 + j, int(row["current_term"]))
                                               for j in range(appended)]
                tmp = cur[s:s + 1].copy()
                if len({t for _, t in ents}) > abi.MAX_RUNS:
                    continue
                try:
                    abi.set_log(tmp, 0, ents, last_written=(int(row["last_written_index"]), int(row["last_written_term"])),
                                snapshot=None if int(row["snapshot_index"]) == abi.UNDEF else
                                (int(row["snapshot_index"]), int(row["snapshot_term"])))
                except AssertionError:
                    continue
                probe = tmp[0]
            want = model(probe, n, bool(int(msgs["flags"][k]) & abi.MF_FORCE), li, int(row["commit_index"]),
                         max_pipe, MAX_BATCH)
            flags = int(dec["flags"][k])
            if want == "crash":
                assert flags & abi.F_INVARIANT, (n, rep, k)
                assert after[s].tobytes() == cur[s].tobytes()
                crashes += 1
                continue
            assert not (flags & abi.F_INVARIANT), (n, rep, k, int(dec["invariant"][k]))
            ni, cis, want_rpcs, more = want
            assert [int(x) for x in after["next_index"][s]] == ni, (n, rep, k)
            assert [int(x) for x in after["commit_index_sent"][s]] == cis, (n, rep, k)
            got = by_msg.get(k, {})
            assert sorted(got) == sorted(want_rpcs), (n, rep, k)
            for p, (kind, prev, pt, cnt) in want_rpcs.items():
                r = got[p]
                assert (int(r["kind"]), int(r["prev_log_index"]), int(r["prev_log_term"]), int(r["n_entries"])) == \
                    (kind, prev, pt, cnt), (n, rep, k, p)
                assert int(r["term"]) == int(row["current_term"]) and int(r["leader_commit"]) == int(row["commit_index"])
                snaps += kind == abi.RPC_SNAPSHOT
                clamped += cnt and cnt < min(MAX_BATCH, li - prev)
            if kinds[k] == abi.MSG_PIPELINE_RPCS:
                assert bool(flags & abi.F_PIPELINE) == more, (n, rep, k)   # {next_event, info, pipeline_rpcs} iff More
                mores += more
            checked += 1
    assert checked > 1200 and crashes > 0 and snaps > 0 and mores > 0
    if max_pipe < 100:
        assert clamped > 0, "the max_pipeline_count clamp never bound"
    cpu.close()
