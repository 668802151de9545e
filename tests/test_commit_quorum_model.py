"""handle_leader({Peer, #append_entries_reply{success = true}}) (src/ra_server.erl:532-571) with
evaluate_quorum/2, increment_commit_index/1, match_indexes/1, agreed_commit/1 (:3633-3688) and
apply_to (:3250-3282) restated the way the reference writes them -- a cluster map, lists:sort,
lists:nth -- against the checker.  Covers what no reference test pins:
6-, 7- and 8-member quorums, even memberships, non-voters, commit DEcrease after a match reset,
the section-5.4.2 term gate, stale-term replies, unknown peers."""
import numpy as np
import pytest

import fuzz
from ra_amd import abi


def term_at(row, idx):
    for i, t in abi.log_entries(row):
        if i == idx:
            return t
    if int(row["snapshot_index"]) == idx and int(row["snapshot_index"]) != abi.UNDEF:
        return int(row["snapshot_term"])
    return None


def model(row, n, frm, term, next_index, last_index):
    """-> (match_index, next_index, commit_index, last_applied) after the clause, or None if the
    message does not reach the success clause's body."""
    ct = int(row["current_term"])
    self_slot = int(row["self"])
    present, voters = int(row["present_mask"]), int(row["voter_mask"])
    mi = [int(x) for x in row["match_index"]]
    ni = [int(x) for x in row["next_index"]]
    ci, la = int(row["commit_index"]), int(row["last_applied"])
    if term != ct:
        return None                                           # other clauses
    if not (frm < n and (present >> frm) & 1):
        return mi, ni, ci, la                                 # unknown peer: {leader, State0, []}
    mi[frm] = max(mi[frm], last_index)
    ni[frm] = max(ni[frm], next_index)
    idxs = [int(row["last_written_index"])] + [mi[i] for i in range(n)
                                                if i != self_slot and (present >> i) & 1 and (voters >> i) & 1]
    idxs.sort(reverse=True)
    cand = idxs[len(idxs) // 2]                               # lists:nth(trunc(L/2)+1, Sorted)
    if term_at(row, cand) == ct:                              # section 5.4.2; plain assignment, no max()
        ci = cand
    if ci > la:                                               # apply_to: up to min(last_index, CI)
        la = max(la, min(int(row["last_index"]), ci))
    return mi, ni, ci, la


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])
def test_success_reply_commit_rule_matches_list_model(oracle_lib, n):
    rng = np.random.default_rng(700 + n)
    G = 400
    st = fuzz.random_states(rng, G, n, max_runs=6)
    lead = np.arange(G) * n + rng.integers(0, n, size=G)
    st["role"][lead] = abi.ROLE_LEADER
    st["self_nonvoter"][lead] = 0
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    checked = decreased = gated = 0
    for rep in range(6):
        cur = cpu.get_state()
        msgs = np.zeros(G, dtype=abi.MSG_DTYPE)
        msgs["server"] = lead
        msgs["kind"] = abi.MSG_AER_REPLY
        msgs["flags"] = abi.MF_SUCCESS
        for k, s in enumerate(lead):
            row = cur[s]
            li = int(row["last_index"])
            msgs["from"][k] = abi.NONE if rng.random() < 0.05 else int(rng.integers(0, n))
            msgs["term"][k] = max(0, int(row["current_term"]) - (1 if rng.random() < 0.1 else 0))
            last = max(0, li - int(rng.integers(0, 8)))
            msgs["b"][k] = last
            msgs["a"][k] = last + 1 + int(rng.integers(0, 3))
        dec, _ = cpu.step(msgs)
        after = cpu.get_state()
        for k, s in enumerate(lead):
            if int(cur["role"][s]) != abi.ROLE_LEADER:
                continue
            want = model(cur[s], n, int(msgs["from"][k]), int(msgs["term"][k]), int(msgs["a"][k]), int(msgs["b"][k]))
            if want is None:
                continue
            assert not (int(dec["flags"][k]) & abi.F_INVARIANT)
            mi, ni, ci, la = want
            row = after[s]
            assert [int(x) for x in row["match_index"]] == mi, (n, rep, k)
            assert [int(x) for x in row["next_index"]] == ni, (n, rep, k)
            assert (int(row["commit_index"]), int(row["last_applied"])) == (ci, la), (n, rep, k)
            assert (int(dec["commit_index"][k]), int(dec["last_applied"][k])) == (ci, la)
            checked += 1
            decreased += ci < int(cur["commit_index"][s])
            gated += ci == int(cur["commit_index"][s])
    assert checked > 1500 and decreased > 0 and gated > 0
    cpu.close()
