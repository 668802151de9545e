"""Two more clause-by-clause restatements against the checker on random states:
 * the failed append_entries_reply repair of match_index / next_index
   (handle_leader, src/ra_server.erl:587-649; the pipelining that follows is not part of the model),
 * handle_follower(#request_vote_rpc{}) with its clause ORDER (:1483-1529) and
   is_candidate_log_up_to_date/3 (:3157-3166)."""
import numpy as np
import pytest

import fuzz
from ra_amd import abi


def log_term(row, idx):
    """ra_log:fetch_term/2: defined only inside the range (no snapshot fallback)."""
    for i, t in abi.log_entries(row):
        if i == idx:
            return t
    return None


@pytest.mark.parametrize("n", [3, 5, 7])
def test_failed_reply_repairs_cursors_like_the_case_expression(oracle_lib, n):
    rng = np.random.default_rng(900 + n)
    G = 500
    st = fuzz.random_states(rng, G, n, max_runs=6)
    lead = np.arange(G) * n + rng.integers(0, n, size=G)
    st["role"][lead] = abi.ROLE_LEADER
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    seen = set()
    for rep in range(5):
        cur = cpu.get_state()
        msgs = np.zeros(G, dtype=abi.MSG_DTYPE)
        msgs["server"] = lead
        msgs["kind"] = abi.MSG_AER_REPLY                      # flags 0: success = false
        exp = []
        for k, s in enumerate(lead):
            row = cur[s]
            frm = int(rng.integers(0, n))
            li = int(row["last_index"])
            mi, ni = int(row["match_index"][frm]), int(row["next_index"][frm])
            mode = rng.random()
            last = max(0, mi - int(rng.integers(1, 4))) if mode < 0.25 else max(0, li + int(rng.integers(-6, 3)))
            t = log_term(row, last)
            lterm = t if (t is not None and rng.random() < 0.5) else int(rng.integers(0, 8))
            nxt = last + 1 + int(rng.integers(0, 2))
            msgs["from"][k] = frm          # no term guard on this clause, but the higher-term clause precedes it
            msgs["term"][k] = int(rng.integers(0, int(row["current_term"]) + 1)) if rng.random() < 0.9 else \
                int(row["current_term"]) + 1
            msgs["a"][k], msgs["b"][k], msgs["c"][k] = nxt, last, lterm
            if int(row["role"]) != abi.ROLE_LEADER or not ((int(row["present_mask"]) >> frm) & 1):
                exp.append(None)
                continue
            if int(msgs["term"][k]) > int(row["current_term"]):
                exp.append("abdicate")                        # the higher-term clause comes first (:572-586)
                continue
            if t is None:
                want, tag = (mi, nxt), "undefined"
            elif t == lterm and last >= mi:
                want, tag = (last, nxt), "forward"
            elif last < mi:
                want, tag = (last, last + 1), "reset"
            else:
                want, tag = (mi, max(min(ni - 1, last), mi + 1)), "decrement"
            exp.append((frm, want, tag))
        dec, _ = cpu.step(msgs)
        after = cpu.get_state()
        for k, s in enumerate(lead):
            if exp[k] is None or int(dec["flags"][k]) & abi.F_INVARIANT:
                continue
            if exp[k] == "abdicate":
                assert int(after["role"][s]) == abi.ROLE_FOLLOWER
                assert int(after["current_term"][s]) == int(msgs["term"][k])
                seen.add("abdicate")
                continue
            frm, (mi, ni), tag = exp[k]
            assert int(after["match_index"][s, frm]) == mi, (n, rep, k, tag)
            # the pipelining that follows may move next_index forward, never below the repaired value
            assert int(after["next_index"][s, frm]) >= ni, (n, rep, k, tag)
            if int(dec["n_rpcs"][k]) == 0:
                assert int(after["next_index"][s, frm]) == ni, (n, rep, k, tag)
            seen.add(tag)
    assert seen == {"undefined", "forward", "reset", "decrement", "abdicate"}
    cpu.close()


@pytest.mark.parametrize("n", [3, 5])
def test_follower_request_vote_clause_order(oracle_lib, n):
    rng = np.random.default_rng(950 + n)
    G = 600
    st = fuzz.random_states(rng, G, n, max_runs=6)
    st["role"][:] = abi.ROLE_FOLLOWER
    st["cond_reason"][:] = abi.COND_NONE
    cpu = oracle_lib.Oracle(G, n)
    cpu.set_state(0, st)
    S = G * n
    seen = set()
    for rep in range(3):
        cur = cpu.get_state()
        msgs = np.zeros(S, dtype=abi.MSG_DTYPE)
        msgs["server"] = np.arange(S)
        msgs["kind"] = abi.MSG_REQUEST_VOTE
        exp = []
        for s in range(S):
            row = cur[s]
            ct, li, lt = int(row["current_term"]), int(row["last_index"]), int(row["last_term"])
            cand = int(rng.integers(0, n))
            term = max(0, ct + int(rng.integers(-1, 2)))
            lli, llt = max(0, li + int(rng.integers(-1, 2))), max(0, lt + int(rng.integers(-1, 2)))
            msgs["from"][s], msgs["term"][s], msgs["a"][s], msgs["b"][s] = cand, term, lli, llt
            vf = int(row["voted_for"])
            if int(row["self_nonvoter"]):
                exp.append(("nonvoter", None, None, ct, vf))
            elif term == ct and vf != abi.NONE and vf != cand:
                exp.append(("already_voted", term, False, ct, vf))
            elif term >= ct:
                vf1 = abi.NONE if term > ct else vf                      # update_term/2
                up = llt > lt or (llt == lt and lli >= li)
                exp.append(("grant" if up else "decline", term, up, term, cand if up else vf1))
            else:
                exp.append(("stale", ct, False, ct, vf))
        dec, _ = cpu.step(msgs)
        after = cpu.get_state()
        for s in range(S):
            tag, rterm, granted, ct2, vf2 = exp[s]
            f = int(dec["flags"][s])
            if tag == "nonvoter":
                assert not (f & abi.F_REPLY)
            else:
                assert f & abi.F_REPLY and f & abi.F_REPLY_VOTE, (n, rep, s, tag)
                assert int(dec["reply_term"][s]) == rterm and bool(f & abi.F_REPLY_SUCCESS) == granted, (n, rep, s, tag)
            assert int(after["current_term"][s]) == ct2 and int(after["voted_for"][s]) == vf2, (n, rep, s, tag)
            seen.add(tag)
    assert seen == {"nonvoter", "already_voted", "grant", "decline", "stale"}
    cpu.close()
