"""heartbeat_rpc_quorum/3 (src/ra_server.erl:3797-3832) restated the way the reference writes it --
a cluster map, lists:sort + lists:nth -- against the checker on random leaders: membership masks,
non-voters, unknown peers, stale and fresh query indexes."""
import numpy as np
import pytest

from ra_amd import abi


def model(query_index, peer_qi, present, voters, self_slot, n, frm, new_qi):
    cluster = {i: peer_qi[i] for i in range(n) if (present >> i) & 1}
    if frm in cluster and new_qi > cluster[frm]:            # update_peer_query_index/3
        cluster[frm] = new_qi
    idxs = [query_index] + [q for i, q in cluster.items() if i != self_slot and (voters >> i) & 1]
    idxs.sort(reverse=True)                                 # agreed_commit/1: nth = trunc(len/2) + 1
    return idxs[len(idxs) // 2], cluster


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8])
def test_consensus_query_index_matches_list_model(oracle_lib, n):
    rng = np.random.default_rng(50 + n)
    G = 200
    cpu = oracle_lib.Oracle(G, n)
    st = cpu.get_state()
    full = (1 << n) - 1
    for g in range(G):
        s = g * n + int(rng.integers(0, n))
        st["role"][s] = abi.ROLE_LEADER
        st["current_term"][s] = 5
        st["leader_id"][s] = s % n
        st["query_index"][s] = int(rng.integers(0, 6))
        st["peer_query_index"][s, :n] = rng.integers(0, 6, size=n)
        st["present_mask"][s] = full if rng.random() < 0.7 else (int(rng.integers(0, full + 1)) | (1 << (s % n)))
        st["voter_mask"][s] = full if rng.random() < 0.7 else (int(rng.integers(0, full + 1)) | (1 << (s % n)))
    cpu.set_state(0, st)
    leaders = np.flatnonzero(st["role"] == abi.ROLE_LEADER)
    for rep in range(4):
        cur = cpu.get_state()
        msgs = np.zeros(len(leaders), dtype=abi.MSG_DTYPE)
        msgs["server"] = leaders
        msgs["kind"] = abi.MSG_HEARTBEAT_REPLY
        msgs["term"] = 5
        msgs["from"] = [abi.NONE if rng.random() < 0.1 else int(rng.integers(0, n)) for _ in leaders]
        msgs["a"] = rng.integers(0, 8, size=len(leaders))
        dec, _ = cpu.step(msgs)
        after = cpu.get_state()
        for k, s in enumerate(leaders):
            row = cur[s]
            want, cluster = model(int(row["query_index"]), [int(x) for x in row["peer_query_index"]],
                                  int(row["present_mask"]), int(row["voter_mask"]), int(s % n), n,
                                  int(msgs["from"][k]), int(msgs["a"][k]))
            assert int(dec["flags"][k]) & abi.F_QUERY_QUORUM
            assert int(dec["reply_next_index"][k]) == want, (n, rep, k)
            for i, q in cluster.items():
                assert int(after["peer_query_index"][s, i]) == q
    cpu.close()
