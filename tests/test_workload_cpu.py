"""The synthetic workload (ra_amd/workload.py) against the CPU checker: valid initial states for
every BASELINE configuration shape, at most one message per server per tick, family order, no
invariant breaches, healthy steady state."""
import numpy as np
import pytest

from ra_amd import abi, workload as W


@pytest.mark.parametrize("name,G,N,kw,mix,bm", [
    ("config2", 256, 5, {}, W.MIX_CONFIG2, False),
    ("config3", 512, 5, {}, W.MIX_CONFIG3, False),
    ("config5", 256, 7, dict(backlog=1024, boundaries=(3, 6)), W.MIX_CONFIG5, True),
    ("three", 128, 3, {}, W.MIX_CONFIG3, False),
])
def test_workload_is_consistent_and_healthy(oracle_lib, name, G, N, kw, mix, bm):
    seed = 0x5EED0003
    st = W.initial_states(G, N, seed, **kw)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    assert cpu.get_state().tobytes() == st.tobytes(), "initial states are not canonical"
    # the leader's commit index is the quorum median of its match indexes
    lead = st["role"] == abi.ROLE_LEADER
    assert lead.sum() == G
    seen = 0
    for t in range(8):
        cur = cpu.get_state()
        if W.heal(cur, N, max_runs=16):
            cpu.set_state(0, cur)
        m = W.gen_tick(cur, N, t, seed, mix, backlog_mode=bm,
                       groups_per_tick=64 if name == "config2" else None)
        assert len(np.unique(m["server"])) == len(m)
        assert np.all(np.diff(abi.family(m)) >= 0)
        d, _ = cpu.step(m)
        assert not np.any(d["flags"] & abi.F_INVARIANT)
        seen |= int(np.bitwise_or.reduce(d["flags"]))
        assert W.algorithmic_bytes(m, N) == int(W.algorithmic_bytes_from_counts(
            np.bincount(m["kind"], minlength=abi.N_KINDS), N))
    assert seen & abi.F_REPLY and seen & abi.F_APPLIED
    if bm:
        assert (cpu.get_state()["role"] == abi.ROLE_AWAIT_CONDITION).sum() > 0   # the repair path ran


def test_splitmix_and_term_at():
    x = W.splitmix64(np.array([0, 1, 2], dtype=np.uint64))
    assert x[0] == np.uint64(0xE220A8397B1DCDAF)          # published splitmix64 test vector
    st = abi.empty_server_states(1, 3)
    abi.set_log(st, 0, [(5, 2), (6, 2), (7, 4)], snapshot=(4, 1))
    assert W.term_at(st[:1], np.array([4]))[0] == 1       # snapshot fallback
    assert W.term_at(st[:1], np.array([6]))[0] == 2
    assert W.term_at(st[:1], np.array([7]))[0] == 4
    assert W.term_at(st[:1], np.array([8]))[0] == -1
