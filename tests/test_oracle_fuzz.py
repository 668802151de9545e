"""CPU-only sanity of the checker itself on random inputs: no crashes, determinism, and the
state export/import round trip is the identity."""
import numpy as np
import pytest

import fuzz
from ra_amd import abi


@pytest.mark.parametrize("n_members,seed", [(3, 1), (5, 2), (7, 3), (8, 4), (1, 5)])
def test_oracle_fuzz_deterministic_and_roundtrip(oracle_lib, n_members, seed):
    rng = np.random.default_rng(seed)
    G = 120
    st = fuzz.random_states(rng, G, n_members)
    a, b = oracle_lib.Oracle(G, n_members), oracle_lib.Oracle(G, n_members)
    a.set_state(0, st)
    got = a.get_state()
    assert got.tobytes() == st.tobytes(), "set_state/get_state is not the identity"
    b.set_state(0, st)
    for _ in range(6):
        before = a.get_state()
        msgs = fuzz.random_msgs(rng, before, n_members)
        da, ra = a.step(msgs)
        db, rb = b.step(msgs)
        assert da.tobytes() == db.tobytes() and ra.tobytes() == rb.tobytes()
        sa = a.get_state()
        assert sa.tobytes() == b.get_state().tobytes()
        # structural invariants of any reachable state
        nonempty = sa["first_index"] <= sa["last_index"]
        assert np.all(sa["n_runs"][nonempty] >= 1)
        assert np.all(sa["last_applied"] <= np.maximum(sa["last_index"], sa["last_applied"]))
        inv = (da["flags"] & abi.F_INVARIANT) != 0
        assert np.all(da["invariant"][inv] > 0) and np.all(da["invariant"][~inv] == 0)
        # an exit/assert of the reference leaves the server exactly as it was
        for srv in msgs["server"][inv]:
            assert sa[srv].tobytes() == before[srv].tobytes(), f"server {srv} changed on invariant"


def test_parallel_step_equals_sequential_on_a_tick(oracle_lib):
    rng = np.random.default_rng(11)
    G, N = 200, 5
    st = fuzz.random_states(rng, G, N)
    a, b = oracle_lib.Oracle(G, N), oracle_lib.Oracle(G, N)
    a.set_state(0, st)
    b.set_state(0, st)
    msgs = fuzz.random_msgs(rng, st, N)
    da, _ = a.step(msgs)
    db, _ = b.step_parallel(msgs, 4)
    assert da.tobytes() == db.tobytes()
    assert a.get_state().tobytes() == b.get_state().tobytes()
