"""ra_amd/shell.py: the host loop around the engine with payloads and state machines on the host.
Here the checker stands in for the engine (same step/get_state interface); on an MI355X the same shell
drives ra_amd.engine.RaGpuBatch (examples/kv_cluster.py)."""
import numpy as np
import pytest

from ra_amd import abi
from ra_amd.shell import RaShell, KvMachine, NOOP


@pytest.mark.parametrize("n_members", [1, 3, 5])
def test_kv_commands_replicate_to_every_member(oracle_lib, n_members):
    G = 8
    eng = oracle_lib.Oracle(G, n_members)
    kv_commands_replicate(eng, G, n_members)


def test_kv_commands_replicate_through_the_emulated_product_engine(emulated_engine):
    """The same shell over ra_amd.engine.RaGpuBatch bound (in this test process only) to the product's HIP
    sources compiled for the CPU: rgb_submit / rgb_collect and the kernels decide, not the checker."""
    G, N = 8, 3
    with emulated_engine.RaGpuBatch(G, N, ring_capacity=G * N, ring_slots=2, max_runs=8) as eng:
        kv_commands_replicate(eng, G, N)


@pytest.mark.gpu
@pytest.mark.parametrize("n_members", [3, 5])
def test_kv_commands_replicate_on_the_gpu(n_members):
    """examples/kv_cluster.py in small: the shell drives the HIP engine on the MI355X."""
    from ra_amd import engine
    G = 64
    with engine.RaGpuBatch(G, n_members, ring_capacity=G * n_members, ring_slots=2, max_runs=8) as eng:
        kv_commands_replicate(eng, G, n_members)


def kv_commands_replicate(eng, G, n_members):
    eng.set_state(0, abi.empty_server_states(G, n_members))
    sh = RaShell(eng, G, n_members)
    for g in range(G):
        sh.trigger_election(g, g % n_members)
    sh.run_until_quiet()
    assert [sh.leader_of(g) for g in range(G)] == [g % n_members for g in range(G)]
    want = [dict() for _ in range(G)]
    rng = np.random.default_rng(1)
    for step in range(40):
        for g in range(G):
            k, v = f"k{int(rng.integers(0, 6))}", int(rng.integers(0, 1000))
            if rng.random() < 0.2:
                assert sh.command(g, ("delete", k)); want[g].pop(k, None)
            else:
                assert sh.command(g, ("put", k, v)); want[g][k] = v
        sh.run(int(rng.integers(1, 4)))                               # commands overlap with replication
    sh.run_until_quiet()
    sh.tick_leaders()                                                 # the last commit index reaches the followers
    sh.run_until_quiet()
    for g in range(G):
        lead = sh.leader_of(g)
        li = int(sh.state[g * n_members + lead]["last_index"])
        assert li == 1 + 40                                           # the noop and forty commands
        for slot in range(n_members):
            s = g * n_members + slot
            assert sh.machines[s].state == want[g], (g, slot)
            assert sh.machines[s].applied == li == int(sh.state[s]["last_applied"])
            assert sh.logs[s] == sh.logs[g * n_members + lead] and sh.logs[s][1] == NOOP


def test_a_new_leader_takes_over_and_history_is_kept(oracle_lib):
    G, N = 4, 3
    eng = oracle_lib.Oracle(G, N)
    eng.set_state(0, abi.empty_server_states(G, N))
    sh = RaShell(eng, G, N)
    for g in range(G):
        sh.trigger_election(g, 0)
    sh.run_until_quiet()
    for i in range(5):
        for g in range(G):
            assert sh.command(g, ("put", f"a{i}", i))
    sh.run_until_quiet(); sh.tick_leaders(); sh.run_until_quiet()
    # a timeout alone does not depose a live leader: it answers the pre_vote_rpc by enforcing its
    # leadership (make_all_rpcs, src/ra_server.erl:961-966) and the stander reverts to follower
    sh.trigger_election(0, 2)
    sh.run_until_quiet()
    assert sh.leader_of(0) == 0 and int(sh.state[2]["role"]) == abi.ROLE_FOLLOWER
    for g in range(G):
        sh.partition(g, 0)                                            # the leaders drop off the network
        sh.trigger_election(g, 2)                                     # member 2 times out and stands
    sh.run_until_quiet()
    for g in range(G):
        sh.heal(g, 0)
        assert sh.command(g, ("put", "b", 42))                        # goes to the leader of the highest term
    sh.run_until_quiet(); sh.tick_leaders(); sh.run_until_quiet()
    assert [sh.leader_of(g) for g in range(G)] == [2] * G
    terms = [int(sh.state[g * N + 2]["current_term"]) for g in range(G)]
    assert all(t >= 2 for t in terms)
    for g in range(G):
        for slot in range(N):
            mac = sh.machines[g * N + slot]
            assert mac.state == {**{f"a{i}": i for i in range(5)}, "b": 42}
            assert int(sh.state[g * N + slot]["role"]) == (abi.ROLE_LEADER if slot == 2 else abi.ROLE_FOLLOWER)
