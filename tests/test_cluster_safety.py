"""Closed-loop clusters (tests/cluster_sim.py): the decisions and rpc records of the restated
transition are routed back as the next messages over a lossy, delaying network, and
the Raft safety properties are checked on the full logs after every tick -- election safety, log
matching, state-machine safety and leader completeness (Ongaro & Ousterhout, figure 3; the properties
ra_server's clauses exist to keep).  After the network heals every group must elect one leader and
replicate every command to every member.  No transcribed vector takes part: this is an end-to-end
check of the semantics themselves, and on the GPU the same message streams must give bit-identical
decisions and states."""
import numpy as np
import pytest

from ra_amd import abi
from cluster_sim import ClusterSim


def run_lossy_then_heal(eng, G, N, seed, lossy_ticks=500, heal_ticks=500, **kw):
    sim = ClusterSim(eng, G, N, seed, **kw)
    for _ in range(lossy_ticks):
        sim.step()
        sim.check_safety()
    sim.heal()
    sim.p_command = 0.05
    for t in range(heal_ticks):
        sim.step()
        sim.check_safety()
    sim.p_command = sim.p_query = sim.p_snapshot = 0.0   # let the tail replicate: run until nothing is in flight
    calm = 0
    for t in range(3000):
        sim.step()
        sim.check_safety()
        calm = calm + 1 if sim.idle() else 0
        if calm >= 60:
            break
    return sim


@pytest.mark.parametrize("n_members,seed", [(3, 1), (5, 2), (3, 3), (7, 4), (2, 5), (1, 6)])
def test_closed_loop_clusters_keep_raft_safety_and_converge(oracle_lib, n_members, seed):
    G = 6
    cpu = oracle_lib.Oracle(G, n_members)
    cpu.set_state(0, abi.empty_server_states(G, n_members))
    sim = run_lossy_then_heal(cpu, G, n_members, seed)
    check_converged(sim, G, n_members)


def check_converged(sim, G, n_members):
    st = sim.state
    assert sim.stats["invariants"] == 0
    assert sim.stats["commands"] > 0 and sim.stats["msgs"] > 1000
    progressed = 0
    for g in range(G):
        rows = st[g * n_members:(g + 1) * n_members]
        leaders = [i for i, r in enumerate(rows) if int(r["role"]) == abi.ROLE_LEADER]
        if not sim.leaders_of_term[g]:
            continue                                 # no election timeout ever fired in this group
        top = max(int(r["current_term"]) for r in rows)
        live = [i for i in leaders if int(rows[i]["current_term"]) == top]
        if sim.elections[g] >= sim.max_leaders and not live:
            continue                                 # ran out of its election budget while partitioned
        assert len(live) == 1, f"group {g}: leaders {leaders} terms {[int(r['current_term']) for r in rows]}"
        lead = rows[live[0]]
        li = int(lead["last_index"])
        assert li >= 1
        for i, r in enumerate(rows):                 # everything replicated, written and committed everywhere
            assert int(r["last_index"]) == li and int(r["last_term"]) == int(lead["last_term"]), (g, i)
            # (a reordered, older append_entries_rpc may have stepped commit_index back -- the follower
            # takes LeaderCommit as it comes, src/ra_server.erl:1331-1332 -- but never last_applied)
            assert int(r["last_applied"]) == li and int(r["commit_index"]) <= li, (g, i, int(r["commit_index"]), li)
            mine, theirs = dict(abi.log_entries(r)), dict(abi.log_entries(lead))
            assert all(theirs[k] == t for k, t in mine.items() if k in theirs)
        progressed += 1
    assert progressed >= G - 1
    return sim


def test_closed_loop_with_tiny_pipeline_limits(oracle_lib):
    """max_append_entries_rpc_batch_size = 2 and max_pipeline_count = 3 (src/ra_server.erl:2285-2346): bursts
    of commands leave the leader with more to send than one round allows, so replication advances through
    the {next_event, info, pipeline_rpcs} loop and the in-flight clamp; safety and convergence as before."""
    G, N = 6, 3
    cpu = oracle_lib.Oracle(G, N, max_pipeline_count=3, max_aer_batch=2)
    cpu.set_state(0, abi.empty_server_states(G, N))
    sim = run_lossy_then_heal(cpu, G, N, 31, lossy_ticks=500, heal_ticks=400, p_command=0.6, drop=0.05)
    check_converged(sim, G, N)
    assert int(sim.state["last_index"].max()) > 60
    loops = sum(int(((h["kind"] == abi.MSG_PIPELINE_RPCS) & (h["flags"] == 0)).sum()) for h in sim.history
                if not isinstance(h, tuple))
    assert loops > 300, loops                        # the pipeline_rpcs round trips carried the replication


@pytest.mark.parametrize("n_members,seed", [(3, 21), (5, 22), (7, 23)])
def test_closed_loop_clusters_with_snapshots(oracle_lib, n_members, seed):
    """The same, with members taking snapshots at last_applied (SNAPSHOT_WRITTEN truncates the log) and
    leaders sending their snapshot to peers that fell behind it (RGB_RPC_SNAPSHOT / RGB_F_SEND_SNAPSHOT;
    the transfer itself is emulated on the host side, tests/cluster_sim.py)."""
    G = 6
    cpu = oracle_lib.Oracle(G, n_members)
    cpu.set_state(0, abi.empty_server_states(G, n_members))
    sim = run_lossy_then_heal(cpu, G, n_members, seed, lossy_ticks=900, heal_ticks=300,
                              p_snapshot=0.03, max_leaders=11, drop=0.15)
    check_converged(sim, G, n_members)
    assert sim.stats["snapshots"] > 50 and sim.stats["installs"] > 0
    assert int(sim.elections.max()) > 9


WAL_DOWN_KW = dict(p_wal_down=0.01, max_leaders=12)


@pytest.mark.parametrize("n_members,seed", [(3, 41), (5, 42), (1, 43)])
def test_closed_loop_clusters_with_wal_outages(oracle_lib, n_members, seed):
    """The same, with every member's WAL going down for a few ticks now and then: followers whose write is refused and
    leaders whose append raises wal_down wait in await_condition (the host recipes of INTEGRATION.md, the two
    wal_down_condition/2 forms of src/ra_server.erl:660-668 and 1377-1385), drop what arrives meanwhile, come back through
    the condition's transition_to with the message re-processed, or through the timeout (the leader's with the
    transfer_leadership effect).  Safety on the full logs after every tick, convergence once the WALs stay up."""
    G = 6
    cpu = oracle_lib.Oracle(G, n_members)
    cpu.set_state(0, abi.empty_server_states(G, n_members))
    sim = run_lossy_then_heal(cpu, G, n_members, seed, lossy_ticks=900, heal_ticks=400, **WAL_DOWN_KW)
    check_converged(sim, G, n_members)
    assert sim.stats["wal_down_leader"] > 0, sim.stats
    if n_members > 1:
        assert sim.stats["wal_down_follower"] > 0 and sim.stats["transfer_leadership"] > 0, sim.stats
        assert sim.stats["wal_down_reprocessed"] > 0, sim.stats
    else:
        assert sim.stats["transfer_leadership"] == 0     # no peer to hand over to (src/ra_server.erl:661-662)


@pytest.mark.gpu
@pytest.mark.parametrize("n_members,seed,snapshots", [(3, 11, False), (5, 12, False), (7, 13, False), (5, 14, True),
                                                      (5, 15, "wal_down")])
def test_gpu_gives_identical_decisions_on_closed_loop_streams(oracle_lib, n_members, seed, snapshots):
    """The streams a live cluster produces (elections, repairs after drops, overwrites by new leaders,
    stale rpcs) replayed through the HIP engine: decisions, rpcs and states bit-identical
    to the checker's at every tick."""
    import os
    from ra_amd import engine
    from test_gpu_parity import assert_same
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    G = 32
    cpu = oracle_lib.Oracle(G, n_members)
    st0 = abi.empty_server_states(G, n_members)
    cpu.set_state(0, st0)
    kw = dict(p_snapshot=0.03, max_leaders=11, drop=0.15) if snapshots is True else WAL_DOWN_KW if snapshots else {}
    sim = run_lossy_then_heal(cpu, G, n_members, seed, lossy_ticks=250, heal_ticks=150, **kw)
    ref = oracle_lib.Oracle(G, n_members)
    ref.set_state(0, st0)
    seen = 0
    with engine.RaGpuBatch(G, n_members, ring_capacity=max(4096, G * n_members), ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st0)
        for t, msgs in enumerate(sim.history):
            if isinstance(msgs, tuple):                  # a host-side state edit (snapshot transfer)
                _, server, row = msgs
                ref.set_state(server, row.reshape(1))
                gpu.set_state(server, row.reshape(1))
                continue
            do, ro = ref.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"closed loop N={n_members} tick {t}", dg, rg, gpu.get_state(), do, ro, ref.get_state())
            seen |= int(np.bitwise_or.reduce(do["flags"]))
    for f in (abi.F_REPLY, abi.F_WROTE, abi.F_BECAME_LEADER, abi.F_SEND_VOTE_REQUESTS, abi.F_PIPELINE, abi.F_APPLIED):
        assert seen & f, hex(f)
    if snapshots is True:
        assert seen & abi.F_SEND_SNAPSHOT
    if snapshots == "wal_down":
        assert seen & abi.F_TRANSFER_LEADERSHIP and sim.stats["wal_down_follower"] > 0 and sim.stats["wal_down_leader"] > 0
