"""RGB_CFG_FUSE_PIPELINE (opt-in, include/ra_gpu_batch.h): a leader's same-term success reply and its own written event
end with {next_event, info, pipeline_rpcs} in the reference (src/ra_server.erl:552, 744), handled before anything else
in the mailbox (:793-801).  With the flag the SAME decision carries that event's rpcs.  Claim under test: a fused
decision + its rpc records == the CHECKER's reply decision followed by the checker's RGB_MSG_PIPELINE_RPCS decision for
the same server, and the states are equal -- on random states and ticks, for several group sizes; the default engine
(flag off) is compared with the checker as everywhere else.  Runs on the CPU block emulation and on the GPU (-m gpu)."""
import numpy as np
import pytest

import fuzz
from ra_amd import abi


def check_fused(engine, oracle_lib, n_members, groups, seed, ticks=4):
    rng = np.random.default_rng(seed)
    st = fuzz.random_states(rng, groups, n_members, max_runs=6)
    S = groups * n_members
    cpu = oracle_lib.Oracle(groups, n_members)
    cpu.set_state(0, st)
    fused = engine.RaGpuBatch(groups, n_members, ring_capacity=max(S, 64), ring_slots=2, flags=abi.CFG_FUSE_PIPELINE)
    fused.set_state(0, st)
    n_fused = n_rpcs_fused = n_more = 0
    per = max(n_members - 1, 1)
    for tick in range(ticks):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), n_members)
        if tick % 2:                                        # more leaders' replies: the clause under test
            lead = np.flatnonzero(cpu.get_state()["role"][msgs["server"]] == 1)
            msgs["kind"][lead[::2]] = abi.MSG_AER_REPLY
            msgs["flags"][lead[::2]] = 1
            msgs["term"][lead[::2]] = cpu.get_state()["current_term"][msgs["server"][lead[::2]]]
            msgs["from"][lead[::2]] = (cpu.get_state()["self"][msgs["server"][lead[::2]]] + 1) % n_members
        got_d, got_r = fused.step(msgs)
        want_d, _ = cpu.step(msgs)
        ask = np.flatnonzero(((want_d["kind"] == abi.MSG_AER_REPLY) | (want_d["kind"] == abi.MSG_WRITTEN)) &
                             ((want_d["flags"] & abi.F_PIPELINE) != 0))
        pm = np.zeros(len(ask), dtype=abi.MSG_DTYPE)
        pm["server"] = want_d["server"][ask]
        pm["kind"] = abi.MSG_PIPELINE_RPCS
        pm["from"] = abi.NONE if hasattr(abi, "NONE") else 0xFF
        pd, pr = cpu.step(pm) if len(pm) else (np.zeros(0, dtype=abi.DECISION_DTYPE), np.zeros(0, dtype=abi.RPC_DTYPE))
        exp = want_d.copy()
        for j, i in enumerate(ask):
            if pd["flags"][j] & abi.F_INVARIANT:
                continue                                    # the event would fail an assertion: not fused, PIPELINE stays
            exp["flags"][i] = (want_d["flags"][i] & ~np.uint32(abi.F_PIPELINE)) | pd["flags"][j]
            exp["n_rpcs"][i] = pd["n_rpcs"][j]
            n_fused += 1
            n_rpcs_fused += int(pd["n_rpcs"][j])
            n_more += int((pd["flags"][j] & abi.F_PIPELINE) != 0)
        if got_d.tobytes() != exp.tobytes():
            bad = int(np.flatnonzero((got_d.view(np.uint8).reshape(-1, 64) != exp.view(np.uint8).reshape(-1, 64)).any(axis=1))[0])
            raise AssertionError(f"tick {tick} message {bad}: msg={msgs[bad]}\n fused={got_d[bad]}\n expected={exp[bad]}")
        # the rpc records of every fused decision = the records of the checker's pipeline_rpcs decision for that server
        by_msg = {}
        for r in got_r:
            by_msg.setdefault(int(r["msg_index"]), []).append(r)
        pr_by = {}
        for r in pr:
            pr_by.setdefault(int(r["msg_index"]), []).append(r)
        for j, i in enumerate(ask):
            if pd["flags"][j] & abi.F_INVARIANT:
                continue
            a = sorted(by_msg.get(int(i), []), key=lambda r: int(r["peer"]))
            b = sorted(pr_by.get(int(j), []), key=lambda r: int(r["peer"]))
            assert len(a) == len(b) == int(pd["n_rpcs"][j]), f"tick {tick}: rpc count of message {i}"
            for x, y in zip(a, b):
                x, y = x.copy(), y.copy()
                x["msg_index"] = 0; y["msg_index"] = 0
                assert x.tobytes() == y.tobytes(), f"tick {tick} message {i}: rpc {x} vs {y}"
        # (decisions that are not fused carry the records they always did: failed replies, appends)
        assert fused.get_state().tobytes() == cpu.get_state().tobytes(), f"tick {tick}: state differs"
    fused.close()
    cpu.close()
    assert n_fused > 0 and n_rpcs_fused > 0, (n_fused, n_rpcs_fused)
    return n_fused, n_rpcs_fused, n_more


@pytest.mark.parametrize("n_members,groups,seed", [(5, 120, 901), (3, 150, 902), (7, 80, 903)])
def test_fused_pipeline_on_the_block_emulation(emulated_engine, oracle_lib, n_members, groups, seed):
    check_fused(emulated_engine, oracle_lib, n_members, groups, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("n_members,groups,seed", [(5, 1500, 911), (3, 2200, 912), (7, 900, 913), (8, 300, 914), (2, 400, 915)])
def test_fused_pipeline_on_the_gpu(oracle_lib, n_members, groups, seed):
    import os
    from ra_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    engine.lib()
    check_fused(engine, oracle_lib, n_members, groups, seed)


def test_closed_loop_cluster_with_fused_pipelining(emulated_engine, capsys):
    """End to end (tests/cluster_sim.py: decisions routed back as the next messages over a lossy network, Raft's safety
    properties checked on the full logs after every tick, then heal and converge): the same closed loop with and
    without RGB_CFG_FUSE_PIPELINE.  Fused, the leaders' {next_event, info, pipeline_rpcs} round trips disappear from
    the message stream: fewer messages for the same replicated commands."""
    from cluster_sim import ClusterSim
    import test_cluster_safety as TCS
    G, N, seed = 6, 3, 77
    out = {}
    for name, flags in (("plain", 0), ("fused", abi.CFG_FUSE_PIPELINE)):
        eng = emulated_engine.RaGpuBatch(G, N, ring_capacity=64, ring_slots=2, max_runs=16, flags=flags)
        eng.set_state(0, abi.empty_server_states(G, N))
        sim = TCS.run_lossy_then_heal(eng, G, N, seed, lossy_ticks=300, heal_ticks=200)
        TCS.check_converged(sim, G, N)
        loops = sum(int(((h["kind"] == abi.MSG_PIPELINE_RPCS) & (h["flags"] == 0)).sum()) for h in sim.history
                    if not isinstance(h, tuple))
        out[name] = dict(msgs=sim.stats["msgs"], ticks=sim.tick, commands=sim.stats["commands"], pipeline_msgs=loops)
        eng.close()
    with capsys.disabled():
        print("\nclosed loop, messages with / without fused pipelining:", out)
    assert out["fused"]["pipeline_msgs"] < out["plain"]["pipeline_msgs"] * 0.5, out
    assert out["fused"]["msgs"] < out["plain"]["msgs"], out
