"""Differential fuzz of the host path's normal shape -- several messages per server in ONE rgb_submit, the rounds
fused into one train launch -- on the CPU emulation: random states (shallow and deep run tables), random messages of
every kind, 3 / 5 / 7 members, the checker bounded like the device.  Four seeds run with the suite; more with
RGB_FUZZ_SEEDS=lo:hi (round 3 ran 312:432 clean after seed 27 of the GPU twin had found the range-lost corner of
ra_log:write; round 4 ran 600:720 clean on its final sources; round 5 ran 900:972 and
RGB_FUZZ_WAL_SEEDS=900:924 clean on its final sources -- the leader-side slices of 32 of groups of seven included --
and the train tests under ASan)."""
import os

import numpy as np
import pytest

import fuzz
from ra_amd import abi
from test_gpu_parity import assert_same


def _seeds(var="RGB_FUZZ_SEEDS", default=(300, 301, 302, 307)):
    spec = os.environ.get(var)
    if spec:
        lo, hi = (int(x) for x in spec.split(":"))
        return list(range(lo, hi))
    return list(default)


def _fused_rounds(emulated_engine, oracle_lib, seed, wal_down):
    N = (5, 3, 7, 5)[seed % 4]
    G = {3: 1800, 5: 1100, 7: 800}[N]
    rng = np.random.default_rng(seed)
    deep = (seed // 4) % 2 == 1
    st = fuzz.random_states(rng, G, N, max_runs=16 if deep else 6, backlog=60 if deep else 24)
    if wal_down:
        # a fifth of the servers waits in one of the two wal_down conditions (src/ra_server.erl:660-668, 1377-1385)
        pick = rng.random(G * N) < 0.2
        st["role"][pick] = abi.ROLE_AWAIT_CONDITION
        st["cond_reason"][pick] = rng.choice([abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER], size=int(pick.sum()))
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st)
    with emulated_engine.RaGpuBatch(G, N, ring_capacity=65536, ring_slots=2, max_runs=16, flags=abi.CFG_SUBMIT_TRAINS) as gpu:
        gpu.set_state(0, st)
        for b in range(2):
            parts = [fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.9) for _ in range(4)]
            msgs = np.concatenate(parts)
            msgs = msgs[msgs["kind"] != abi.MSG_NOP]
            rng.shuffle(msgs)
            if wal_down:
                msgs["flags"] |= np.where(rng.random(len(msgs)) < 0.5, abi.MF_CAN_WRITE, 0).astype(msgs["flags"].dtype)
                waiting = cpu.get_state()["role"][msgs["server"]] == abi.ROLE_AWAIT_CONDITION
                msgs["kind"][waiting & (rng.random(len(msgs)) < 0.2)] = abi.MSG_AWAIT_TIMEOUT
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"seed {seed} batch {b}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
        assert gpu.submit_trains() >= 1


@pytest.mark.parametrize("seed", _seeds())
def test_fused_rounds_equal_the_sequential_checker(emulated_engine, oracle_lib, seed):
    _fused_rounds(emulated_engine, oracle_lib, seed, False)


@pytest.mark.parametrize("seed", _seeds("RGB_FUZZ_WAL_SEEDS", [500, 503]))
def test_fused_rounds_with_servers_in_the_wal_down_conditions(emulated_engine, oracle_lib, seed):
    """The same with servers waiting in the follower's and the leader's wal_down condition, messages with and without
    RGB_MF_CAN_WRITE and await_condition timeouts (round 4 ran RGB_FUZZ_WAL_SEEDS=500:580 and 700:760 clean)."""
    _fused_rounds(emulated_engine, oracle_lib, seed, True)


def test_rounds_are_fused_only_on_request(emulated_engine, oracle_lib):
    """Round 5: one launch per sub-tick round is rgb_submit's default; RGB_CFG_SUBMIT_TRAINS opts in to one train launch
    per batch; RGB_CFG_ROUNDS_PER_LAUNCH wins over it.  The same batch gives the same decisions, rpc records and state
    in all three contexts (and the checker's)."""
    N, G = 5, 1100
    rng = np.random.default_rng(4711)
    st = fuzz.random_states(rng, G, N, max_runs=6, backlog=24)
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st)
    parts = [fuzz.random_msgs(rng, st, N, frac=0.9) for _ in range(4)]
    msgs = np.concatenate(parts)
    msgs = msgs[msgs["kind"] != abi.MSG_NOP]
    rng.shuffle(msgs)
    assert len(msgs) >= 4096
    do, ro = cpu.step(msgs)
    for flags, want_trains in ((0, 0), (abi.CFG_SUBMIT_TRAINS, 1), (abi.CFG_SUBMIT_TRAINS | abi.CFG_ROUNDS_PER_LAUNCH, 0)):
        with emulated_engine.RaGpuBatch(G, N, ring_capacity=65536, ring_slots=2, max_runs=16, flags=flags) as gpu:
            gpu.set_state(0, st)
            dg, rg = gpu.step(msgs)
            assert_same(f"flags {flags}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            assert gpu.submit_trains() == want_trains, f"flags {flags}: {gpu.submit_trains()} batches ran as trains"
    cpu.close()
