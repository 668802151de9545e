#!/usr/bin/env python3
"""Replay one case of tests/test_gpu_parity.py::test_hip_equals_oracle_on_random_ticks and print the first
server whose state differs between the engine and the checker, with its message, both decisions and the
state before and after (edit n, seed, groups, target below).  Needs the GPU."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, fuzz
from oracle import oracle as O
from ra_amd import abi, engine
n, seed, groups, target = 5, 107, 1300, 3731
rng = np.random.default_rng(seed)
st = fuzz.random_states(rng, groups, n, max_runs=6)
cpu = O.Oracle(groups, n); cpu.set_state(0, st)
with engine.RaGpuBatch(groups, n, ring_capacity=max(4096, groups * n), ring_slots=2, max_runs=16) as gpu:
    gpu.set_state(0, st)
    for tick in range(6):
        cur = cpu.get_state()
        msgs = fuzz.random_msgs(rng, cur, n)
        before_g = gpu.get_state()[target].copy()
        do, ro = cpu.step(msgs); dg, rg = gpu.step(msgs)
        sg, so = gpu.get_state(), cpu.get_state()
        k = np.flatnonzero(msgs["server"] == target)
        if sg[target].tobytes() != so[target].tobytes():
            print("tick", tick, "msg", msgs[k] if len(k) else None)
            print("before gpu", before_g)
            print("before cpu", cur[target])
            print("dec gpu", dg[k], "\ndec cpu", do[k])
            print("after gpu", sg[target]); print("after cpu", so[target])
            break
        else:
            print("tick", tick, "ok; msg kind", msgs["kind"][k] if len(k) else None)
