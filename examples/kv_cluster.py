#!/usr/bin/env python3
"""4096 three-member Raft groups on one MI355X, each replicating a small key-value store: the engine
decides (ra_amd.engine.RaGpuBatch, one kernel launch per tick), ra_amd.shell.RaShell plays
ra_server_proc (routing, in-memory logs, state machines).  Needs the GPU: there is no CPU fallback."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ra_amd import abi, engine
from ra_amd.shell import RaShell

G, N = 4096, 3
eng = engine.RaGpuBatch(G, N, ring_capacity=G * N, ring_slots=2, max_runs=8)      # raises without the HIP library / a GPU
eng.set_state(0, abi.empty_server_states(G, N))
sh = RaShell(eng, G, N)
t0 = time.perf_counter()
for g in range(G):
    sh.trigger_election(g, g % N)
sh.run_until_quiet()
print(f"{G} leaders elected in {sh.ticks} ticks, {time.perf_counter() - t0:.2f} s")
rng = np.random.default_rng(0)
for round_ in range(10):
    for g in range(G):
        sh.command(g, ("put", f"k{int(rng.integers(0, 8))}", round_))
    sh.run(2)
sh.run_until_quiet(); sh.tick_leaders(); sh.run_until_quiet()
same = all(sh.machines[g * N].state == sh.machines[g * N + 1].state == sh.machines[g * N + 2].state for g in range(G))
print(f"10 commands x {G} groups replicated and applied on every member: {same}; {sh.ticks} ticks, "
      f"{time.perf_counter() - t0:.2f} s (host-side Python dominates; bench.py measures the engine)")
eng.close()
