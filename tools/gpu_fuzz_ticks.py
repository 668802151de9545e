#!/usr/bin/env python3
"""tests/test_gpu_parity.py::test_hip_equals_oracle_on_random_ticks over more seeds on a real GPU, every group size
(decisions, rpc records through the device-written results, the full state and the checksum of checksums against the
oracle, six ticks each):  python tools/gpu_fuzz_ticks.py [first_seed [n_seeds]]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from ra_amd import engine
from oracle import oracle as O
import test_gpu_parity as G
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = time.time(); bad = 0
for seed in range(first, first + n):
    N = 1 + seed % 8
    groups = (300, 1300, 150, 2200)[seed % 4] * 5 // max(N, 2)
    try:
        G.test_hip_equals_oracle_on_random_ticks(engine, O, N, seed, groups)
    except AssertionError as e:
        if "fuzz never produced" in str(e): continue          # (coverage assertion of the test's own seeds)
        bad += 1; print(f"seed {seed} N {N} groups {groups}: {str(e)[:400]}")
print(f"{n} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
