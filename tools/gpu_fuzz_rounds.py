#!/usr/bin/env python3
"""The fused-rounds parity test (tests/test_gpu_parity.py::test_rounds_of_one_batch_run_as_one_train_launch) over more
seeds on a real GPU, in one process:  python tools/gpu_fuzz_rounds.py [first_offset [n_offsets]]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from ra_amd import engine
from oracle import oracle as O
import test_gpu_parity as G
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
bad = 0
for off in range(first, first + n):
    os.environ["RGB_FUZZ_SEED_OFFSET"] = str(off)
    for runs in (6, 16):
        try:
            N = (5, 3, 7, 6)[off % 4]            # (groups of six and seven: the leader-side slices of 32 with LDS peers rows)
            G.test_rounds_of_one_batch_run_as_one_train_launch(engine, O, runs, G=(1500 if off % 2 else 2400) * 5 // N, N=N)
        except AssertionError as e:
            bad += 1
            print(f"offset {off} table_runs {runs}: {str(e)[:400]}")
print(f"{2 * n} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
