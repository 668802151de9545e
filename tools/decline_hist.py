#!/usr/bin/env python3
"""Why do lanes decline the fast paths?  RGB_LIB = a -DRGB_X_DECLINE_HIST build (tools/build_variants.sh
hist:"-DRGB_X_DECLINE_HIST=1"): ages the bench's 65 536 x 5 closed-loop stream AGE ticks, zeroes the counters, runs TICKS
more ticks with the per-tick class kernel (same fast paths as the train) and prints, per class, lanes taken and the
decline reasons (codes: the FP_DECLINE calls of rgb_kernels.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ra_amd import engine, workload as W
G, N = int(os.environ.get("G", 65536)), 5
AGE, TICKS = int(os.environ.get("AGE", 512)), int(os.environ.get("TICKS", 16))
seed = 0x5EED0003
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
eng.set_state(0, W.initial_states(G, N, seed))
S = G * N
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dr = torch.zeros(S * 4 * 56, dtype=torch.uint8, device="cuda")
L = engine.lib(); L.rgb_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
def run(t0, t1):
    for t in range(t0, t1):
        eng.synth_tick_device(seed, t, dm.data_ptr(), 0, 0, 0)
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), 0)
    eng.synchronize()
def read():
    buf = np.zeros(128, dtype=np.uint64)
    assert L.rgb_debug_read(eng._h, buf.ctypes.data, 128) == 0
    return buf.astype(np.int64)
run(0, AGE)
b0 = read()
run(AGE, AGE + TICKS)
h = read() - b0
names = {0: "append_entries_rpc", 1: "append_entries_reply", 2: "written"}
for c in (0, 1, 2):
    row = h[c * 32:(c + 1) * 32]
    tot = int(row.sum())
    print(f"class {c} {names[c]}: {tot / TICKS:.0f} lanes per tick, fast path {row[0] / max(tot, 1):.3f}")
    for code in range(1, 32):
        if row[code]:
            print(f"    reason {code:2d}: {row[code] / TICKS:9.1f} per tick  {row[code] / max(tot, 1):.4f}")
