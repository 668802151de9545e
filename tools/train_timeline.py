#!/usr/bin/env python3
"""Per-wavefront timeline of ONE train launch (variant built with -DRGB_X_TRAIN_TIMELINE, tools/build_variants.sh):
where a wavefront's life goes -- message load, dependency wait, row fetch, clause code, publish, decision store --
per class, and the cadence of the ticks.   usage (GPU box): RGB_LIB=.../variants/timeline.so python tools/train_timeline.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
G, N = int(os.environ.get("TL_GROUPS", "65536")), 5
T, AGE = int(os.environ.get("TL_TICKS", "16")), int(os.environ.get("TL_AGE", "128"))
S = G * N; tb = S * 64
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
eng.set_state(0, W.initial_states(G, N, 0x5EED0003))
if os.environ.get("TL_HINT") is not None:
    eng.synth_set_hint(int(os.environ["TL_HINT"]))
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sp = stream.cuda_stream
dm = torch.empty(T * tb, dtype=torch.uint8, device="cuda"); dd = torch.empty(T * tb, dtype=torch.uint8, device="cuda")
dr = torch.empty(4 * S * 4 * 56, dtype=torch.uint8, device="cuda")
dn = torch.zeros(T, dtype=torch.int32, device="cuda"); bc = torch.zeros(T * 256, dtype=torch.int32, device="cuda")
for t in range(AGE):
    eng.synth_tick_buckets_device(0x5EED0003, t, dm.data_ptr(), 0, 0, 0, sp)
    eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
torch.cuda.synchronize()
st = eng.get_state()
for t in range(T):
    eng.synth_tick_buckets_device(0x5EED0003, AGE + t, dm.data_ptr() + t * tb, 0, dn.data_ptr() + t * 4, bc.data_ptr() + t * 1024, sp)
    eng.synth_apply_tick_device(dm.data_ptr() + t * tb, S, dd.data_ptr() + t * tb, dr.data_ptr(), sp)
torch.cuda.synchronize()
counts = dn.cpu().numpy().astype(np.uint32)
plan = eng.train_plan(bc.cpu().numpy().reshape(T, 256).astype(np.uint32))
ds = torch.zeros(T * S, dtype=torch.uint8, device="cuda")
for rep in range(2):                      # the second run is the one read back (warm instruction caches)
    eng.set_state(0, st)
    eng.train_stamp_device(dm.data_ptr(), ds.data_ptr(), S, counts, sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    eng.train_run_device(plan, 0, T, dm.data_ptr(), ds.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), 4, sp)
    e1.record(stream)
    torch.cuda.synchronize()
print("train of", T, "ticks:", round(e0.elapsed_time(e1) * 1e3 / T, 2), "us per tick; flags", eng.train_status(check=False)[0],
      "blocks per tick", plan.blocks_per_tick)
import timeline_stats
timeline_stats.report(eng, T, plan.blocks_per_tick)
