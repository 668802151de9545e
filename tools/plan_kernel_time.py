#!/usr/bin/env python3
"""How long does rgb_train_plan_build_device take for n ticks (HIP events around the kernel, best of 20)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import engine, workload as W
G, N, T = 65536, 5, 240
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
eng.set_state(0, W.initial_states(G, N, 0x5EED0003))
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sp = stream.cuda_stream
S = G * N
dm = torch.empty(S * 64, dtype=torch.uint8, device="cuda")
bc = torch.zeros(T * 256, dtype=torch.int32, device="cuda")
for t in range(T):
    eng.synth_tick_buckets_device(0x5EED0003, t, dm.data_ptr(), 0, 0, bc.data_ptr() + t * 1024, sp)
    if t < 40:
        dd = torch.empty(S * 64, dtype=torch.uint8, device="cuda"); dr = torch.empty(S * 4 * 56, dtype=torch.uint8, device="cuda")
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
torch.cuda.synchronize()
plan = engine.TrainPlan(eng, None, snapshot_every=16, device_ticks=T)
for n in (1, 20, 240):
    best = 1e9
    for rep in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); plan.build_device(0, n, bc.data_ptr(), sp); e1.record(stream); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    print(f"plan of {n:3d} ticks: {best:7.2f} us")
print("rows of tick 17:", plan.download(17)[0][0], "table capacity", plan.blocks_per_tick // 8)
