#!/usr/bin/env python3
"""The fixed cost of a host round trip: rgb_submit + rgb_collect_view of 64 / 1 024 / 4 096 single-round messages
(one class launch + the two results kernels), p50 of 300, and rgb_submit's own share."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
G, N, seed = 65536, 5, 0x5EED0003
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=2, ring_capacity=1 << 16)
st0 = W.initial_states(G, N, seed); eng.set_state(0, st0)
S = G * N
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
with torch.cuda.stream(stream):
    eng.synth_tick_device(seed, 0, dm.data_ptr(), 0, dn.data_ptr(), sp)
torch.cuda.synchronize()
tick = dm[:int(dn.item()) * 64].cpu().numpy().view(abi.MSG_DTYPE).copy()
for n in (64, 1024, 4096, 16384):
    m = tick[:n].copy()
    for _ in range(30):
        eng.submit(m); eng.release(eng.collect_view()[3])
    rt, sub = [], []
    for _ in range(300):
        t0 = time.perf_counter(); eng.submit(m); t1 = time.perf_counter(); eng.release(eng.collect_view()[3]); t2 = time.perf_counter()
        rt.append(t2 - t0); sub.append(t1 - t0)
    rt.sort(); sub.sort()
    print(f"n {n:6d}: round trip p50 {rt[150] * 1e6:6.1f} us, rgb_submit p50 {sub[150] * 1e6:6.1f} us")
eng.close()
