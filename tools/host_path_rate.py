#!/usr/bin/env python3
"""PCIe-inclusive rate of the host path (rgb_submit -> kernel(s) -> rgb_collect), the number an
Erlang caller of the NIF would see.  Not the bench metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
G, N, seed = 65536, 5, 0x5EED0003
S = G * N
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_capacity=262144, ring_slots=4)
st0 = W.initial_states(G, N, seed)
eng.set_state(0, st0)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dn = torch.zeros(1, dtype=torch.int32, device="cuda")
ticks = []
for t in range(12):
    with torch.cuda.stream(stream):
        eng.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), 0, sp)
    torch.cuda.synchronize()
    ticks.append(dm[:int(dn.item()) * 64].cpu().numpy().view(abi.MSG_DTYPE).copy())
bufs = (np.empty(262144, dtype=abi.DECISION_DTYPE), np.empty(262144 * 4, dtype=abi.RPC_DTYPE))
for size in (256, 4096, 65536, None):
    eng.set_state(0, st0)
    t0 = time.perf_counter(); nd = 0
    # keep the ring full: submit up to 3 batches ahead, then collect
    pending = 0
    for m in ticks:
        step = 262144 if size is None else size          # never above the ring capacity
        chunks = [m[i:i + step] for i in range(0, len(m), step)]
        if size is not None and size < 4096:
            chunks = chunks[:64]
        for c in chunks:
            while pending >= 3:
                d, r, _ = eng.collect(out=bufs); pending -= 1
            try:
                eng.submit(c)
            except Exception as e:
                k = c["kind"]; bad = (c["server"] >= S) | (k > 12) | ((c["from"] != 255) & (c["from"] >= 8)) | ((k == 1) & (c["n_run0"] > c["n_entries"])) | ((k == 5) & (c["a"] > c["b"]))
                print("submit failed", e, "len", len(c), "bad msgs", int(bad.sum()), c[bad][:3]); raise
            pending += 1; nd += len(c)
    while pending:
        eng.collect(out=bufs); pending -= 1
    dt = time.perf_counter() - t0
    print(f"batch {size or 'full tick (~212k)'}: {nd / dt / 1e6:8.1f} M decisions/s through submit/collect ({nd} decisions, {dt*1e3:.1f} ms)")
