#!/usr/bin/env python3
"""200 round trips of the small four-round batch (one scheduler's mailbox drain, ~15 k messages) through rgb_submit /
rgb_collect_view -- the loop to put under `rocprofv3 --kernel-trace --memory-copy-trace --stats` to see what the device
does behind one rgb_submit (tools/host_path_ab.py times it):
    rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/small_trace -- python tools/small_batch_loop.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W

G, N, seed = 65536, 5, 0x5EED0003
S = G * N
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=2, ring_capacity=1 << 16, flags=abi.CFG_ROUNDS_PER_LAUNCH)
st0 = W.initial_states(G, N, seed)
eng.set_state(0, st0)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dn = torch.zeros(1, dtype=torch.int32, device="cuda")
ticks = []
for t in range(4):
    with torch.cuda.stream(stream):
        eng.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), 0, sp)
    torch.cuda.synchronize()
    ticks.append(dm[:int(dn.item()) * 64].cpu().numpy().view(abi.MSG_DTYPE).copy())
small = np.concatenate([m[m["server"] < 1024 * N] for m in ticks])
eng.set_state(0, st0)
rt = []
for k in range(220):
    t0 = time.perf_counter(); eng.submit(small); eng.release(eng.collect_view()[3]); rt.append(time.perf_counter() - t0)
rt = sorted(rt[20:])
print(f"{len(small)} messages, round trip p50 {rt[len(rt) // 2] * 1e6:.1f} us")
eng.close()
