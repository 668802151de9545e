import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ra_amd import abi, engine, workload as W
G, N, seed = 65536, 5, 0x5EED0003
S = G * N; B = 131072
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_capacity=B, ring_slots=4)
st0 = W.initial_states(G, N, seed); eng.set_state(0, st0)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dn = torch.zeros(1, dtype=torch.int32, device="cuda")
ticks = []
for t in range(8):
    with torch.cuda.stream(stream):
        eng.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), 0, sp)
    torch.cuda.synchronize()
    ticks.append(dm[:int(dn.item()) * 64].cpu().numpy().view(abi.MSG_DTYPE).copy())
bufs = (np.empty(B, dtype=abi.DECISION_DTYPE), np.empty(B * 4, dtype=abi.RPC_DTYPE))
eng.set_state(0, st0)
ts = tc = 0.0; nd = 0
for m in ticks:
    for i in range(0, len(m), B):
        c = m[i:i + B]
        t0 = time.perf_counter(); eng.submit(c); t1 = time.perf_counter()
        eng.synchronize(); t2 = time.perf_counter()
        d, r, _ = eng.collect(out=bufs); t3 = time.perf_counter()
        ts += t1 - t0; tc += t3 - t2; nd += len(c)
        if i == 0 and m is ticks[0]: print("batch", len(c), "submit %.2f ms  gpu-wait %.2f ms  collect %.2f ms  rpcs %d" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, len(r)))
print("per message: submit %.1f ns, collect %.1f ns; serial rate %.1f M/s" % (ts / nd * 1e9, tc / nd * 1e9, nd / (ts + tc) / 1e6))
