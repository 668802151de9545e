#!/usr/bin/env python3
"""The host path of ONE library (RGB_LIB, default the product): rgb_submit -> kernels -> rgb_collect, PCIe both ways.
Prints one JSON line: the single-thread pipelined rate in 131 072-message batches, the four-round small batch's round
trip with its breakdown, and a digest of everything rgb_collect handed out (decisions AND rpc records, first pass) so
that two libraries can be compared byte for byte on the same box:
    RGB_LIB=ra_amd/csrc/variants/hp_base.so python tools/host_path_ab.py
Not the bench metric; bench.py's host_path leg reports the same figures for the product."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W

G, N, seed = 65536, 5, 0x5EED0003
S = G * N
B = 131072
TICKS = int(os.environ.get("HP_TICKS", "12"))
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_capacity=B, ring_slots=4)
st0 = W.initial_states(G, N, seed)
eng.set_state(0, st0)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dn = torch.zeros(1, dtype=torch.int32, device="cuda")
ticks = []
for t in range(TICKS):
    with torch.cuda.stream(stream):
        eng.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), 0, sp)
    torch.cuda.synchronize()
    ticks.append(dm[:int(dn.item()) * 64].cpu().numpy().view(abi.MSG_DTYPE).copy())
bufs = (np.empty(B, dtype=abi.DECISION_DTYPE), np.empty(B * (N - 1), dtype=abi.RPC_DTYPE))
out = {"lib": os.environ.get("RGB_LIB", "product")}

# ---- digest pass: everything handed out, in order ----
eng.set_state(0, st0)
h = hashlib.sha256(); n_dec = n_rpc = 0
for m in ticks:
    for i in range(0, len(m), B):
        eng.submit(m[i:i + B])
        d, r, _ = eng.collect(out=bufs)
        h.update(d.tobytes()); h.update(r.tobytes()); n_dec += len(d); n_rpc += len(r)
out["digest"] = h.hexdigest()[:16]; out["decisions"] = n_dec; out["rpcs"] = n_rpc
out["state_checksum"] = f"{eng.state_checksum():#018x}"

# ---- single thread, three batches ahead (bench.py's host_path.value) ----
best = 0.0
for rep in range(3):
    eng.set_state(0, st0)
    pending = nd = 0
    t0 = time.perf_counter()
    for m in ticks:
        for i in range(0, len(m), B):
            while pending >= 3:
                eng.collect(out=bufs); pending -= 1
            eng.submit(m[i:i + B]); pending += 1; nd += len(m[i:i + B])
    while pending:
        eng.collect(out=bufs); pending -= 1
    best = max(best, nd / (time.perf_counter() - t0))
out["one_thread_M_per_s"] = round(best / 1e6, 1)

# ---- the same with rgb_collect_view / rgb_release (ABI v9): the results are read where the device wrote them ----
if hasattr(eng._L, "rgb_collect_view"):
    best = 0.0
    for rep in range(3):
        eng.set_state(0, st0)
        pending = nd = 0
        t0 = time.perf_counter()
        for m in ticks:
            for i in range(0, len(m), B):
                while pending >= 3:
                    eng.release(eng.collect_view()[3]); pending -= 1
                eng.submit(m[i:i + B]); pending += 1; nd += len(m[i:i + B])
        while pending:
            eng.release(eng.collect_view()[3]); pending -= 1
        best = max(best, nd / (time.perf_counter() - t0))
    out["one_thread_view_M_per_s"] = round(best / 1e6, 1)

# ---- per-call split of full batches: submit / device behind it / collect ----
eng.set_state(0, st0)
ts = td = tc = 0.0; nd = 0
for m in ticks[:6]:
    for i in range(0, len(m), B):
        c = m[i:i + B]
        t0 = time.perf_counter(); eng.submit(c); t1 = time.perf_counter()
        eng.synchronize(); t2 = time.perf_counter()
        eng.collect(out=bufs); t3 = time.perf_counter()
        ts += t1 - t0; td += t2 - t1; tc += t3 - t2; nd += len(c)
out["ns_per_message"] = {"rgb_submit": round(ts / nd * 1e9, 2), "device_behind_submit": round(td / nd * 1e9, 2),
                         "rgb_collect": round(tc / nd * 1e9, 2)}
eng.close()

# ---- the small four-round batch (one scheduler's mailbox drain): round trip and its parts ----
small = np.concatenate([m[m["server"] < 1024 * N] for m in ticks[:4]])
lat = {}
for label, flags in (("launch_per_round", abi.CFG_ROUNDS_PER_LAUNCH), ("fused_train", getattr(abi, "CFG_SUBMIT_TRAINS", 0))):
    e2 = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=2, ring_capacity=1 << 16, flags=flags)
    b2 = (np.empty(1 << 16, dtype=abi.DECISION_DTYPE), np.empty((1 << 16) * (N - 1), dtype=abi.RPC_DTYPE))
    e2.set_state(0, st0)
    for _ in range(20):
        e2.submit(small); e2.collect(out=b2)
    rt = []
    for _ in range(200):
        t0 = time.perf_counter(); e2.submit(small); e2.collect(out=b2); rt.append(time.perf_counter() - t0)
    rt.sort()
    bd = [[], [], []]
    for _ in range(100):
        t0 = time.perf_counter(); e2.submit(small)
        t1 = time.perf_counter(); e2.synchronize()
        t2 = time.perf_counter(); e2.collect(out=b2)
        t3 = time.perf_counter()
        bd[0].append(t1 - t0); bd[1].append(t2 - t1); bd[2].append(t3 - t2)
    lat[label] = {"round_trip_us_p50": round(rt[len(rt) // 2] * 1e6, 1), "round_trip_us_p10": round(rt[len(rt) // 10] * 1e6, 1),
                  "breakdown_us_p50": {k: round(sorted(v)[len(v) // 2] * 1e6, 1)
                                       for k, v in zip(("rgb_submit", "device_behind_submit", "rgb_collect"), bd)}}
    if hasattr(e2._L, "rgb_collect_view"):
        rv = []
        for _ in range(200):
            t0 = time.perf_counter(); e2.submit(small); e2.release(e2.collect_view()[3]); rv.append(time.perf_counter() - t0)
        rv.sort()
        lat[label]["view_round_trip_us_p50"] = round(rv[len(rv) // 2] * 1e6, 1)
    e2.close()
out["small_batch"] = {"messages": int(len(small)), **lat}
print(json.dumps(out))
