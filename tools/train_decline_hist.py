#!/usr/bin/env python3
"""Why do lanes decline the fast paths IN A TRAIN LAUNCH (run tables in LDS, hint sub-buckets)?  RGB_LIB = a
-DRGB_X_DECLINE_HIST build.  Ages the closed-loop stream, generates T ticks, replays them as one train launch and
prints per class the lanes that took the fast path and the decline reasons (FP_DECLINE codes of rgb_kernels.hip),
plus the messages per plan class (sub-bucket) of the launch."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import engine, workload as W
G, N = int(os.environ.get("TL_GROUPS", "65536")), 5
T, AGE = int(os.environ.get("TL_TICKS", "32")), int(os.environ.get("TL_AGE", "512"))
S = G * N; tb = S * 64
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
eng.set_state(0, W.initial_states(G, N, 0x5EED0003))
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
buckets = bc.cpu().numpy().reshape(T, 256).astype(np.uint32)
plan = eng.train_plan(buckets)
ds = torch.zeros(T * S, dtype=torch.uint8, device="cuda")
L = engine.lib(); L.rgb_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
def read():
    buf = np.zeros(128, dtype=np.uint64)
    assert L.rgb_debug_read(eng._h, buf.ctypes.data, 128) == 0
    return buf.astype(np.int64)
eng.set_state(0, st)
eng.train_stamp_device(dm.data_ptr(), ds.data_ptr(), S, counts, sp)
torch.cuda.synchronize()
b0 = read()
eng.train_run_device(plan, 0, T, dm.data_ptr(), ds.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), 4, sp)
torch.cuda.synchronize()
h = read() - b0
print("train of", T, "ticks; flags", eng.train_status(check=False)[0], "decisions per tick", counts.mean())
# messages per (class, sub-bucket): bucket = (class * 8 + shard) * 2 + sub
bk = buckets.reshape(T, 16, 8, 2).sum(axis=(0, 2)) / T
names = {0: "append_entries_rpc", 1: "append_entries_reply", 2: "written", 3: "append", 4: "pipeline_rpcs"}
for c in range(16):
    if bk[c].sum():
        print(f"class {c:2d} {names.get(c, ''):22s} sub0 {bk[c, 0]:9.1f}  sub1 {bk[c, 1]:9.1f} per tick")
print("snapshot_written: in-memory runs moved per lane that released runs (0..15+):", (h[96:112] / T).round(1).tolist())
for c in (0, 1, 2):
    row = h[c * 32:(c + 1) * 32]
    tot = int(row.sum())
    if not tot: continue
    print(f"class {c} {names[c]}: {tot / T:.0f} lanes per tick reached its fast path, taken {row[0] / max(tot, 1):.3f}")
    for code in range(1, 32):
        if row[code]:
            print(f"    reason {code:2d}: {row[code] / T:9.1f} per tick  {row[code] / max(tot, 1):.4f}")
