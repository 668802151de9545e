#!/usr/bin/env python3
"""Kernel-time breakdown probe: times rgb_tick_kernel on single-kind ticks (replaying one
device-resident tick many times) to separate launch floor / message+decision streaming / per
clause-family cost.  Not a benchmark: the replayed messages go stale after the first pass."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ra_amd import abi, engine, workload as W

G, N = 65536, 5
REPS = int(os.environ.get("REPS", "300"))
dev = torch.device("cuda", 0)
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
st0 = W.initial_states(G, N, 0x5EED0003)
eng.set_state(0, st0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
full = W.gen_tick(st0, N, 0, 0x5EED0003, W.MIX_CONFIG3)

def time_tick(name, msgs, reps=REPS):
    n = len(msgs)
    dm = torch.from_numpy(np.ascontiguousarray(msgs).view(np.uint8).reshape(-1)).to(dev)
    dd = torch.empty(max(n, 1) * 64, dtype=torch.uint8, device=dev)
    dr = torch.empty(max(n, 1) * 4 * 56, dtype=torch.uint8, device=dev)
    counts = np.full(reps, n, dtype=np.uint32)
    # replay the same tick: stride 0 is not allowed by the layout, so launch one tick `reps` times
    eng.set_state(0, st0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        eng.run_ticks_device(dm.data_ptr(), n, 1, dd.data_ptr(), dr.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        eng.run_ticks_device(dm.data_ptr(), n, 1, dd.data_ptr(), dr.data_ptr(), stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} n={n:7d}  {us:8.2f} us/launch  {n / us / 1e3 if us else 0:8.2f} G msgs/s", flush=True)

nop = np.zeros(81920, dtype=abi.MSG_DTYPE)
time_tick("nop x81920", nop)
time_tick("nop x1024", nop[:1024])
time_tick("nop x327680", np.zeros(327680, dtype=abi.MSG_DTYPE))
for kind, nm in ((abi.MSG_AER, "aer"), (abi.MSG_AER_REPLY, "aer_reply"), (abi.MSG_REQUEST_VOTE, "request_vote"),
                 (abi.MSG_APPEND, "append"), (abi.MSG_WRITTEN, "written")):
    sel = full[full["kind"] == kind]
    time_tick(f"{nm} (from mix)", sel)
time_tick("full mix tick 0", full)
# a dense tick: one reply per leader + one AER per follower
big = []
cur = st0
lead = np.flatnonzero(st0["role"] == abi.ROLE_LEADER)
m = np.zeros(len(lead), dtype=abi.MSG_DTYPE)
m["server"] = lead; m["kind"] = abi.MSG_AER_REPLY; m["flags"] = abi.MF_SUCCESS
m["from"] = (st0["self"][lead] + 1) % N
m["term"] = st0["current_term"][lead]; m["b"] = st0["last_index"][lead]; m["a"] = st0["last_index"][lead] + 1
m["c"] = st0["last_term"][lead]
time_tick("reply_ok to every leader", m)
fol = np.flatnonzero(st0["role"] == abi.ROLE_FOLLOWER)
a = np.zeros(len(fol), dtype=abi.MSG_DTYPE)
a["server"] = fol; a["kind"] = abi.MSG_AER; a["from"] = st0["leader_id"][fol]
a["term"] = st0["current_term"][fol]; a["a"] = st0["last_index"][fol]; a["b"] = st0["last_term"][fol]
a["c"] = st0["commit_index"][fol]; a["n_entries"] = 2; a["n_run0"] = 2; a["run0_term"] = st0["current_term"][fol]
time_tick("aer to every follower", a)
time_tick("aer to 65536 followers", a[:65536])
time_tick("reply+aer, 1 msg/server", np.concatenate([a, m]))
