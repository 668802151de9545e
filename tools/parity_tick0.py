#!/usr/bin/env python3
"""Diagnostic: ticks 0..T-1 of the generator's stream at full size against the oracle; prints every kind of mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
from oracle import oracle as O
G, N, seed, T = int(os.environ.get("G", 65536)), 5, 0x5EED0003, int(os.environ.get("T", 3))
S = G * N
st0 = W.initial_states(G, N, seed)
cpu = O.Oracle(G, N, max_runs=16); cpu.set_state(0, st0)
gpu = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64); gpu.set_state(0, st0)
if os.environ.get("HINT") is not None: gpu.synth_set_hint(int(os.environ["HINT"]))
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dr = torch.zeros(S * 4 * 56, dtype=torch.uint8, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
for t in range(T):
    dd.zero_()
    gpu.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
    gpu.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
    stream.synchronize()
    n = int(dn.item())
    msgs = dm[:n * 64].cpu().numpy().view(abi.MSG_DTYPE)
    raw = dd[:n * 64].cpu().numpy().view(abi.DECISION_DTYPE)
    got = abi.expand_decisions(raw)
    want, _ = cpu.step_parallel(msgs)
    bad = np.flatnonzero((got.view(np.uint8).reshape(n, 64) != want.view(np.uint8).reshape(n, 64)).any(axis=1))
    print(f"tick {t}: {n} messages, {len(bad)} mismatches")
    if len(bad):
        up0 = (raw.view(np.uint64).reshape(n, 8)[bad, 4:] == 0).all(axis=1)
        print("  upper half zero in", int(up0.sum()), "of them; kinds", np.bincount(msgs["kind"][bad], minlength=16).tolist())
        print("  lane of slot (mod 64):", np.bincount(bad % 64, minlength=64).tolist())
        print("  first slots", bad[:10].tolist(), "flags", [hex(int(x)) for x in got["flags"][bad[:10]]])
        bk = engine.train_bucket(msgs["kind"], msgs["flags"], msgs["server"], N)
        print("  buckets (kind class*16+shard*2+flag) of the bad:", np.unique(bk[bad])[:20].tolist())
        # where do the bad ones sit inside their bucket?
        for b in np.unique(bk[bad])[:4]:
            idx = np.flatnonzero(bk == b)
            print(f"  bucket {b}: slots {idx[0]}..{idx[-1]} ({len(idx)}), bad at offsets", (bad[np.isin(bad, idx)] - idx[0])[:12].tolist())
        break
