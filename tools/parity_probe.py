#!/usr/bin/env python3
"""Debug aid (GPU box): apply N generator ticks with the library in RGB_LIB, compare every decision with the oracle,
report mismatching slots per tick (count, first few, kinds) instead of stopping at the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
from oracle import oracle as O
G, N, seed, T = int(os.environ.get("PP_GROUPS", "65536")), 5, 0x5EED0003, int(os.environ.get("PP_TICKS", "4"))
S = G * N
st0 = W.initial_states(G, N, seed)
cpu = O.Oracle(G, N, max_runs=16); cpu.set_state(0, st0)
eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64); eng.set_state(0, st0)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
dr = torch.zeros(S * 4 * 56, dtype=torch.uint8, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
for t in range(T):
    dd.zero_()
    eng.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
    eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
    stream.synchronize(); torch.cuda.synchronize()
    n = int(dn.item())
    msgs = dm[:n * 64].cpu().numpy().view(abi.MSG_DTYPE)
    got = abi.expand_decisions(dd[:n * 64].cpu().numpy().view(abi.DECISION_DTYPE))
    want, _ = cpu.step_parallel(msgs)
    bad = np.flatnonzero((got.view(np.uint8).reshape(n, 64) != want.view(np.uint8).reshape(n, 64)).any(axis=1))
    kinds = np.bincount(msgs["kind"], minlength=16)
    offs = np.concatenate([[0], np.cumsum(np.bincount(abi.family(msgs), minlength=32))])
    print(f"tick {t}: n={n} mismatches={len(bad)} first={bad[:12].tolist()} kinds_of_bad={np.bincount(msgs['kind'][bad], minlength=16).tolist() if len(bad) else []}")
    for b in bad[:4]:
        print("   slot", int(b), "msg srv", int(msgs['server'][b]), "kind", int(msgs['kind'][b]), "| gpu srv", int(got['server'][b]), "kind", int(got['kind'][b]), "flags", hex(int(got['flags'][b])), "| cpu flags", hex(int(want['flags'][b])))
    if len(bad):
        # where is the decision the gpu put there supposed to be?
        srv_to_slot = {int(s): i for i, s in enumerate(msgs["server"])}
        print("   gpu decision's own slot:", [srv_to_slot.get(int(got['server'][b]), -1) for b in bad[:8]], "family offsets", offs[:8].tolist())
    if len(bad):
        b0 = int(bad[0]) // 64 * 64
        print("   wave at", b0, "got servers", got["server"][b0:b0 + 64].tolist())
        print("   want servers            ", want["server"][b0:b0 + 64].tolist())
        w = got.view(np.uint32).reshape(n, 16); ww = want.view(np.uint32).reshape(n, 16)
        print("   differing dwords per slot (first 8 bad):", [np.flatnonzero(w[b] != ww[b]).tolist() for b in bad[:8]])
        for b in bad[:4]:
            print("   slot", int(b), "gpu dwords", [hex(int(x)) for x in w[b]])
            print("   slot", int(b), "cpu dwords", [hex(int(x)) for x in ww[b]])
        bw = np.unique(bad // 64)
        print("   bad waves (slot/64):", bw[:20].tolist(), "lanes hit:", sorted(set((bad % 64).tolist())))
    state_ok = eng.get_state().tobytes() == cpu.get_state().tobytes()
    print("   state equal:", state_ok)
