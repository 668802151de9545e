#!/usr/bin/env python3
"""Probe: K independent sub-shards of the groups on ONE GPU, each with its own engine context,
stream and per-16-tick hipGraphs, replayed concurrently.  Does overlapping one sub-shard's
kernel-boundary flush / memory phases with the others' compute raise decisions/s?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
Gtot, N, T, PER = 65536, 5, 96, 16
for K in [int(x) for x in (sys.argv[1:] or ["1", "2", "4"])]:
    G = Gtot // K; S = G * N; NK = abi.N_KINDS; tb = S * 64
    shards = []
    for k in range(K):
        eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
        seed = 0x5EED0003 + k
        st0 = W.initial_states(G, N, seed); eng.set_state(0, st0)
        stream = torch.cuda.Stream(); sp = stream.cuda_stream
        dm = torch.empty(T * tb, dtype=torch.uint8, device="cuda"); dd = torch.empty(T * tb, dtype=torch.uint8, device="cuda")
        dr = torch.empty(S * 4 * 56, dtype=torch.uint8, device="cuda")
        kc = torch.zeros(T * NK, dtype=torch.int32, device="cuda"); dn = torch.zeros(T, dtype=torch.int32, device="cuda")
        with torch.cuda.stream(stream):
            for t in range(T):
                eng.synth_tick_device(seed, t, dm.data_ptr() + t * tb, kc.data_ptr() + t * NK * 4, dn.data_ptr() + t * 4, sp)
                eng.synth_apply_tick_device(dm.data_ptr() + t * tb, S, dd.data_ptr() + t * tb, dr.data_ptr(), sp)
        torch.cuda.synchronize()
        kch = kc.cpu().numpy().reshape(T, NK).astype(np.uint32)
        eng.set_state(0, st0)
        graphs = []
        for p in range(T // PER):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                eng.run_ticks_device(dm.data_ptr() + p * PER * tb, S, PER, dd.data_ptr() + p * PER * tb, dr.data_ptr(),
                                     stream.cuda_stream, kind_counts=kch[p * PER:(p + 1) * PER])
            graphs.append(g)
        shards.append(dict(eng=eng, st0=st0, stream=stream, graphs=graphs, n=int(kch[:, 1:].sum()), keep=(dm, dd, dr)))
    best = 1e9
    for rep in range(4):
        for s in shards: s["eng"].set_state(0, s["st0"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in range(T // PER):
            for s in shards:
                with torch.cuda.stream(s["stream"]): s["graphs"][p].replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ndec = sum(s["n"] for s in shards)
    print(f"K={K}: {ndec} decisions in {best*1e6:.0f} us wall -> {ndec/best/1e9:.2f} G decisions/s, {best*1e6/T:.2f} us per full tick", flush=True)
    del shards; torch.cuda.empty_cache()
