#!/usr/bin/env python3
"""bench.py -- append_entries decisions/sec across N Raft groups on MI355X (BASELINE.json).

Workload (config.workload): BASELINE.json configs[2] -- 65 536 five-member Raft groups per GPU,
mixed append_entries / append_entries_reply / request_vote (5 % term churn) plus the
housekeeping events ({commands,_} appends, {written,..} log events) that keep the logs moving.
A "step" is one tick = one pass of the hot path (one kernel launch) over one batch of synthetic
messages, at most one message per server, messages already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  Groups shard by hash
with no data-path collective (weak scaling: 65 536 groups per GPU); every 16 ticks each rank
produces its ra_leaderboard/metrics shard and the shards are all-gathered over xGMI.

The CPU oracle (oracle/) is used here ONLY as (a) the checker of the first ticks and (b) the
reported cpu_baseline; the timed path is the HIP library behind the C ABI.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SNAPSHOT_EVERY = 16
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--groups", type=int, default=65536, help="Raft groups per GPU")
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0003)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check-ticks", type=int, default=2,
                    help="ticks compared bit-for-bit with the oracle before timing (rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ra_amd import abi, engine, shard, workload as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    G, N = args.groups, args.members
    K, Wm = args.steps, args.warmup
    T = Wm + K
    # this rank's shard of the global group id space (hash partition)
    my_groups = shard.local_group_ids(G * world, world, rank, per_rank=G)
    seed = (args.seed ^ (rank * 0x9E3779B97F4A7C15)) & ((1 << 64) - 1)

    eng = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=2, ring_capacity=1024)
    st0 = W.initial_states(G, N, seed)
    eng.set_state(0, st0)

    # a real (non-default) stream: the kernels, the HIP events and RCCL all run on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert sptr != 0

    # ---- pass 1 (untimed): synthesise the tick stream from the evolving device state ----
    t_gen = time.time()
    ticks = []
    scratch_dec = None
    for t in range(T):
        cur = eng.get_state()
        m = W.gen_tick(cur, N, t, seed, W.MIX_CONFIG3)
        ticks.append(m)
        dm = torch.from_numpy(m.view(np.uint8).reshape(-1)).to(dev)
        if scratch_dec is None or scratch_dec.numel() < len(m) * 64:
            scratch_dec = torch.empty(len(m) * 64 + 4096, dtype=torch.uint8, device=dev)
        eng.run_ticks_device(dm.data_ptr(), len(m), 1, scratch_dec.data_ptr(), stream=sptr)
        torch.cuda.synchronize()
        del dm
    gen_s = time.time() - t_gen
    width = max(len(m) for m in ticks)
    width = (width + 255) // 256 * 256
    host = np.zeros((T, width), dtype=abi.MSG_DTYPE)
    for t, m in enumerate(ticks):
        host[t, :len(m)] = m
    n_dec = np.array([len(m) for m in ticks], dtype=np.int64)     # non-NOP decisions per tick
    alg_bytes = np.array([W.algorithmic_bytes(m, N) for m in ticks], dtype=np.int64)
    d_msgs = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev)
    d_dec = torch.empty(T * width * 64, dtype=torch.uint8, device=dev)
    d_rpcs = torch.empty(width * max(N - 1, 1) * 56, dtype=torch.uint8, device=dev)  # rewritten every tick
    counts = n_dec.astype(np.uint32)
    lb_local = torch.empty(G * 32, dtype=torch.uint8, device=dev)
    lb_all = torch.empty(world * G * 32, dtype=torch.uint8, device=dev) if world > 1 else None
    tick_bytes = width * 64

    def run(t0, t1, with_snapshots=True):
        """Enqueue ticks [t0, t1) on the current torch stream."""
        t = t0
        while t < t1:
            nxt = min(t1, (t // SNAPSHOT_EVERY + 1) * SNAPSHOT_EVERY)
            eng.run_ticks_device(d_msgs.data_ptr() + t * tick_bytes, width, nxt - t,
                                 d_dec.data_ptr() + t * tick_bytes, d_rpcs.data_ptr(), sptr,
                                 tick_counts=counts[t:nxt])
            if with_snapshots and nxt % SNAPSHOT_EVERY == 0:
                eng.snapshot_device(lb_local.data_ptr(), sptr)
                if world > 1:
                    dist.all_gather_into_tensor(lb_all, lb_local)
            t = nxt

    # ---- correctness gate before timing (rank 0): first ticks bit-exact vs the oracle ----
    checked = 0
    if rank == 0 and args.check_ticks > 0:
        from oracle import oracle as O
        cpu = O.Oracle(G, N)
        cpu.set_state(0, st0)
        eng.set_state(0, st0)
        nchk = min(args.check_ticks, T)
        run(0, nchk, with_snapshots=False)
        torch.cuda.synchronize()
        got = d_dec[:nchk * tick_bytes].cpu().numpy().view(abi.DECISION_DTYPE).reshape(nchk, width)
        for t in range(nchk):
            nt = int(n_dec[t])
            want, _ = cpu.step_parallel(host[t, :nt])
            g = got[t, :nt]
            if g.tobytes() != want.tobytes():
                bad = int(np.flatnonzero((g.view(np.uint8).reshape(nt, 64) !=
                                          want.view(np.uint8).reshape(nt, 64)).any(axis=1))[0])
                raise SystemExit(f"PARITY FAILURE tick {t} decision {bad}: gpu={g[bad]} cpu={want[bad]}")
        assert eng.get_state().tobytes() == cpu.get_state().tobytes(), "state differs from the oracle"
        checked = nchk
        cpu.close()

    # ---- pass 2: reset, warm up, time exactly K ticks ----
    eng.set_state(0, st0)
    run(0, Wm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # the K timed ticks are captured once into a hipGraph (launch-bound inner loop: the eager host
    # launch rate is ~3.7 us per kernel on this box, the kernels themselves are shorter)
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            run(Wm, T)
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wall0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        run(Wm, T)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - wall0
    ev_ms = ev0.elapsed_time(ev1)
    elapsed = max(wall, ev_ms / 1e3)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([int(n_dec[Wm:].sum()), int(alg_bytes[Wm:].sum())], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_dec, total_bytes = int(tot[0].item()), int(tot[1].item())
    else:
        total_dec, total_bytes = int(n_dec[Wm:].sum()), int(alg_bytes[Wm:].sum())
    final_checksum = eng.state_checksum()

    # ---- cpu baseline (rank 0, N=1 only): the oracle on this box's host cores ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cpu = O.Oracle(G, N)
        ncpu = os.cpu_count() or 1
        sample = min(T, 16)
        # pick the thread count that is fastest on this box (OpenMP fork/join dominates small ticks)
        best_thr, best_rate = 1, 0.0
        for thr in sorted({1, 8, 16, 32, 64, ncpu}):
            if thr > ncpu:
                continue
            cpu.set_state(0, st0)
            t0 = time.perf_counter()
            nd = 0
            for t in range(min(sample, 4)):
                cpu.step_parallel(ticks[t], thr)
                nd += len(ticks[t])
            rate = nd / (time.perf_counter() - t0)
            if rate > best_rate:
                best_thr, best_rate = thr, rate
        threads = best_thr
        spent, done_dec, reps = 0.0, 0, 0
        while spent < args.cpu_seconds and reps < 256:
            cpu.set_state(0, st0)
            for t in range(sample):
                t0 = time.perf_counter()
                cpu.step_parallel(ticks[t], threads)
                spent += time.perf_counter() - t0
                done_dec += len(ticks[t])
            reps += 1
        cpu_baseline = {
            "value": done_dec / spent, "unit": "decisions/s", "cores": threads, "kind": "port",
            "sample": f"first {sample} ticks of the same stream x {reps} repetitions "
                      f"({done_dec} decisions, {spent:.1f} s of oracle time, OpenMP over messages, "
                      f"best of 1/8/16/32/64/{ncpu} threads on {ncpu} host cores)",
        }
        cpu.close()

    if rank == 0:
        per_launch_s = (ev_ms / 1e3) / K
        launch_bytes = float(alg_bytes[Wm:].mean())
        achieved = launch_bytes / per_launch_s / 1e9
        kinds = {}
        allm = np.concatenate(ticks[Wm:])
        for name, code in (("aer", abi.MSG_AER), ("aer_reply", abi.MSG_AER_REPLY),
                           ("request_vote", abi.MSG_REQUEST_VOTE), ("append", abi.MSG_APPEND),
                           ("written", abi.MSG_WRITTEN)):
            kinds[name] = round(float((allm["kind"] == code).mean()), 4)
        out = {
            "metric": "append_entries decisions/sec across N Raft groups; achieved HBM GB/s vs peak",
            "value": total_dec / elapsed,
            "unit": "decisions/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "configs[2]: 65536 groups x 5 members per GPU, mixed append_entries + "
                            "request_vote (5% term churn), device-resident message batches",
                "groups_per_gpu": G, "members": N, "decisions_per_tick": float(n_dec[Wm:].mean()),
                "tick_width": width, "message_mix": kinds,
                "leaderboard_allgather_every": SNAPSHOT_EVERY,
                "parallelism": f"hash-sharded groups x{world}, no data-path collective",
                "oracle_checked_ticks": checked, "state_checksum": f"{final_checksum:#018x}",
                "stream_generation_s": round(gen_s, 1),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "rgb_tick_kernel<5>",
                "algorithmic_bytes_per_launch": launch_bytes,
                "avg_launch_us": per_launch_s * 1e6,
            },
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    _ = my_groups


if __name__ == "__main__":
    main()
