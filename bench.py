#!/usr/bin/env python3
"""bench.py -- append_entries decisions/sec across N Raft groups on MI355X (BASELINE.json).

Workload (config.workload): BASELINE.json configs[2] -- 65 536 five-member Raft groups per GPU,
mixed append_entries_rpc / append_entries_reply / request_vote_rpc with 5 % term churn (5 % of
the groups per tick see a request_vote with term+1; deposed groups re-elect through
election_timeout / pre_vote / request_vote_result on the device), plus the {commands,_} appends
and {written,..} log events that keep the logs moving.  The stream is produced by the
device-side load generator (include/ra_gpu_batch_synth.h) from the evolving device state.

A "step" is one tick: one pass of the hot path (ONE launch of rgb_tick_kernel) over one batch
holding at most one message per server, messages already resident in HBM.  `value` counts real
decisions only (empty NOP slots of the dense tick layout are not decisions).

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
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise);
# must be in the environment before the HIP runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SNAPSHOT_EVERY = 16
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

LITERAL = {
    # (groups, members, seed, mix, gen_tick kwargs, initial_states kwargs)
    "2": dict(groups=4096, members=5, seed=0x5EED0002, mix="MIX_CONFIG2", gen=dict(groups_per_tick=256, housekeeping=False),
              init=dict(), what="configs[1]: 4096 groups x 5, 256 messages per tick, 50% follower append_entries / "
                                "50% leader success replies"),
    "3": dict(groups=65536, members=5, seed=0x5EED0003, mix="MIX_CONFIG3", gen=dict(housekeeping=False), init=dict(),
              what="configs[2] literal: 65536 groups x 5, ONE message per group per tick (65536 per tick), 70% reply ok / "
                   "20% append_entries / 5% reply failed / 5% request_vote term+1"),
    "5": dict(groups=65536, members=7, seed=0x5EED0005, mix="MIX_CONFIG5", gen=dict(backlog_mode=True),
              init=dict(backlog=1024, boundaries=(3, 6)),
              what="configs[4]: 65536 groups x 7, 1024-entry uncommitted backlogs over 3-6 term boundaries, "
                   "append_entries inside the backlog (prev_log_term wrong in 50%), failed replies driving the repair"),
}


def run_literal(name, ticks, torch, engine, W, abi, dev, local_rank, reps=3, on_train=None):
    """One literal SURVEY 8(d) configuration: the host generator (ra_amd/workload.gen_tick) produces tick t from
    the CHECKER's state after tick t-1 (the oracle is the state evolver here and nothing else), the stored ticks
    are applied on the device once untimed with EVERY decision and the final state compared with the oracle's,
    then replayed from the initial state inside one hipGraph between HIP events (best of `reps`)."""
    from oracle import oracle as O
    c = LITERAL[name]
    G, N, seed = c["groups"], c["members"], c["seed"]
    S = G * N
    st0 = W.initial_states(G, N, seed, **c["init"])
    cpu = O.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st0)
    msgs, decs = [], []
    t_host = time.perf_counter()
    for t in range(ticks):
        m = W.gen_tick(cpu.get_state(), N, t, seed, getattr(W, c["mix"]), **c["gen"])
        d, _ = cpu.step_parallel(m)
        msgs.append(m)
        decs.append(d)
    want_final = cpu.get_state()
    cpu.close()
    t_host = time.perf_counter() - t_host
    NK = abi.N_KINDS
    stride = max(len(m) for m in msgs)
    kc = np.zeros((ticks, NK), dtype=np.uint32)
    host = np.zeros(ticks * stride, dtype=abi.MSG_DTYPE)
    for t, m in enumerate(msgs):
        host[t * stride:t * stride + len(m)] = m
        kc[t] = np.bincount(m["kind"], minlength=NK)[:NK]
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    d_msgs = torch.from_numpy(host.view(np.uint8)).to(dev)
    d_dec = torch.zeros(ticks * stride * 64, dtype=torch.uint8, device=dev)
    d_rpcs = torch.empty(stride * max(N - 1, 1) * 56, dtype=torch.uint8, device=dev)
    eng = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=1, ring_capacity=64)
    try:
        def enqueue():
            eng.run_ticks_device(d_msgs.data_ptr(), stride, ticks, d_dec.data_ptr(), d_rpcs.data_ptr(), sptr,
                                 tick_counts=kc.sum(axis=1).astype(np.uint32), kind_counts=kc)
        # parity pass: every decision of every tick, then the whole final state
        eng.set_state(0, st0)
        enqueue()
        torch.cuda.synchronize()
        got = abi.expand_decisions(d_dec.cpu().numpy().view(abi.DECISION_DTYPE))    # (device streams hold compact records)
        checked = 0
        for t in range(ticks):
            g = got[t * stride:t * stride + len(msgs[t])]
            if g.tobytes() != decs[t].tobytes():
                bad = int(np.flatnonzero((g.view(np.uint8).reshape(-1, 64) != decs[t].view(np.uint8).reshape(-1, 64)).any(axis=1))[0])
                raise SystemExit(f"PARITY FAILURE config {name} tick {t} slot {bad}: msg={msgs[t][bad]} gpu={g[bad]} cpu={decs[t][bad]}")
            checked += len(g)
        assert eng.get_state().tobytes() == want_final.tobytes(), f"config {name}: final state differs from the oracle's"
        # timed replays
        g = torch.cuda.CUDAGraph()
        eng.set_state(0, st0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            enqueue()
        best = None
        for _ in range(reps):
            eng.set_state(0, st0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        # ---- the same ticks as ONE train launch (rgb_train_run_device): every tick in bucket order (class, shard,
        # success flag -- a stable sort of the generator's tick, untimed like the family order above), the per-server
        # sequence stamps order the ticks instead of kernel boundaries.  Every decision is compared with the oracle's
        # again (through the sort's permutation) and the final state with the oracle's.
        train = None
        if ticks <= 255:
            perms, bcs = [], np.zeros((ticks, engine.TRAIN_BUCKETS), dtype=np.uint32)
            host_t = np.zeros(ticks * stride, dtype=abi.MSG_DTYPE)
            for t, m in enumerate(msgs):
                bk = engine.train_bucket(m["kind"], m["flags"], m["server"], N)
                perm = np.argsort(bk, kind="stable")
                perms.append(perm)
                bcs[t] = np.bincount(bk, minlength=engine.TRAIN_BUCKETS)
                host_t[t * stride:t * stride + len(m)] = m[perm]
            d_msgs_t = torch.from_numpy(host_t.view(np.uint8)).to(dev)
            d_dec_t = torch.zeros(ticks * stride * 64, dtype=torch.uint8, device=dev)
            d_stamps = torch.zeros(ticks * stride, dtype=torch.uint8, device=dev)
            RING = 4
            d_rpcs_t = torch.empty(RING * stride * max(N - 1, 1) * 56, dtype=torch.uint8, device=dev)
            counts = np.array([len(m) for m in msgs], dtype=np.uint32)
            plan = eng.train_plan(bcs)

            def enqueue_train():
                eng.train_run_device(plan, 0, ticks, d_msgs_t.data_ptr(), d_stamps.data_ptr(), stride, d_dec_t.data_ptr(),
                                     d_rpcs_t.data_ptr(), RING, sptr)
            best_t = None
            # the train stamps are the PRODUCER's: the host that generated and bucket-sorted the ticks counts the
            # messages it has sent to every server (what rgb_submit does for a host batch); the sequence bytes of this
            # fresh engine start at 0 and only trains move them
            sent = np.zeros(S, dtype=np.uint8)
            for rep in range(reps + 1):                                 # rep 0: the parity pass
                eng.set_state(0, st0)
                h_st = np.zeros(ticks * stride, dtype=np.uint8)
                for t, m in enumerate(msgs):
                    srv = m["server"][perms[t]]
                    h_st[t * stride:t * stride + len(m)] = sent[srv]
                    sent[srv] += 1                                       # one message per server per tick; wraps mod 256
                d_stamps.copy_(torch.from_numpy(h_st))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                enqueue_train()
                e1.record(stream)
                torch.cuda.synchronize()
                eng.train_status()
                if rep == 0:
                    got_t = abi.expand_decisions(d_dec_t.cpu().numpy().view(abi.DECISION_DTYPE))
                    for t in range(ticks):
                        gt = got_t[t * stride:t * stride + len(msgs[t])]
                        if gt.tobytes() != decs[t][perms[t]].tobytes():
                            raise SystemExit(f"PARITY FAILURE config {name}, train launch, tick {t}")
                    assert eng.get_state().tobytes() == want_final.tobytes(), f"config {name}: train final state differs"
                else:
                    ms = e0.elapsed_time(e1)
                    best_t = ms if best_t is None else min(best_t, ms)
            train = {"us_per_tick": best_t * 1e3 / ticks, "blocks_per_tick": plan.blocks_per_tick}
            if on_train is not None:                      # tools/cfg5_probe.py: the last train launch's per-wavefront stamps
                on_train(eng, ticks, plan.blocks_per_tick)
            plan.close()
    finally:
        eng.close()
    n_dec = int(sum(len(m) for m in msgs))
    alg = int(sum(W.algorithmic_bytes(m, N) for m in msgs))
    per_tick_ms = best
    if train is not None and train["us_per_tick"] * ticks / 1e3 < best:
        best = train["us_per_tick"] * ticks / 1e3
    tk = kc.sum(axis=0)
    names = ["nop", "aer", "aer_reply", "request_vote", "vote_result", "written", "pipeline_rpcs", "append"]
    return {
        "workload": c["what"], "seed": hex(seed), "ticks": ticks, "decisions": n_dec,
        "decisions_per_tick": n_dec / ticks, "us_per_tick": best * 1e3 / ticks,
        "value": n_dec / (best / 1e3), "unit": "decisions/s",
        "algorithmic_bytes_per_launch": alg / ticks, "achieved_GBps": alg / (best / 1e3) / 1e9,
        "frac": alg / (best / 1e3) / 1e9 / HBM_PEAK_GBPS,
        "message_mix": {names[i]: round(float(tk[i]) / float(tk.sum()), 4) for i in range(1, len(names)) if tk[i]},
        "oracle_checked_decisions": checked, "oracle_checked_ticks": ticks, "final_state_equal": True,
        "host_generation_s": round(t_host, 1),
        "launch": "train" if best != per_tick_ms else "tick",
        "per_tick_launches": {"us_per_tick": per_tick_ms * 1e3 / ticks, "value": n_dec / (per_tick_ms / 1e3),
                              "frac": alg / (per_tick_ms / 1e3) / 1e9 / HBM_PEAK_GBPS},
        "train_launch": None if train is None else {**train, "value": n_dec / (train["us_per_tick"] * ticks / 1e6),
                                                    "frac": alg / (train["us_per_tick"] * ticks / 1e6) / 1e9 / HBM_PEAK_GBPS,
                                                    "oracle_checked_decisions": checked},
        "note": "every tick generated from the checker's state and checked decision by decision, in both forms: one "
                "launch per tick (hipGraph replay from the initial state) and all ticks as ONE train launch (ticks in "
                "bucket order, HIP events around the launch); device-resident batches, best of %d; the headline "
                "fields are the faster form" % reps,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--groups", type=int, default=65536, help="Raft groups per GPU")
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0003)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the PCIe-inclusive rgb_submit/rgb_collect measurement (rank 0, N=1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--generic-kernel", action="store_true",
                    help="the kind-generic kernel instead of the class-dispatch kernel")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check-ticks", type=int, default=3,
                    help="ticks (the first of the timed window's stream, i.e. on the AGED state) compared "
                         "bit-for-bit with the oracle before timing (rank 0)")
    ap.add_argument("--age", type=int, default=512,
                    help="untimed fast-forward: generator ticks applied before the warm-up/timed window, so that a "
                         "short run (--steps 20) times the long-run state (term-run tables of 2..16 runs, "
                         "compaction active) and not ticks 5..24 of a fresh one")
    ap.add_argument("--launch", choices=("train", "tick"), default="train",
                    help="train: the ticks of one leaderboard period run in ONE launch (rgb_train_run_device: per-server "
                         "sequence stamps instead of kernel boundaries); tick: one class-kernel launch per tick")
    ap.add_argument("--train-form", choices=("auto", "persistent"), default="auto",
                    help="auto: the dealt form where the calibration launch shows round-robin dispatch (every block "
                         "verifies it), else persistent; persistent: RGB_CFG_TRAIN_PERSISTENT")
    ap.add_argument("--hint", choices=("none", "state", "header"), default="header",
                    help="the generator's ordering hint (include/ra_gpu_batch_synth.h, rgb_synth_set_hint): none, the "
                         "owner's state name (round 4), + the owner's O(1) compare of the rpc header with fields it "
                         "holds (default); it only orders a tick, results are the same")
    ap.add_argument("--plan", choices=("producer", "host", "region"), default="producer",
                    help="where the train's row plan is built.  producer (default): ON THE DEVICE by the stream's producer -- "
                         "one rgb_train_plan_build_device kernel behind the generator, from the bucket counts it left in "
                         "device memory: nothing of the plan passes through the host, no copy of the counts, and no plan "
                         "kernel in front of a launch (the grid of a dealt launch is the rows bound of a tick); host: "
                         "rgb_train_plan_create* from a host copy of the counts (rounds 3-5); region: the device build as "
                         "one kernel per launch INSIDE the timed region, in front of the launch")
    ap.add_argument("--device-plan", action="store_true", help="= --plan region (round 5's flag)")
    ap.add_argument("--graph", action="store_true", help="capture the timed region into a hipGraph even when it is one or two launches")
    ap.add_argument("--snapshot-kernel", action="store_true",
                    help="train: one launch per leaderboard period with the snapshot KERNEL between the launches (the "
                         "round-3 form; default: the snapshots run as rows of launches of up to 255 ticks)")
    ap.add_argument("--snapshot-every", type=int, default=0,
                    help="leaderboard period in ticks (default 16 = SURVEY 8(d) config 4); a train covers one period")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3] as written: 262 144 groups x 5 in TOTAL, hashed (rgb_route / splitmix64) over "
                         "the --gpus ranks -- strong scaling, N=1 holds all of them; the leaderboard all-gather every 16 "
                         "ticks goes through the C entry point (rgb_leaderboard_allgather)")
    ap.add_argument("--total-groups", type=int, default=262144, help="--config4: groups over all ranks")
    ap.add_argument("--literal-ticks", type=int, default=32,
                    help="ticks of each literal SURVEY 8(d) configuration (configs 2, 3 and 5; host-generated, every "
                         "tick oracle-checked, then replayed and timed on the device); 0 = skip (rank 0, N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ra_amd import abi, engine, shard, workload as W
    if not os.path.exists(engine.LIB_PATH):
        # fresh checkout: local rank 0 compiles the HIP library in-tree (hipcc, gfx950), the others wait
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            engine.build()
        else:
            t_wait = time.time()
            while not os.path.exists(engine.LIB_PATH) and time.time() - t_wait < 900:
                time.sleep(2.0)
            time.sleep(2.0)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RGB_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU boxes)
    use_dist = world > 1 or bool(os.environ.get("RGB_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    global SNAPSHOT_EVERY
    if args.snapshot_every > 0:
        SNAPSHOT_EVERY = args.snapshot_every
    use_train = args.launch == "train" and not args.generic_kernel
    G, N = args.groups, args.members
    if args.config4 and args.steps == 1000:
        args.steps, args.warmup = 192, 16          # 1.3 M servers: 84 MB per tick of messages, as much of decisions
    if args.config4:
        # configs[3]: a FIXED population hashed over the ranks (strong scaling); this rank's share of the global ids
        my_groups = shard.local_group_ids(args.total_groups, world, rank)
        G = len(my_groups)
        args.no_cpu_baseline = args.no_host_path = True
        args.literal_ticks = 0
    else:
        # this rank's shard of the global group-id space (hash partition), 65 536 per rank
        my_groups = shard.local_group_ids(G * world, world, rank, per_rank=G)
    S = G * N
    K, Wm = args.steps, args.warmup
    T = Wm + K
    NK = abi.N_KINDS
    seed = (args.seed ^ (rank * 0x9E3779B97F4A7C15)) & ((1 << 64) - 1)

    eng = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=2, ring_capacity=1024,
                            flags=abi.CFG_TRAIN_PERSISTENT if args.train_form == "persistent" else 0)
    st0 = W.initial_states(G, N, seed)
    eng.set_state(0, st0)
    hint_level = {"none": 0, "state": 1, "header": 2}[args.hint]
    if hasattr(engine.lib(), "rgb_synth_set_hint"):
        eng.synth_set_hint(hint_level)
    else:                                     # (RGB_LIB = a build of round 4: state-name hint only)
        hint_level = min(hint_level, 1)

    comm = None
    if use_dist:
        idt = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).to(dev)
        dist.broadcast(idt, 0)            # the id travels by the host's own means (here torch.distributed)
        comm = engine.Comm(eng, idt.cpu().numpy().tobytes(), world, rank)
    # a real (non-default) stream: the kernels, the HIP events and RCCL all run on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert sptr != 0
    tick_bytes = S * 64
    d_msgs = torch.empty(T * tick_bytes, dtype=torch.uint8, device=dev)
    d_dec = torch.zeros(T * tick_bytes, dtype=torch.uint8, device=dev)     # zeroed: the unwritten half of a compact
                                                                           # record compares equal between the two passes
    RPC_RING = 4 if use_train else 1       # ticks of one train overlap: tick k of a launch writes region k mod 4
    d_rpcs = torch.empty(RPC_RING * S * max(N - 1, 1) * 56, dtype=torch.uint8, device=dev)   # rewritten every tick
    d_bc = torch.zeros(T * engine.TRAIN_BUCKETS, dtype=torch.int32, device=dev)     # messages per train bucket
    d_kc = torch.zeros(T * NK, dtype=torch.int32, device=dev)
    d_n = torch.zeros(T, dtype=torch.int32, device=dev)           # real size of every tick
    # the leaderboard all-gather goes through the C entry point (rgb_leaderboard_allgather: ncclAllGather behind the
    # boundary, on the launch stream).  Every rank contributes the same number of rows: hash sharding leaves the shards
    # unequal (--config4), so the rows are padded to the largest shard
    lb_rows = G
    if use_dist:
        gm = torch.tensor([G], dtype=torch.int64, device=dev)
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        lb_rows = int(gm.item())
    lb_local = torch.zeros(lb_rows * 32, dtype=torch.uint8, device=dev)
    lb_all = torch.empty(world * lb_rows * 32, dtype=torch.uint8, device=dev) if use_dist else None

    # ---- pass 0 (untimed): age the state.  `age` generator ticks are applied without being kept; the aged
    # state is the starting point of both the generation pass and the timed replay ----
    A = max(args.age, 0)
    t_gen = time.time()
    for t in range(A):
        eng.synth_tick_device(seed, t, d_msgs.data_ptr(), d_kc.data_ptr(), d_n.data_ptr(), sptr)
        eng.synth_apply_tick_device(d_msgs.data_ptr(), S, d_dec.data_ptr(), d_rpcs.data_ptr(), sptr)
    torch.cuda.synchronize()
    d_kc.zero_()
    d_dec.zero_()            # (the ageing ticks wrote here: the unwritten half of a compact record must compare equal)
    st_aged = eng.get_state() if A else st0
    # ---- pass 1 (untimed): generate tick A+t from the device state, then apply it.  The generator is the PRODUCER
    # of the stream: it writes every tick in bucket order and, beside every message, the train stamp -- its own count
    # of the messages it has addressed to that server (what rgb_submit's bucketing pass does for a host batch).  No
    # pass over the finished stream is needed before the timed replay ----
    d_stamps = torch.zeros(T * S, dtype=torch.uint8, device=dev) if use_train else None
    old_build = not hasattr(engine.lib(), "rgb_synth_tick_stamped_device")     # RGB_LIB=<round-3 build>: A/B timing only
    # Leaderboard snapshots INSIDE the train (one rank: nothing happens between two leaderboard periods, so a launch
    # covers several of them -- the snapshot's rows run as rows of the launch, ordered like one more message to every
    # server).  With --gpus N the periods stay one launch each: the all-gather sits between them.
    snap_in_train = (use_train and not use_dist and not args.snapshot_kernel
                     and hasattr(engine.lib(), "rgb_train_run_snap_device"))
    n_bound = T // SNAPSHOT_EVERY                                   # boundaries k = 0.. in front of tick (k + 1) * every
    seqb = eng.train_seq_bytes() if snap_in_train else 0
    d_snap_stamps = torch.zeros(max(n_bound, 1) * max(seqb, 1), dtype=torch.uint8, device=dev) if snap_in_train else None
    lb_ref = torch.zeros(max(n_bound, 1) * G * 32, dtype=torch.uint8, device=dev) if snap_in_train else None
    lb_got = torch.zeros(max(n_bound, 1) * G * 32, dtype=torch.uint8, device=dev) if snap_in_train else None
    for t in range(T):
        if snap_in_train and t and t % SNAPSHOT_EVERY == 0:
            k = t // SNAPSHOT_EVERY - 1
            eng.synth_snapshot_mark_device(d_snap_stamps.data_ptr() + k * seqb, sptr)     # the producer's side of it
            eng.snapshot_device(lb_ref.data_ptr() + k * G * 32, sptr)                     # what the rows must be
        if old_build:
            eng.synth_tick_buckets_device(seed, A + t, d_msgs.data_ptr() + t * tick_bytes, d_kc.data_ptr() + t * NK * 4,
                                          d_n.data_ptr() + t * 4, d_bc.data_ptr() + t * engine.TRAIN_BUCKETS * 4, sptr)
        else:
            eng.synth_tick_stamped_device(seed, A + t, d_msgs.data_ptr() + t * tick_bytes, d_kc.data_ptr() + t * NK * 4,
                                          d_n.data_ptr() + t * 4, d_bc.data_ptr() + t * engine.TRAIN_BUCKETS * 4,
                                          d_stamps.data_ptr() + t * S if use_train else 0, sptr)
        eng.synth_apply_tick_device(d_msgs.data_ptr() + t * tick_bytes, S, d_dec.data_ptr() + t * tick_bytes,
                                    d_rpcs.data_ptr(), sptr)
    torch.cuda.synchronize()
    gen_s = time.time() - t_gen
    checksum_pass1 = eng.state_checksum()
    kc = d_kc.cpu().numpy().reshape(T, NK).astype(np.int64)
    n_dec = kc[:, 1:].sum(axis=1)                                  # decisions per tick
    assert np.array_equal(n_dec, d_n.cpu().numpy().astype(np.int64))
    counts = n_dec.astype(np.uint32)
    alg_bytes = W.algorithmic_bytes_from_counts(kc, N)
    n_runs_hist = np.bincount(st_aged["n_runs"], minlength=17).tolist()

    def tick_msgs(t):
        nt = int(n_dec[t])
        return d_msgs[t * tick_bytes:t * tick_bytes + nt * 64].cpu().numpy().view(abi.MSG_DTYPE)

    # ---- correctness gate (rank 0): the first ticks bit-for-bit against the oracle ----
    checked = 0
    first_ticks = []
    if rank == 0 and (args.check_ticks > 0 or not args.no_cpu_baseline or not args.no_host_path):
        n_keep = max(args.check_ticks, 0 if args.no_cpu_baseline else 16, 0 if args.no_host_path else 12)
        first_ticks = [tick_msgs(t) for t in range(min(n_keep, T))]
        if args.check_ticks > 0:
            from oracle import oracle as O
            cpu = O.Oracle(G, N, max_runs=16)
            cpu.set_state(0, st_aged)
            for t in range(min(args.check_ticks, T)):
                want, _ = cpu.step_parallel(first_ticks[t])
                nt = int(n_dec[t])
                got = abi.expand_decisions(d_dec[t * tick_bytes:t * tick_bytes + nt * 64].cpu().numpy().view(abi.DECISION_DTYPE))
                if got.tobytes() != want.tobytes():
                    bad = int(np.flatnonzero((got.view(np.uint8).reshape(nt, 64) !=
                                              want.view(np.uint8).reshape(nt, 64)).any(axis=1))[0])
                    raise SystemExit(f"PARITY FAILURE tick {t} slot {bad}: msg={first_ticks[t][bad]} "
                                     f"gpu={got[bad]} cpu={want[bad]}")
                checked += 1
            cpu.close()

    # ---- train mode: the plan of every tick (host) and the sequence stamps of the whole stream, counted from
    # the aged state (device, untimed: the order rgb_submit's bucketing would establish on the host path) ----
    plan = d_dec2 = None
    plan_host_ms = None
    plan_mode = "region" if args.device_plan else args.plan
    if use_train and plan_mode != "host" and not hasattr(engine.lib(), "rgb_train_plan_build_device"):
        plan_mode = "host"                                         # (RGB_LIB=<a build of an earlier ABI>)
    device_plan = use_train and plan_mode == "region"
    producer_plan = use_train and plan_mode == "producer"
    if device_plan or producer_plan:
        # nothing of the plan passes through the host.  region: an empty plan now, every launch's ticks are built by a
        # kernel in front of it, on the launch stream, inside the timed region (launch_ticks).  producer: the whole
        # stream's plan by ONE kernel here, behind the generator that left the counts in device memory -- the producer
        # of a device-resident stream hands over messages, stamps AND plan; the timed region is launches only
        plan = engine.TrainPlan(eng, None, snapshot_every=SNAPSHOT_EVERY if snap_in_train else 0, device_ticks=T)
        plan_host_ms = 0.0
        if producer_plan:
            plan.build_device(0, T, d_bc.data_ptr(), sptr)
            if hasattr(plan, "fit") and hasattr(engine.lib(), "rgb_train_plan_fit"):
                plan.fit(0, T, sptr)      # four bytes per tick come back (its rows): the launches' grids are the rows, not the bound
            torch.cuda.synchronize()
        d_dec2 = torch.zeros(T * tick_bytes, dtype=torch.uint8, device=dev)
    elif use_train:
        buckets = d_bc.cpu().numpy().reshape(T, engine.TRAIN_BUCKETS).astype(np.uint32)
        assert np.array_equal(buckets.sum(axis=1), counts)
        if os.environ.get("RGB_TRAIN_LEAD"):     # tuning probe: "class:lead,..." in ticks (tools/gpu_ab.sh)
            import ctypes as C
            lead = np.zeros(15, dtype=np.float32)
            for kv in os.environ["RGB_TRAIN_LEAD"].split(","):
                c, v = kv.split(":"); lead[int(c)] = float(v)
            engine.lib().rgb_train_set_lead(lead.ctypes.data_as(C.c_void_p))
        t_plan = time.perf_counter()
        plan = (eng.train_plan_snap(buckets, SNAPSHOT_EVERY) if snap_in_train
                else eng.train_plan(buckets))                                 # host: 256 bucket counts per tick -> row order
        plan_host_ms = (time.perf_counter() - t_plan) * 1e3
        d_dec2 = torch.zeros(T * tick_bytes, dtype=torch.uint8, device=dev)   # pass 1's decisions stay for comparison

    def launch_ticks(t, nxt):
        """ticks [t, nxt) on the stream: ONE train launch, or one class-kernel launch per tick"""
        if device_plan:
            plan.build_device(t, nxt - t, d_bc.data_ptr() + t * engine.TRAIN_BUCKETS * 4, sptr)
        if snap_in_train:
            eng.train_run_snap_device(plan, t, nxt - t, d_msgs.data_ptr(), d_stamps.data_ptr(), S, d_dec2.data_ptr(),
                                      d_rpcs.data_ptr(), RPC_RING, d_snap_stamps.data_ptr(), lb_got.data_ptr(), sptr)
        elif use_train:
            eng.train_run_device(plan, t, nxt - t, d_msgs.data_ptr(), d_stamps.data_ptr(), S, d_dec2.data_ptr(),
                                 d_rpcs.data_ptr(), RPC_RING, sptr)
        else:
            eng.run_ticks_device(d_msgs.data_ptr() + t * tick_bytes, S, nxt - t,
                                 d_dec.data_ptr() + t * tick_bytes, d_rpcs.data_ptr(), sptr,
                                 tick_counts=counts[t:nxt],
                                 kind_counts=None if args.generic_kernel else kc[t:nxt].astype(np.uint32))

    # a launch: one leaderboard period -- or, with the snapshots inside the train, as many whole periods as a launch
    # holds (255 ticks); a boundary that ends a launch is taken behind it (rgb_snapshot_train_device: the snapshot
    # kernel + the sequence bytes advanced, what the rows inside a launch do group by group)
    # (a server's sequence byte must not come round within one launch: ticks + snapshots inside it <= 255)
    WIN = (255 * SNAPSHOT_EVERY // (SNAPSHOT_EVERY + 1)) // SNAPSHOT_EVERY * SNAPSHOT_EVERY if snap_in_train else SNAPSHOT_EVERY
    if snap_in_train and WIN == 0:
        raise SystemExit("--snapshot-every > 254 needs --snapshot-kernel")

    def segments(t0, t1):
        t = t0
        while t < t1:
            nxt = min(t1, (t // WIN + 1) * WIN)
            yield t, nxt
            t = nxt

    def run_segment(t, nxt):
        launch_ticks(t, nxt)
        if nxt % SNAPSHOT_EVERY == 0:
            if snap_in_train:
                eng.snapshot_train_device(lb_got.data_ptr() + (nxt // SNAPSHOT_EVERY - 1) * G * 32, sptr)
            else:
                eng.snapshot_device(lb_local.data_ptr(), sptr)

    def run(t0, t1):
        """Enqueue ticks [t0, t1) on the stream."""
        for t, nxt in segments(t0, t1):
            run_segment(t, nxt)
            if use_dist and nxt % SNAPSHOT_EVERY == 0:
                comm.allgather_leaderboard(lb_local.data_ptr(), lb_rows, lb_all.data_ptr(), sptr)

    # ---- pass 2: back to the aged state, warm up, time exactly K ticks ----
    # (rgb_upload_state and the per-tick launches of pass 1 leave the servers' sequence bytes where they were when the
    # generator started counting, so the stream's stamps are the ones this replay needs)
    eng.set_state(0, st_aged)
    if use_train and old_build:
        eng.train_stamp_device(d_msgs.data_ptr(), d_stamps.data_ptr(), S, counts, sptr)
    run(0, Wm)
    torch.cuda.synchronize()

    # The timed ticks are captured into hipGraphs, one per leaderboard period (16 ticks + the
    # snapshot kernel): the inner loop is launch-bound (the eager host launch rate is ~3.7 us per
    # kernel on this box).  The RCCL all-gather stays outside the graphs, on the same stream.
    graphs = None
    # a timed region of one or two train launches (the driver's 20-step form: ONE launch) goes eager: a graph launch
    # costs more than the two kernel launches it replaces (same box, tools/r04_graph_ab.sh: 18.5 us per tick by events
    # through a graph, 17.8 eager).  --graph forces the graph
    n_seg = sum(1 for _ in segments(Wm, T))
    want_graph = not args.no_graph and (args.graph or use_dist or not use_train or n_seg > 2)
    if want_graph:
        try:
            graphs = []
            if use_dist:
                for t, nxt in segments(Wm, T):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        run_segment(t, nxt)
                    graphs.append((g, nxt))
            else:
                # one rank: nothing happens between the leaderboard periods, so the whole timed region is ONE
                # graph launch (a 20-step run pays one launch latency instead of two)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for t, nxt in segments(Wm, T):
                        run_segment(t, nxt)
                graphs.append((g, T))
            torch.cuda.synchronize()
        except Exception as e:                      # pragma: no cover - fall back to eager launches
            print(f"[bench] hipGraph capture failed ({e!r}); eager launches", file=sys.stderr)
            graphs = None

    ag_events = []      # (before, after) of every leaderboard all-gather of the timed region, on the launch stream

    def allgather():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        comm.allgather_leaderboard(lb_local.data_ptr(), lb_rows, lb_all.data_ptr(), sptr)
        e1.record(stream)
        ag_events.append((e0, e1))

    def timed():
        if graphs is not None:
            for g, nxt in graphs:
                g.replay()
                if use_dist and nxt % SNAPSHOT_EVERY == 0:
                    allgather()
        else:
            for t, nxt in segments(Wm, T):
                run_segment(t, nxt)
                if use_dist and nxt % SNAPSHOT_EVERY == 0:
                    allgather()

    TPL = max(nxt - t for t, nxt in segments(Wm, T)) if use_train else 1      # ticks of the timed region's longest launch
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (a torch event creates its HIP event at the first record -- 12-15 us of host time that sat inside the timed
    # region of the driver's 20-step form, profiles/r05_wall_breakdown.txt: both events are recorded once before)
    ev0.record(stream); ev1.record(stream)
    torch.cuda.synchronize()
    wall0 = time.perf_counter()
    ev0.record(stream)
    wall_a = time.perf_counter()
    timed()
    wall_b = time.perf_counter()
    ev1.record(stream)
    wall_c = time.perf_counter()
    torch.cuda.synchronize()
    wall_d = time.perf_counter()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - wall0
    # where the host's share of the timed region goes (us): event record | enqueue of the launches | event record | wait
    wall_breakdown = {"record0_us": round((wall_a - wall0) * 1e6, 1), "enqueue_us": round((wall_b - wall_a) * 1e6, 1),
                      "record1_us": round((wall_c - wall_b) * 1e6, 1), "wait_us": round((wall_d - wall_c) * 1e6, 1)}
    ev_ms = ev0.elapsed_time(ev1)
    elapsed = max(wall, ev_ms / 1e3)
    train_info = None
    if use_train:
        flags, xcc = eng.train_status(check=False)
        if flags:
            raise SystemExit(f"TRAIN LAUNCH FAILED: flags={flags} (1 = blocks of a shard on different XCDs, 2 = a "
                             f"dependency did not commit within the spin bound); xcc of shards {xcc.tolist()}")
        # every decision of every tick of the replay against the generation pass (per-tick launches)
        for t in range(T if not os.environ.get("RGB_BENCH_NOCHECK") else 0):   # NOCHECK: timing probes of broken variants
            nb = int(n_dec[t]) * 64
            if not torch.equal(d_dec2[t * tick_bytes:t * tick_bytes + nb], d_dec[t * tick_bytes:t * tick_bytes + nb]):
                raise SystemExit(f"PARITY FAILURE: train decisions of tick {t} differ from the per-tick launches")
        # every leaderboard snapshot taken inside (or between) the launches against the snapshot kernel of pass 1
        snaps_checked = 0
        if snap_in_train and not os.environ.get("RGB_BENCH_NOCHECK"):
            for k in range(n_bound):
                if (k + 1) * SNAPSHOT_EVERY >= T:
                    break                                            # (the boundary behind the last tick has no reference)
                a = lb_got[k * G * 32:(k + 1) * G * 32]; b = lb_ref[k * G * 32:(k + 1) * G * 32]
                if not torch.equal(a, b):
                    raise SystemExit(f"PARITY FAILURE: leaderboard snapshot {k} (in front of tick {(k + 1) * SNAPSHOT_EVERY}) "
                                     "differs from the snapshot kernel between per-tick launches")
                snaps_checked += 1
        # how many decisions of a timed tick went out in the 32-byte compact form, and which kinds stayed full
        tW = min(Wm, T - 1)
        rawW = d_dec2[tW * tick_bytes:tW * tick_bytes + int(n_dec[tW]) * 64].cpu().numpy().view(abi.DECISION_DTYPE)
        isc = (rawW["flags"] & abi.F_COMPACT) != 0
        full_by_kind = np.bincount(rawW["kind"][~isc], minlength=NK)[:NK]
        compact_info = {"fraction": round(float(isc.mean()), 4),
                        "decision_bytes_per_tick": int(isc.sum()) * 32 + int((~isc).sum()) * 64,
                        "full_records_by_kind": {str(k): int(v) for k, v in enumerate(full_by_kind) if v}}
        train_info = {"ticks_per_launch": TPL, "blocks_per_tick": plan.blocks_per_tick,
                      "form": eng.train_form(),
                      "leaderboard_snapshots": ("rows of the launch (rgb_train_run_snap_device): ordered per server like one "
                                                "more message; a boundary that ends a launch: rgb_snapshot_train_device"
                                                if snap_in_train else "rgb_snapshot_device between the launches"),
                      "leaderboard_snapshots_compared_with_snapshot_kernel": snaps_checked,
                      "compact_decisions": compact_info,
                      "plan": ("built on the device inside the timed region (rgb_train_plan_build_device: one kernel per "
                               "launch in front of it, from the generator's bucket counts in device memory); grid = the rows bound of a tick"
                               if device_plan else
                               "built on the device by the stream's producer (ONE rgb_train_plan_build_device kernel behind the "
                               "generator, from the bucket counts it left in device memory; no host copy of the counts, no plan "
                               "kernel in the timed region); the host is told the rows of every tick (rgb_train_plan_fit: 4 bytes "
                               "per tick) and sizes the launches' grids by them"
                               if producer_plan else "built on the host before the timed region from the bucket counts "
                               "(rgb_train_plan_create*): the dealt form needs the rows of a tick for its grid"),
                      "ordering_hint": {0: "none", 1: "owner's state name",
                                        2: "owner's state name + its O(1) compare of the rpc header with current_term, "
                                           "leader_id, last index / term (rgb_synth_set_hint 2); orders the tick only"}[hint_level],
                      "stamps": "written by the stream's producer (rgb_synth_tick_stamped_device) with the messages" +
                                ("; the row plan by the producer as well, on the device" if producer_plan else
                                 ": nothing of the train's input preparation is outside the timed region except the "
                                 "row plan (256 bucket counts per tick)"),
                      "plan_host_ms_total": round(plan_host_ms, 3), "plan_host_us_per_tick": round(plan_host_ms * 1e3 / T, 2),
                      "xcd_of_shard": [int(v) for v in xcc], "decisions_compared_with_per_tick_launches": int(n_dec.sum())}
    checksum_pass2 = eng.state_checksum()
    # the row plan on the device (rgb_train_plan_build_device): what building the timed region's plan costs as ONE kernel
    # from the generator's bucket counts in device memory, and that its tables are the host's bit for bit (every tick)
    device_plan_info = None
    if producer_plan and rank == 0:
        # behind the timed region: the producer's plan against the host's merge of the same counts, every tick
        buckets = d_bc.cpu().numpy().reshape(T, engine.TRAIN_BUCKETS).astype(np.uint32)
        assert np.array_equal(buckets.sum(axis=1), counts)
        hpl = (eng.train_plan_snap(buckets, SNAPSHOT_EVERY) if snap_in_train else eng.train_plan(buckets))
        same = sum(1 for t in range(T) if all(np.array_equal(x, y) for x, y in zip(hpl.download(t), plan.download(t))))
        hpl.close()
        if same != T:
            raise SystemExit(f"PLAN MISMATCH: the device-built plan differs from the host's in {T - same} of {T} ticks")
        train_info["device_plan"] = {"ticks_whose_tables_equal_the_host_plan": same, "ticks": T}
    if use_train and not device_plan and not producer_plan and rank == 0 and hasattr(engine.lib(), "rgb_train_plan_build_device"):
        dpl = engine.TrainPlan(eng, None, snapshot_every=SNAPSHOT_EVERY if snap_in_train else 0, device_ticks=T)
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dpl.build_device(Wm, K, d_bc.data_ptr() + Wm * engine.TRAIN_BUCKETS * 4, sptr)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            best = us if best is None or us < best else best
        dpl.build_device(0, T, d_bc.data_ptr(), sptr)
        torch.cuda.synchronize()
        same = 0
        for t in range(T):
            a, b = plan.download(t), dpl.download(t)
            if all(np.array_equal(x, y) for x, y in zip(a, b)):
                same += 1
        dpl.close()
        device_plan_info = {"build_kernel_us_for_the_timed_ticks": round(best, 2), "timed_ticks": K,
                            "ticks_whose_tables_equal_the_host_plan": same, "ticks": T,
                            "note": "--device-plan runs the timed region from it (plan kernel inside the "
                                    "region); this line's region uses the host-built plan in the dealt form"}
        if same != T:
            raise SystemExit(f"PLAN MISMATCH: the device-built plan differs from the host's in {T - same} of {T} ticks")
        train_info["device_plan"] = device_plan_info
    assert checksum_pass2 == checksum_pass1 or os.environ.get("RGB_BENCH_NOCHECK"), "replay diverged from the generation pass"
    per_rank = None
    if use_dist:
        # every rank's own clock and its all-gathers' own time: the scaling curve can be read rank by rank
        ag_us = sum(a.elapsed_time(b) for a, b in ag_events) * 1e3
        mine = torch.tensor([elapsed * 1e3 / K, ev_ms / K, ag_us / max(len(ag_events), 1), float(len(ag_events))],
                            dtype=torch.float64, device=dev)
        allr = torch.empty(world * 4, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        rr = allr.cpu().numpy().reshape(world, 4)
        per_rank = {"ms_per_step_wall": [round(float(v), 6) for v in rr[:, 0]],
                    "ms_per_step_hip_events": [round(float(v), 6) for v in rr[:, 1]],
                    "allgather_us_per_call": [round(float(v), 2) for v in rr[:, 2]],
                    "allgathers_in_timed_region": int(rr[0, 3]),
                    "allgather_bytes_per_rank": int(lb_rows * 32),
                    "allgather": "rgb_leaderboard_allgather (C ABI; ncclAllGather over xGMI on the launch stream)"}
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([int(n_dec[Wm:].sum()), int(alg_bytes[Wm:].sum())], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_dec = int(tot[0].item())
    else:
        total_dec = int(n_dec[Wm:].sum())

    # ---- cpu baseline (rank 0, N=1 only): the oracle on this box's host cores, 1 thread AND every usable
    # core (both lines, core count and CPU model stated) ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cpu = O.Oracle(G, N, max_runs=16)
        ncpu = os.cpu_count() or 1
        # a container may see every host core and still be limited to a few by its CPU quota
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
        except Exception:
            quota = None
        try:
            affinity = len(os.sched_getaffinity(0))
        except Exception:
            affinity = ncpu
        model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except Exception:
            pass
        sample = first_ticks[:16]
        usable = max(1, min(ncpu, affinity, quota or ncpu))

        def rate(threads, seconds):
            spent, done, reps = 0.0, 0, 0
            while spent < seconds and reps < 256:
                cpu.set_state(0, st_aged)
                for m in sample:
                    t0 = time.perf_counter()
                    cpu.step_parallel(m, threads)
                    spent += time.perf_counter() - t0
                    done += len(m)
                reps += 1
            return done / spent, done, spent, reps

        one, d1, s1, r1 = rate(1, args.cpu_seconds / 2)
        allc, da, sa, ra = rate(usable, args.cpu_seconds / 2)
        cpu_baseline = {
            "value": max(one, allc), "unit": "decisions/s", "cores": usable if allc >= one else 1, "kind": "port",
            "one_thread": one, "all_cores": allc, "all_cores_threads": usable, "cpu_model": model,
            "sample": f"first {len(sample)} ticks of the timed stream (aged state) x {r1}+{ra} repetitions: "
                      f"{d1} decisions in {s1:.1f} s on 1 thread, {da} decisions in {sa:.1f} s on {usable} threads "
                      f"(OpenMP static over the tick's messages); host cores {ncpu}, affinity {affinity}, "
                      f"cgroup cpu quota {quota if quota else 'none'}; the Erlang reference itself: n/a (no OTP here)",
        }
        cpu.close()

    # ---- host path (rank 0, N=1 only): the same ticks through rgb_submit -> kernel -> rgb_collect,
    # host buffers in, host buffers out (pinned ring, PCIe both ways).  Reported beside `value`,
    # never as `value`. ----
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path and first_ticks:
        BATCH = 131072
        eng_h = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=4, ring_capacity=BATCH)
        bufs = (np.empty(BATCH, dtype=abi.DECISION_DTYPE), np.empty(BATCH * max(N - 1, 1), dtype=abi.RPC_DTYPE))
        best = 0.0
        for rep in range(3):
            eng_h.set_state(0, st_aged)
            pending, nd = 0, 0
            t0 = time.perf_counter()
            for m in first_ticks[:12]:
                for i in range(0, len(m), BATCH):
                    while pending >= 3:
                        eng_h.collect(out=bufs); pending -= 1
                    eng_h.submit(m[i:i + BATCH]); pending += 1; nd += len(m[i:i + BATCH])
            while pending:
                eng_h.collect(out=bufs); pending -= 1
            best = max(best, nd / (time.perf_counter() - t0))
        # the same with rgb_collect_view / rgb_release (ABI v9): the consumer reads decisions and rpc records where the
        # device wrote them (the pinned slot), nothing is copied out
        best_v = 0.0
        for rep in range(3):
            eng_h.set_state(0, st_aged)
            pending, nd = 0, 0
            t0 = time.perf_counter()
            for m in first_ticks[:12]:
                for i in range(0, len(m), BATCH):
                    while pending >= 3:
                        eng_h.release(eng_h.collect_view()[3]); pending -= 1
                    eng_h.submit(m[i:i + BATCH]); pending += 1; nd += len(m[i:i + BATCH])
            while pending:
                eng_h.release(eng_h.collect_view()[3]); pending -= 1
            best_v = max(best_v, nd / (time.perf_counter() - t0))
        eng_h.close()
        host_path = {"value": best, "unit": "decisions/s", "batch": BATCH,
                     "note": "first 12 ticks through rgb_submit/rgb_collect from ONE thread (ctypes caller, pinned ring, "
                             "PCIe both ways, 64-B message in / 64-B decision + compacted rpc records out, both written "
                             "into the pinned slot by the device), best of 3",
                     "collect_view": {"value": best_v, "unit": "decisions/s",
                                      "note": "the same from one thread with rgb_collect_view + rgb_release: results read in "
                                              "place, no copy out of the slot"}}
        # four producer threads and two consumer threads on one context (the boundary's threading contract: producers
        # prepare their batches in parallel, the stream's work is enqueued in ticket order, consumers copy different
        # batches at once).  The batches of the first 12 ticks are dealt round robin to the producers, so the order
        # in which they reach the device is not the sequential one: a throughput figure, not a parity run
        try:
            import threading
            P_THREADS, C_THREADS = 4, 3
            eng_p = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=8, ring_capacity=BATCH)
            chunks = [m[i:i + BATCH] for m in first_ticks[:12] for i in range(0, len(m), BATCH)]
            best_p = 0.0
            for rep in range(3):
                eng_p.set_state(0, st_aged)
                todo = len(chunks)
                got = [0] * C_THREADS
                lock = threading.Lock()
                taken = [0]

                def producer(k):
                    for c in chunks[k::P_THREADS]:
                        while True:
                            try:
                                eng_p.submit(c); break
                            except engine.RgbError as e:
                                if e.code != abi.E_FULL: raise
                                time.sleep(0.0002)

                def consumer(k):
                    bufs_c = (np.empty(BATCH, dtype=abi.DECISION_DTYPE), np.empty(BATCH * max(N - 1, 1), dtype=abi.RPC_DTYPE))
                    while True:
                        with lock:
                            if taken[0] >= todo: return
                            taken[0] += 1
                        while True:
                            try:
                                d, _r, _t = eng_p.collect(out=bufs_c); got[k] += len(d); break
                            except engine.RgbError as e:
                                if e.code != abi.E_EMPTY: raise
                                eng_p.wait(50)

                ths = [threading.Thread(target=producer, args=(k,)) for k in range(P_THREADS)] + \
                      [threading.Thread(target=consumer, args=(k,)) for k in range(C_THREADS)]
                t0 = time.perf_counter()
                for t in ths: t.start()
                for t in ths: t.join()
                dt = time.perf_counter() - t0
                assert sum(got) == sum(len(c) for c in chunks)
                best_p = max(best_p, sum(got) / dt)
            eng_p.close()
            host_path["threads4"] = {"value": best_p, "unit": "decisions/s", "producer_threads": P_THREADS,
                                     "consumer_threads": C_THREADS, "batch": BATCH, "ring_slots": 8,
                                     "note": "the same 12 ticks in 131072-message batches, dealt round robin to 4 Python "
                                             "threads calling rgb_submit (ctypes releases the GIL), 3 threads in rgb_collect; best of 3"}
        except Exception as e:                                              # noqa: BLE001 - reported, not raised
            host_path["threads4"] = {"error": f"{type(e).__name__}: {e}"}
        # the normal shape of a real batch: several messages per server in ONE submit (a leader's N-1 replies arrive
        # together).  Four consecutive ticks per batch = four sub-tick rounds: fused into one train launch, and -- same
        # batches, the default since round 5 -- one launch per round (fused: RGB_CFG_SUBMIT_TRAINS)
        try:
            RB = 1 << 20
            big = [np.concatenate(first_ticks[i:i + 4]) for i in range(0, 12, 4)]
            big = [b for b in big if len(b) <= RB]
            rounds4 = {}
            for label, flags in (("fused_train", getattr(abi, "CFG_SUBMIT_TRAINS", 0)), ("launch_per_round", abi.CFG_ROUNDS_PER_LAUNCH)):
                eng_r = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=2, ring_capacity=RB, flags=flags)
                bufs_r = (np.empty(RB, dtype=abi.DECISION_DTYPE), np.empty(RB * max(N - 1, 1), dtype=abi.RPC_DTYPE))
                best_r, sums = 0.0, []
                for rep in range(3):
                    eng_r.set_state(0, st_aged)
                    nd = 0
                    t0 = time.perf_counter()
                    for b in big:
                        eng_r.submit(b); eng_r.collect(out=bufs_r); nd += len(b)
                    best_r = max(best_r, nd / (time.perf_counter() - t0))
                    sums.append(eng_r.state_checksum())
                rounds4[label] = {"value": best_r, "trains": eng_r.submit_trains(), "state_checksum": f"{sums[-1]:#018x}"}
                eng_r.close()
            assert rounds4["fused_train"]["state_checksum"] == rounds4["launch_per_round"]["state_checksum"]
            host_path["rounds4"] = {"unit": "decisions/s", "batch_messages": [int(len(b)) for b in big], **rounds4,
                                    "note": "four ticks per rgb_submit = four sub-tick rounds per batch, submit + collect "
                                            "back to back (no pipelining), best of 3; both forms end in the same state"}
        except Exception as e:                                              # noqa: BLE001 - reported, not raised
            host_path["rounds4"] = {"error": f"{type(e).__name__}: {e}"}
        # the same shape at the size of ONE scheduler's mailbox drain: the four ticks' messages of the first 1 024
        # groups (~13 k messages, four rounds) -- what a round trip costs when the launches, not the copies, are the
        # time: this is where fusing the rounds into one train launch is meant to pay
        try:
            small = np.concatenate([m[m["server"] < 1024 * N] for m in first_ticks[:4]])
            lat = {}
            for label, flags in (("fused_train", getattr(abi, "CFG_SUBMIT_TRAINS", 0)), ("launch_per_round", abi.CFG_ROUNDS_PER_LAUNCH)):
                eng_s = engine.RaGpuBatch(G, N, device=local_rank, max_runs=16, ring_slots=2, ring_capacity=1 << 16, flags=flags)
                bufs_s = (np.empty(1 << 16, dtype=abi.DECISION_DTYPE), np.empty((1 << 16) * max(N - 1, 1), dtype=abi.RPC_DTYPE))
                eng_s.set_state(0, st_aged)
                for _ in range(20):
                    eng_s.submit(small); eng_s.collect(out=bufs_s)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter()
                    eng_s.submit(small); eng_s.collect(out=bufs_s)
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                lat[label] = {"round_trip_us_p50": round(ts[len(ts) // 2] * 1e6, 1), "round_trip_us_p10": round(ts[len(ts) // 10] * 1e6, 1),
                              "trains": eng_s.submit_trains()}
                # where the round trip goes: rgb_submit (validation, rounds, bucket sort into the pinned slot, enqueue of the
                # copy in + one launch per round + count / un-permute kernels + copies out), the device's part behind it
                # (until the stream is idle), rgb_collect (event wait, copy-out of decisions and rpc records)
                bd = [[], [], []]
                for _ in range(100):
                    t0 = time.perf_counter(); eng_s.submit(small)
                    t1 = time.perf_counter(); eng_s.synchronize()
                    t2 = time.perf_counter(); eng_s.collect(out=bufs_s)
                    t3 = time.perf_counter()
                    bd[0].append(t1 - t0); bd[1].append(t2 - t1); bd[2].append(t3 - t2)
                lat[label]["breakdown_us_p50"] = {k: round(sorted(v)[len(v) // 2] * 1e6, 1)
                                                  for k, v in zip(("rgb_submit", "device_behind_submit", "rgb_collect"), bd)}
                tv = []
                for _ in range(200):
                    t0 = time.perf_counter()
                    eng_s.submit(small); eng_s.release(eng_s.collect_view()[3])
                    tv.append(time.perf_counter() - t0)
                tv.sort()
                lat[label]["collect_view_round_trip_us_p50"] = round(tv[len(tv) // 2] * 1e6, 1)
                if label == "fused_train" and hasattr(eng_s, "inject_train_fault"):
                    # what a FAILED train launch costs (asked for in rounds 4 and 5): the next batch's launch is made to
                    # fail (two messages bucketed under each other's shard: the order check of every wavefront);
                    # rgb_submit / rgb_collect repair it -- undo log back, the batch again with one launch per round:
                    # the failed batch's own round trip, then the round trips after it
                    form0, rec0 = eng_s.train_form(), eng_s.train_recoveries()
                    eng_s.inject_train_fault(2)
                    t0 = time.perf_counter()
                    eng_s.submit(small); eng_s.collect(out=bufs_s)
                    t_fault = time.perf_counter() - t0
                    ts2 = []
                    for _ in range(100):
                        t0 = time.perf_counter()
                        eng_s.submit(small); eng_s.collect(out=bufs_s)
                        ts2.append(time.perf_counter() - t0)
                    ts2.sort()
                    lat["after_injected_train_failure"] = {
                        "failed_batch_round_trip_us": round(t_fault * 1e6, 1), "recoveries": eng_s.train_recoveries() - rec0,
                        "form_before": form0, "form_after": eng_s.train_form(),
                        "round_trip_us_p50_after": round(ts2[len(ts2) // 2] * 1e6, 1),
                        "note": "repair = the undo log restored + the batch re-run with one launch per round (a mis-bucketed "
                                "pair of messages: RGB_TRAIN_ERR_ORDER; a PLACEMENT failure would also move the context to "
                                "the persistent form)"}
                eng_s.close()
            host_path["rounds4_small"] = {"batch_messages": int(len(small)), **lat,
                                          "note": "submit + collect of one small four-round batch, 200 round trips after 20 of "
                                                  "warm-up (the state moves on; same batch every time)"}
        except Exception as e:                                              # noqa: BLE001 - reported, not raised
            host_path["rounds4_small"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- second kernel of the path's neighbourhood, rank 0, N=1: batched WAL entry
    # checksums, 262 144 entries x 4 KiB = 1 GiB resident in HBM (four times the Infinity Cache); reported
    # beside the headline, never as `value` ----
    wal = wal_frame = wal_frame_256 = None
    if rank == 0 and world == 1 and not args.no_host_path:
        import zlib
        n_e, ln = 262144, 4096
        ent = np.zeros(n_e, dtype=abi.WAL_ENTRY_DTYPE)
        ent["index"] = np.arange(n_e); ent["term"] = 3
        ent["data_offset"] = np.arange(n_e, dtype=np.uint64) * ln; ent["data_len"] = ln
        d_pay = torch.randint(0, 256, (n_e * ln + 16,), dtype=torch.uint8, device=dev)
        d_ent = torch.from_numpy(ent.view(np.uint8)).to(dev)
        d_sum = torch.zeros(n_e, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            eng.wal_adler32_device(d_ent.data_ptr(), n_e, d_pay.data_ptr(), n_e * ln + 16, d_sum.data_ptr(), sptr)
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record(stream)
        for _ in range(20):
            eng.wal_adler32_device(d_ent.data_ptr(), n_e, d_pay.data_ptr(), n_e * ln + 16, d_sum.data_ptr(), sptr)
        w1.record(stream)
        torch.cuda.synchronize()
        w_us = w0.elapsed_time(w1) * 1e3 / 20
        host = d_pay[:64 * ln].cpu().numpy()
        want = [zlib.adler32(int(i).to_bytes(8, "big") + (3).to_bytes(8, "big") + host[i * ln:(i + 1) * ln].tobytes())
                for i in range(64)]
        assert d_sum[:64].cpu().numpy().view(np.uint32).tolist() == want, "WAL checksum mismatch vs zlib"
        w_bytes = n_e * (ln + 36)
        wal = {"kernel": "rgb_wal_adler32_kernel<64>", "entries": n_e, "payload_bytes_each": ln,
               "us_per_launch": w_us, "achieved": w_bytes / (w_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
               "unit": "GB/s", "frac": w_bytes / (w_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
               "note": "erlang:adler32([<<Idx:64,Term:64>> | Data]) per WAL entry (src/ra_log_wal.erl:528-534); "
                       "algorithmic bytes = payload + 32-B entry record + 4-B checksum; first 64 checked against zlib"}
        # the framing kernel over the same batch (checksum + 27-byte prefix + payload copy in one pass,
        # src/ra_log_wal.erl:513-537); never allowed to disturb the headline: any failure reports itself
        try:
            import struct
            hdr = ((1 << 22) | 9).to_bytes(3, "big")                       # <<Trunc:1, 1:1, IdRef:22>> of a known writer
            d_pay[n_e * ln:n_e * ln + 3] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
            recs = np.zeros(n_e, dtype=abi.WAL_RECORD_DTYPE)
            recs["index"] = ent["index"]; recs["term"] = 3
            recs["data_offset"] = ent["data_offset"]; recs["data_len"] = ln
            recs["hdr_offset"] = n_e * ln; recs["hdr_len"] = 3
            out_bytes = engine.wal_layout(recs, 5)
            d_rec = torch.from_numpy(recs.view(np.uint8)).to(dev)
            d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):
                eng.wal_frame_device(d_rec.data_ptr(), n_e, d_pay.data_ptr(), n_e * ln + 16, d_out.data_ptr(), out_bytes, 0, 0, sptr)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(20):
                eng.wal_frame_device(d_rec.data_ptr(), n_e, d_pay.data_ptr(), n_e * ln + 16, d_out.data_ptr(), out_bytes, 0, 0, sptr)
            f1.record(stream)
            torch.cuda.synchronize()
            f_us = f0.elapsed_time(f1) * 1e3 / 20
            got = d_out[5:5 + 8 * (27 + ln)].cpu().numpy().tobytes()
            exp = b"".join(hdr + struct.pack(">II", want[i], ln) + struct.pack(">QQ", i, 3) + host[i * ln:(i + 1) * ln].tobytes()
                           for i in range(8))
            assert got == exp, "framed records differ from struct.pack + zlib"
            f_bytes = n_e * (2 * ln + 48 + 3 + 27)
            wal_frame = {"kernel": "rgb_wal_frame_kernel<64>", "records": n_e, "payload_bytes_each": ln,
                         "us_per_launch": f_us, "achieved": f_bytes / (f_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": f_bytes / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                         "note": "Record = [HeaderData, <<Checksum:32, Len:32>>, <<Idx:64, Term:64>> | Data]; algorithmic "
                                 "bytes = 2 x payload + 48-B descriptor + 3 + 27 per record; first 8 checked"}
            del d_rec, d_out
        except Exception as e:                                              # noqa: BLE001 - reported, not raised
            wal_frame = {"error": f"{type(e).__name__}: {e}"}
        # the weak case of the framing kernel, in the driver-run record as well: 256-byte payloads (2 M records,
        # 512 MiB of payload; eight lanes per record) -- the per-record work that does not shrink with the payload
        try:
            import struct
            n_s, ln_s = 1 << 21, 256
            hdr = ((1 << 22) | 9).to_bytes(3, "big")
            d_pay_s = torch.randint(0, 256, (n_s * ln_s + 16,), dtype=torch.uint8, device=dev)
            d_pay_s[n_s * ln_s:n_s * ln_s + 3] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
            recs = np.zeros(n_s, dtype=abi.WAL_RECORD_DTYPE)
            recs["index"] = np.arange(n_s); recs["term"] = 3
            recs["data_offset"] = np.arange(n_s, dtype=np.uint64) * ln_s; recs["data_len"] = ln_s
            recs["hdr_offset"] = n_s * ln_s; recs["hdr_len"] = 3
            out_bytes = engine.wal_layout(recs, 5)
            d_rec = torch.from_numpy(recs.view(np.uint8)).to(dev)
            d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):
                eng.wal_frame_device(d_rec.data_ptr(), n_s, d_pay_s.data_ptr(), n_s * ln_s + 16, d_out.data_ptr(), out_bytes, 0, 0, sptr)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(20):
                eng.wal_frame_device(d_rec.data_ptr(), n_s, d_pay_s.data_ptr(), n_s * ln_s + 16, d_out.data_ptr(), out_bytes, 0, 0, sptr)
            f1.record(stream)
            torch.cuda.synchronize()
            f_us = f0.elapsed_time(f1) * 1e3 / 20
            host_s = d_pay_s[:8 * ln_s].cpu().numpy()
            got = d_out[5:5 + 8 * (27 + ln_s)].cpu().numpy().tobytes()
            exp = b"".join(hdr + struct.pack(">II", zlib.adler32(struct.pack(">QQ", i, 3) + host_s[i * ln_s:(i + 1) * ln_s].tobytes()), ln_s) +
                           struct.pack(">QQ", i, 3) + host_s[i * ln_s:(i + 1) * ln_s].tobytes() for i in range(8))
            assert got == exp, "framed 256-byte records differ from struct.pack + zlib"
            f_bytes = n_s * (2 * ln_s + 48 + 3 + 27)
            wal_frame_256 = {"kernel": "rgb_wal_frame_kernel<8>", "records": n_s, "payload_bytes_each": ln_s,
                             "us_per_launch": f_us, "achieved": f_bytes / (f_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": f_bytes / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                             "note": "the framing kernel's weak case: 256-byte payloads; first 8 records checked"}
            del d_rec, d_out, d_pay_s
        except Exception as e:                                              # noqa: BLE001 - reported, not raised
            wal_frame_256 = {"error": f"{type(e).__name__}: {e}"}
        del d_pay, d_ent, d_sum

    # ---- the literal SURVEY 8(d) configurations (rank 0, N=1 only): reported beside the headline ----
    literal = None
    if rank == 0 and world == 1 and args.literal_ticks > 0:
        literal = {}
        for name in ("2", "3", "5"):
            try:
                literal["config" + name] = run_literal(name, args.literal_ticks, torch, engine, W, abi, dev, local_rank)
            except SystemExit:
                raise
            except Exception as e:                                          # noqa: BLE001 - reported, not raised
                literal["config" + name] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        per_launch_s = (ev_ms / 1e3) / K
        launch_bytes = float(alg_bytes[Wm:].mean())
        achieved = launch_bytes / per_launch_s / 1e9
        names = ["nop", "aer", "aer_reply", "request_vote", "vote_result", "written", "pipeline_rpcs",
                 "append", "await_timeout", "election_timeout", "pre_vote_rpc", "pre_vote_result",
                 "snapshot_written", "heartbeat_rpc", "heartbeat_reply", "consistent_query"]
        tk = kc[Wm:].sum(axis=0)
        mix = {names[i]: round(float(tk[i]) / float(tk[1:].sum()), 4) for i in range(1, NK) if tk[i]}
        traffic, traffic_src = None, None
        try:   # PMC passes cannot run inside this process: the newest committed per-launch figure, if any
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            same_call = os.environ.get("RGB_TRAFFIC_JSON")         # written by the PMC passes of the same gpurun call
            if same_call and os.path.exists(same_call):
                cands = [same_call]
            if cands:
                traffic_src = os.path.relpath(cands[-1], ROOT)
                with open(cands[-1]) as f:
                    tj = json.load(f)
                    traffic = float(tj["traffic_bytes_per_launch"])
                    if use_train and "traffic_bytes_per_tick" in tj:
                        traffic = float(tj["traffic_bytes_per_tick"]) * TPL
                        # the driver's form (one 20-tick launch): the PMC passes of 20-tick launches, when the file has them
                        short = tj.get("closed_loop_20_tick_launches_the_drivers_form")
                        if short and TPL <= 2 * int(short.get("ticks_per_launch", 0)):
                            traffic = float(short["traffic_bytes_per_tick"]) * TPL
                            traffic_src += " (closed_loop_20_tick_launches_the_drivers_form)"
                    elif not use_train and "per_tick_kernel_same_run" in tj:
                        traffic = float(tj["per_tick_kernel_same_run"]["traffic_bytes_per_launch"])
        except Exception:
            traffic = None
        out = {
            "metric": "append_entries decisions/sec across N Raft groups; achieved HBM GB/s vs peak",
            "value": total_dec / elapsed,
            "unit": "decisions/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed * 1e3 / K,
            "wall_minus_events_us": round((wall - ev_ms / 1e3) * 1e6, 1), "wall_breakdown": wall_breakdown,
            "higher_is_better": True, "scaling": "strong" if args.config4 else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": (f"configs[3] as written: {args.total_groups} groups x {N} members in TOTAL, hashed "
                             f"(splitmix64(group) mod {world} = rgb_route) over {world} GPU(s), rank 0 holds {G}; the "
                             "configs[2] closed-loop mix per GPU; leaderboard all-gather every 16 ticks through the C "
                             "entry point; strong scaling (the population is fixed)") if args.config4 else
                            "configs[2] closed loop: 65536 groups x 5 members per GPU, mixed append_entries + "
                            "request_vote (5% of the groups per tick see a request_vote with term+1 and re-elect), "
                            "every server may get a message every tick (device-side generator), device-resident "
                            "message batches; the ticks of one leaderboard period run as ONE train launch (per-server "
                            "sequence bytes order the ticks, --launch tick = one kernel launch per tick); the literal "
                            "one-message-per-group form of SURVEY 8(d) is under literal_configs.config3",
                "aged_ticks": A, "n_runs_histogram_at_start": n_runs_hist,
                "groups_per_gpu": G, "members": N, "decisions_per_tick": float(n_dec[Wm:].mean()),
                "message_mix": mix,
                "leaderboard_allgather_every": SNAPSHOT_EVERY,
                "parallelism": f"hash-sharded groups x{world}, no data-path collective",
                "oracle_checked_ticks": checked, "state_checksum": f"{checksum_pass2:#018x}",
                "stream_generation_s": round(gen_s, 2), "hip_graph": graphs is not None,
                "launch": (("train: one rgb_train_kernel launch per %d ticks, leaderboard snapshots every %d ticks as rows "
                            "of the launch" % (TPL, SNAPSHOT_EVERY)) if snap_in_train else
                           "train: one rgb_train_kernel launch per leaderboard period" if use_train
                           else "one rgb_tick_classes_kernel launch per tick"),
                "train": train_info,
                "per_rank": per_rank,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": f"HBM bytes per launch from {traffic_src} (rocprofv3 PMC passes, FETCH_SIZE with the "
                                "guide's gfx950 x2 correction + WRITE_SIZE)" +
                                ("; PMC passes of the same gpurun call, in front of this run" if os.environ.get("RGB_TRAFFIC_JSON")
                                 else "; not re-measured in this run"),
                "kernel": ((f"rgb_train_dealt_kernel<{N}>" if train_info and train_info.get("form") == "dealt"
                            else f"rgb_train_kernel<{N}>") if use_train else f"rgb_tick_classes_kernel<{N}>")
                          if not args.generic_kernel else f"rgb_tick_kernel<{N},generic>",
                "ticks_per_launch": TPL,
                "algorithmic_bytes_per_launch": launch_bytes * TPL,
                "avg_launch_us": per_launch_s * 1e6 * TPL,
                "algorithmic_bytes_per_tick": launch_bytes, "avg_tick_us": per_launch_s * 1e6,
            },
            "cpu_baseline": cpu_baseline,
            "literal_configs": literal,
            "host_path": host_path,
            "aux_kernels": {"wal_adler32": wal, "wal_frame": wal_frame, "wal_frame_256": wal_frame_256},
        }
        try:                                   # RCCL prints its version banner through C stdio: out before the line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    _ = my_groups


if __name__ == "__main__":
    main()
