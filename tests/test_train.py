"""Train launches (rgb_train_*, include/ra_gpu_batch.h): several device-resident ticks in ONE launch, ordered per
server by sequence stamps instead of kernel boundaries.  The claim under test: a train computes exactly what
rgb_run_ticks_device / the per-tick class kernel compute on the same ticks -- every decision, every rpc record, the
final state -- and what the oracle computes.

The same test bodies run on the GPU (`-m gpu`, real library, torch device buffers) and on the CPU block emulation
(tests/native: blocks run one after another in grid order, so every dependency is met on the first poll -- it checks
the plan, the bucket order, the stamps and the commit protocol's arithmetic, not the races).  The GPU cases include
trains of more than 16 small ticks, all resident at once: dozens of ticks genuinely in flight together."""
import ctypes as C

import numpy as np
import pytest

from ra_amd import abi
from ra_amd import workload as W


class Buf:
    """`nbytes` of memory the library can use as DEVICE memory: a CUDA tensor on the GPU, numpy on the emulation."""

    def __init__(self, nbytes, on_gpu):
        self.on_gpu = on_gpu
        if on_gpu:
            import torch
            self.t = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device="cuda")
            self.ptr = self.t.data_ptr()
        else:
            self.a = np.zeros(max(nbytes, 16), dtype=np.uint8)
            self.ptr = self.a.ctypes.data

    def host(self):
        return self.t.cpu().numpy() if self.on_gpu else self.a

    def sync(self):
        if self.on_gpu:
            import torch
            torch.cuda.synchronize()


def _generate(engine, G, N, T, seed, on_gpu, max_runs=16, age=0, flags=0):
    """T generator ticks applied one by one with the per-tick class kernel.  Returns everything a train needs."""
    S = G * N
    eng = engine.RaGpuBatch(G, N, max_runs=max_runs, ring_slots=1, ring_capacity=64, flags=flags)
    st0 = W.initial_states(G, N, seed)
    eng.set_state(0, st0)
    tb = S * 64
    msgs, dec = Buf(T * tb, on_gpu), Buf(T * tb, on_gpu)
    rpcs = Buf(T * S * max(N - 1, 1) * 56, on_gpu)
    kc, dn = Buf(T * abi.N_KINDS * 4, on_gpu), Buf(T * 4, on_gpu)
    bc = Buf(T * engine.TRAIN_BUCKETS * 4, on_gpu)
    scratch_m, scratch_d = Buf(tb, on_gpu), Buf(tb, on_gpu)
    for t in range(age):                                    # untimed ageing: ticks applied and not kept
        eng.synth_tick_buckets_device(seed, t, scratch_m.ptr, 0, 0, 0)
        eng.synth_apply_tick_device(scratch_m.ptr, S, scratch_d.ptr, rpcs.ptr)
    eng.synchronize()
    st_start = eng.get_state()
    rs = S * max(N - 1, 1) * 56
    for t in range(T):
        eng.synth_tick_buckets_device(seed, age + t, msgs.ptr + t * tb, kc.ptr + t * abi.N_KINDS * 4, dn.ptr + t * 4,
                                      bc.ptr + t * engine.TRAIN_BUCKETS * 4)
        eng.synth_apply_tick_device(msgs.ptr + t * tb, S, dec.ptr + t * tb, rpcs.ptr + t * rs)
    eng.synchronize()
    counts = dn.host().view(np.uint32)[:T].copy()
    buckets = bc.host().view(np.uint32)[:T * engine.TRAIN_BUCKETS].reshape(T, engine.TRAIN_BUCKETS).copy()
    assert np.array_equal(buckets.sum(axis=1), counts)
    kinds = kc.host().view(np.uint32)[:T * abi.N_KINDS].reshape(T, abi.N_KINDS).copy()
    return dict(eng=eng, S=S, tb=tb, rs=rs, msgs=msgs, dec=dec, rpcs=rpcs, counts=counts, buckets=buckets, kinds=kinds,
                st_start=st_start, st_end=eng.get_state(), sum_end=eng.state_checksum())


def _tick(buf, t, tb, n, dtype):
    a = buf.host()[t * tb:t * tb + n * 64].view(dtype).copy()
    return abi.expand_decisions(a) if dtype is abi.DECISION_DTYPE else a      # device streams hold compact records


def check_train_equals_per_tick_launches(engine, oracle_lib, G, N, T, seed, on_gpu, chunks=(None,), age=0, flags=0):
    r = _generate(engine, G, N, T, seed, on_gpu, age=age, flags=flags)
    eng, S, tb, rs = r["eng"], r["S"], r["tb"], r["rs"]
    want_dec = [_tick(r["dec"], t, tb, int(r["counts"][t]), abi.DECISION_DTYPE) for t in range(T)]
    want_rpc = r["rpcs"].host()[:T * rs].copy()
    plain = [_tick(r["msgs"], t, tb, int(r["counts"][t]), abi.MSG_DTYPE) for t in range(T)]
    # rgb_run_ticks_device over the same stream: the class-dispatch kernel (kind counts) and the kind-generic kernel
    for kinds in (r["kinds"], None):
        eng.set_state(0, r["st_start"])
        dec1 = Buf(T * tb, on_gpu)
        eng.run_ticks_device(r["msgs"].ptr, S, T, dec1.ptr, 0, tick_counts=r["counts"], kind_counts=kinds)
        eng.synchronize()
        for t in range(T):
            got = _tick(dec1, t, tb, int(r["counts"][t]), abi.DECISION_DTYPE)
            assert got.tobytes() == want_dec[t].tobytes(), f"rgb_run_ticks_device (kind counts: {kinds is not None}) tick {t}"
        assert eng.get_state().tobytes() == r["st_end"].tobytes()
    for t in range(T):      # the ticks are in bucket order, which keeps every class (and every (class, shard)) contiguous
        bk = engine.train_bucket(plain[t]["kind"], plain[t]["flags"], plain[t]["server"], N) >> 1
        assert np.all(np.diff(bk.astype(np.int64)) >= 0)
        assert np.array_equal(np.bincount(bk, minlength=engine.TRAIN_BUCKETS // 2), r["buckets"][t].reshape(-1, 2).sum(axis=1))
    # the oracle on the same stream
    if oracle_lib is not None:
        cpu = oracle_lib.Oracle(G, N, max_runs=16)
        cpu.set_state(0, r["st_start"])
        for t in range(T):
            want, _ = cpu.step_parallel(plain[t]) if hasattr(cpu, "step_parallel") else cpu.step(plain[t])
            assert want.tobytes() == want_dec[t].tobytes(), f"per-tick kernel differs from the oracle at tick {t}"
        assert cpu.get_state().tobytes() == r["st_end"].tobytes()
        cpu.close()
    plan = eng.train_plan(r["buckets"])
    assert plan.blocks_per_tick % 8 == 0 and plan.blocks_per_tick > 0
    stamps = Buf(T * S, on_gpu)
    seen = np.zeros(S, dtype=np.int64)                      # messages every server has received through trains
    for chunk in chunks:
        # back to the start state; the train runs in `chunk`-tick launches (None = one launch for all T ticks)
        eng.set_state(0, r["st_start"])
        dec2, rpc2 = Buf(T * tb, on_gpu), Buf(T * rs, on_gpu)
        step = chunk or T
        t = 0
        while t < T:
            n = min(step, T - t)
            # the stamps of a launch's ticks count on from what the servers hold now
            eng.train_stamp_device(r["msgs"].ptr + t * tb, stamps.ptr + t * S, S, r["counts"][t:t + n])
            eng.train_run_device(plan, t, n, r["msgs"].ptr, stamps.ptr, S, dec2.ptr, rpc2.ptr + t * rs, rpc_ring=n)
            t += n
        eng.synchronize()
        flags, xcc = eng.train_status()
        assert flags == 0
        # the messages are untouched; the stamp of a message = messages its server received through trains before it
        # (the sequence bytes are never reset: not by rgb_upload_state either)
        st_h = stamps.host()
        for t in range(T):
            got = _tick(dec2, t, tb, int(r["counts"][t]), abi.DECISION_DTYPE)
            if got.tobytes() != want_dec[t].tobytes():
                bad = int(np.flatnonzero((got.view(np.uint8).reshape(-1, 64) !=
                                          want_dec[t].view(np.uint8).reshape(-1, 64)).any(axis=1))[0])
                raise AssertionError(f"chunk {chunk}: tick {t} slot {bad}: msg={plain[t][bad]}\n train={got[bad]}\n "
                                     f"per-tick={want_dec[t][bad]}")
            assert _tick(r["msgs"], t, tb, int(r["counts"][t]), abi.MSG_DTYPE).tobytes() == plain[t].tobytes()
            n_t = int(r["counts"][t])
            assert np.array_equal(st_h[t * S:t * S + n_t], (seen[plain[t]["server"]] & 255).astype(np.uint8)), f"stamps of tick {t}"
            seen[plain[t]["server"]] += 1
        # rpc records: message i of tick t owns slots [i (N-1), (i+1)(N-1)), the first n_rpcs are valid
        per = max(N - 1, 1)
        for t in range(T):
            n_r = want_dec[t]["n_rpcs"].astype(np.int64)
            a = want_rpc[t * rs:(t + 1) * rs].view(abi.RPC_DTYPE)
            b = rpc2.host()[t * rs:(t + 1) * rs].view(abi.RPC_DTYPE)
            for i in np.flatnonzero(n_r):
                k = int(n_r[i])
                x, y = a[i * per:i * per + k].copy(), b[i * per:i * per + k].copy()
                # msg_index is the global index: the per-tick apply numbered from 0, the train from t * stride
                assert np.array_equal(y["msg_index"].astype(np.int64), x["msg_index"].astype(np.int64) + t * S)
                x["msg_index"] = 0; y["msg_index"] = 0
                assert x.tobytes() == y.tobytes(), f"rpc records of tick {t} message {i}"
        assert eng.get_state().tobytes() == r["st_end"].tobytes(), f"chunk {chunk}: final state differs"
        assert eng.state_checksum() == r["sum_end"]
    # the plan built ON THE DEVICE from the bucket counts in device memory (rgb_train_plan_build_device): the same
    # tables as the host's, bit for bit, tick by tick; and a train that runs from it (persistent form) computes the same
    dbc = Buf(T * engine.TRAIN_BUCKETS * 4, on_gpu)
    if on_gpu:
        import torch
        dbc.t.copy_(torch.from_numpy(r["buckets"].reshape(-1).view(np.uint8).copy()))
    else:
        dbc.a[:] = r["buckets"].reshape(-1).view(np.uint8)
    dplan = engine.TrainPlan(eng, None, device_ticks=T)
    half = T // 2                                           # built in two calls: ticks [0, half) and [half, T)
    dplan.build_device(0, half, dbc.ptr)
    dplan.build_device(half, T - half, dbc.ptr + half * engine.TRAIN_BUCKETS * 4)
    for t in range(T):
        hh, ho, hc, hr = plan.download(t)
        dh, do_, dc, dr = dplan.download(t)
        assert np.array_equal(hh, dh), f"tick {t}: header {hh} vs {dh}"
        assert np.array_equal(ho, do_) and np.array_equal(hc, dc), f"tick {t}: offsets / counts"
        assert np.array_equal(hr, dr), f"tick {t}: row table differs at {np.flatnonzero(hr != dr)[:5]}"
    eng.set_state(0, r["st_start"])
    dec3, rpc3 = Buf(T * tb, on_gpu), Buf(T * rs, on_gpu)
    eng.train_stamp_device(r["msgs"].ptr, stamps.ptr, S, r["counts"])
    eng.train_run_device(dplan, 0, T, r["msgs"].ptr, stamps.ptr, S, dec3.ptr, rpc3.ptr, rpc_ring=T)
    eng.synchronize()
    assert eng.train_status()[0] == 0
    for t in range(T):
        got = _tick(dec3, t, tb, int(r["counts"][t]), abi.DECISION_DTYPE)
        assert got.tobytes() == want_dec[t].tobytes(), f"device-built plan: decisions of tick {t}"
    assert eng.get_state().tobytes() == r["st_end"].tobytes(), "device-built plan: final state differs"
    # the same plan once the host has been told its ticks' rows (rgb_train_plan_fit): the grid is the rows, not the bound
    dplan.fit(0, T)
    eng.set_state(0, r["st_start"])
    dec4, rpc4 = Buf(T * tb, on_gpu), Buf(T * rs, on_gpu)
    eng.train_stamp_device(r["msgs"].ptr, stamps.ptr, S, r["counts"])
    eng.train_run_device(dplan, 0, T, r["msgs"].ptr, stamps.ptr, S, dec4.ptr, rpc4.ptr, rpc_ring=T)
    eng.synchronize()
    assert eng.train_status()[0] == 0
    for t in range(T):
        got = _tick(dec4, t, tb, int(r["counts"][t]), abi.DECISION_DTYPE)
        assert got.tobytes() == want_dec[t].tobytes(), f"device-built plan, fitted grid: decisions of tick {t}"
    assert eng.get_state().tobytes() == r["st_end"].tobytes(), "device-built plan, fitted grid: final state differs"
    dplan.close()
    plan.close()
    eng.close()


@pytest.mark.parametrize("G,N,T,seed", [(192, 5, 20, 0x5EED0003), (96, 3, 12, 7), (64, 7, 10, 11), (72, 6, 8, 13), (80, 8, 8, 17)])
def test_train_on_the_block_emulation(emulated_engine, oracle_lib, G, N, T, seed):
    check_train_equals_per_tick_launches(emulated_engine, oracle_lib, G, N, T, seed, False, chunks=(None, 3))


@pytest.mark.parametrize("G,N,T,seed", [(160, 5, 12, 21), (64, 7, 8, 23)])
def test_fused_pipelining_inside_a_train_on_the_block_emulation(emulated_engine, G, N, T, seed):
    """RGB_CFG_FUSE_PIPELINE inside train launches: a fused event stores its rpc records itself while the message's own
    records wait in LDS for the publish (emit_rpc) -- the train's decisions, rpc records and state must be those of the
    per-tick launches of the same (fused) engine.  (The checker does not fuse: no oracle leg here; the fused semantics
    themselves are tests/test_fused_pipeline.py.)"""
    check_train_equals_per_tick_launches(emulated_engine, None, G, N, T, seed, False, chunks=(None, 3), flags=abi.CFG_FUSE_PIPELINE)


def test_train_with_a_wrong_stamp_fails_in_bounded_time(emulated_engine):
    """A message whose stamp never comes up: the wavefront gives up (RGB_TRAIN_ERR_SPIN), the launch ends, the host
    sees the error -- never a hang."""
    engine = emulated_engine
    r = _generate(engine, 64, 5, 3, 5, False)
    eng, S = r["eng"], r["S"]
    eng.set_state(0, r["st_start"])
    plan = eng.train_plan(r["buckets"])
    stamps = Buf(3 * S, False)
    eng.train_stamp_device(r["msgs"].ptr, stamps.ptr, S, r["counts"])
    stamps.host()[0] = 5                                    # the first message of tick 0 must find 0
    dec2 = Buf(3 * r["tb"], False)
    eng.train_run_device(plan, 0, 3, r["msgs"].ptr, stamps.ptr, S, dec2.ptr)
    eng.synchronize()
    flags, _ = eng.train_status(check=False)
    assert flags & 2
    assert eng.train_status(check=False)[0] == 0          # reading the status clears it
    eng.close()


def check_generator_stamps(engine, G, N, T, seed, on_gpu):
    """The stamps the load generator writes with its ticks = what rgb_train_stamp_device counts over the same stream,
    and a replay from the same sequence bytes accepts them."""
    S, tb = G * N, G * N * 64
    eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
    eng.set_state(0, W.initial_states(G, N, seed))
    st0 = eng.get_state()
    msgs, dec, dec2 = Buf(T * tb, on_gpu), Buf(T * tb, on_gpu), Buf(T * tb, on_gpu)
    rpcs = Buf(4 * S * max(N - 1, 1) * 56, on_gpu)
    dn, bc = Buf(T * 4, on_gpu), Buf(T * engine.TRAIN_BUCKETS * 4, on_gpu)
    stamps, stamps2 = Buf(T * S, on_gpu), Buf(T * S, on_gpu)
    for t in range(T):
        eng.synth_tick_stamped_device(seed, t, msgs.ptr + t * tb, 0, dn.ptr + t * 4, bc.ptr + t * engine.TRAIN_BUCKETS * 4,
                                      stamps.ptr + t * S)
        eng.synth_apply_tick_device(msgs.ptr + t * tb, S, dec.ptr + t * tb, rpcs.ptr)
    eng.synchronize()
    counts = dn.host().view(np.uint32)[:T].copy()
    buckets = bc.host().view(np.uint32)[:T * engine.TRAIN_BUCKETS].reshape(T, engine.TRAIN_BUCKETS).copy()
    sum_end = eng.state_checksum()
    plan = eng.train_plan(buckets)
    eng.train_stamp_device(msgs.ptr, stamps2.ptr, S, counts)
    eng.synchronize()
    a, b = stamps.host(), stamps2.host()
    for t in range(T):
        n = int(counts[t])
        assert np.array_equal(a[t * S:t * S + n], b[t * S:t * S + n]), f"generator stamps of tick {t}"
    eng.set_state(0, st0)
    eng.train_run_device(plan, 0, T, msgs.ptr, stamps.ptr, S, dec2.ptr, rpcs.ptr, rpc_ring=4)
    eng.synchronize()
    assert eng.train_status()[0] == 0
    for t in range(T):
        n = int(counts[t])
        assert _tick(dec2, t, tb, n, abi.DECISION_DTYPE).tobytes() == _tick(dec, t, tb, n, abi.DECISION_DTYPE).tobytes()
    assert eng.state_checksum() == sum_end
    plan.close()
    eng.close()


def check_snapshots_inside_a_train(engine, G, N, T, every, seed, on_gpu, windows):
    """Leaderboard snapshots as rows of the launch (rgb_train_plan_create_snap / rgb_train_run_snap_device): the stream
    is generated with a mark at every boundary (ticks k * every), replayed in `windows` (launches; a boundary at a
    window's start is taken outside with rgb_snapshot_train_device) -- every decision, every snapshot row and the final
    state equal the per-tick launches with rgb_snapshot_device between them."""
    S, tb = G * N, G * N * 64
    eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
    eng.set_state(0, W.initial_states(G, N, seed))
    st0 = eng.get_state()
    seqb = eng.train_seq_bytes()
    assert seqb >= S
    n_snap = (T - 1) // every
    msgs, dec, dec2 = Buf(T * tb, on_gpu), Buf(T * tb, on_gpu), Buf(T * tb, on_gpu)
    rpcs = Buf(4 * S * max(N - 1, 1) * 56, on_gpu)
    dn, bc = Buf(T * 4, on_gpu), Buf(T * engine.TRAIN_BUCKETS * 4, on_gpu)
    stamps = Buf(T * S, on_gpu)
    snap_stamps = Buf(max(n_snap, 1) * seqb, on_gpu)
    rows_ref, rows_got = Buf(max(n_snap, 1) * G * 32, on_gpu), Buf(max(n_snap, 1) * G * 32, on_gpu)
    for t in range(T):
        if t and t % every == 0:
            k = t // every - 1
            eng.synth_snapshot_mark_device(snap_stamps.ptr + k * seqb)
            eng.snapshot_device(rows_ref.ptr + k * G * 32)
        eng.synth_tick_stamped_device(seed, t, msgs.ptr + t * tb, 0, dn.ptr + t * 4, bc.ptr + t * engine.TRAIN_BUCKETS * 4,
                                      stamps.ptr + t * S)
        eng.synth_apply_tick_device(msgs.ptr + t * tb, S, dec.ptr + t * tb, rpcs.ptr)
    eng.synchronize()
    counts = dn.host().view(np.uint32)[:T].copy()
    buckets = bc.host().view(np.uint32)[:T * engine.TRAIN_BUCKETS].reshape(T, engine.TRAIN_BUCKETS).copy()
    sum_end = eng.state_checksum()
    plan = eng.train_plan_snap(buckets, every)
    # the same plan built on the device (the snapshot's rows lead every tick k * every): equal tables; the one-launch
    # cases replay from the device-built plan (persistent form)
    dplan = engine.TrainPlan(eng, None, snapshot_every=every, device_ticks=T)
    dplan.build_device(0, T, bc.ptr)
    eng.synchronize()
    for t in range(T):
        hh, ho, hc, hr = plan.download(t)
        dh, do_, dc, dr = dplan.download(t)
        assert np.array_equal(hh, dh) and np.array_equal(ho, do_) and np.array_equal(hc, dc), f"tick {t}: header / offsets"
        assert np.array_equal(hr, dr), f"tick {t}: row table differs at {np.flatnonzero(hr != dr)[:5]}"
    run_plan = dplan if len(windows) == 1 else plan
    eng.set_state(0, st0)
    t = 0
    for n in windows:
        n = min(n, T - t)
        if n <= 0:
            break
        if t and t % every == 0:                             # the boundary in front of a launch: outside it
            eng.snapshot_train_device(rows_got.ptr + (t // every - 1) * G * 32)
        eng.train_run_snap_device(run_plan, t, n, msgs.ptr, stamps.ptr, S, dec2.ptr, rpcs.ptr, 4, snap_stamps.ptr, rows_got.ptr)
        t += n
    assert t == T
    eng.synchronize()
    assert eng.train_status()[0] == 0
    for t in range(T):
        n = int(counts[t])
        assert _tick(dec2, t, tb, n, abi.DECISION_DTYPE).tobytes() == _tick(dec, t, tb, n, abi.DECISION_DTYPE).tobytes(), f"tick {t}"
    assert eng.state_checksum() == sum_end
    a, b = rows_ref.host()[:n_snap * G * 32], rows_got.host()[:n_snap * G * 32]
    for k in range(n_snap):
        assert a[k * G * 32:(k + 1) * G * 32].tobytes() == b[k * G * 32:(k + 1) * G * 32].tobytes(), f"snapshot {k}"
    assert n_snap >= 2 and a.any()
    # a plan with snapshots cannot run without their buffers (the sequence bytes would fall behind the stamps)
    with pytest.raises(engine.RgbError):
        eng.train_run_device(plan, 0, T, msgs.ptr, stamps.ptr, S, dec2.ptr)
    dplan.close()
    plan.close()
    eng.close()


@pytest.mark.parametrize("G,N,T,every,windows", [(192, 5, 14, 4, (14,)), (192, 5, 14, 4, (6, 2, 6)), (70, 3, 10, 3, (10,)),
                                                 (64, 7, 9, 4, (4, 5))])
def test_snapshots_inside_a_train_on_the_block_emulation(emulated_engine, G, N, T, every, windows):
    check_snapshots_inside_a_train(emulated_engine, G, N, T, every, 0x5EED0003, False, windows)


@pytest.mark.gpu
@pytest.mark.parametrize("G,N,T,every,windows", [(4096, 5, 40, 16, (40,)), (4096, 5, 40, 16, (11, 5, 24)), (1000, 3, 20, 4, (20,)),
                                                 (1024, 7, 18, 8, (18,))])
def test_snapshots_inside_a_train_on_the_gpu(G, N, T, every, windows):
    from ra_amd import engine
    check_snapshots_inside_a_train(engine, G, N, T, every, 0x5EED0003, True, windows)


def check_repair_stream_as_a_train(engine, oracle_lib, G, N, T, on_gpu):
    """BASELINE configs[4] (7 members, 1 024-entry backlogs over 3-6 term boundaries, wrong prev_log_term half of the
    time) as ONE train launch: the append_entries_rpc wavefronts of groups of six and more members take the first line
    of their servers' run tables with the hot rows (prev_log_term != term) and walk it in LDS -- every decision and the
    final state against the oracle."""
    seed = 0x5EED0005
    S = G * N
    st0 = W.initial_states(G, N, seed, backlog=1024, boundaries=(3, 6))
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st0)
    msgs, decs = [], []
    for t in range(T):
        m = W.gen_tick(cpu.get_state(), N, t, seed, W.MIX_CONFIG5, backlog_mode=True)
        d, _ = cpu.step(m)
        msgs.append(m); decs.append(d)
    want_final = cpu.get_state()
    cpu.close()
    assert sum(int(((m["kind"] == abi.MSG_AER) & (m["b"] != m["term"])).sum()) for m in msgs) > 0
    stride = max(len(m) for m in msgs)
    eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
    eng.set_state(0, st0)
    d_msgs, d_dec, d_stamps = Buf(T * stride * 64, on_gpu), Buf(T * stride * 64, on_gpu), Buf(T * stride, on_gpu)
    d_rpcs = Buf(4 * stride * max(N - 1, 1) * 56, on_gpu)
    host = np.zeros(T * stride, dtype=abi.MSG_DTYPE)
    h_st = np.zeros(T * stride, dtype=np.uint8)
    bcs = np.zeros((T, engine.TRAIN_BUCKETS), dtype=np.uint32)
    sent = np.zeros(S, dtype=np.uint8)
    perms = []
    for t, m in enumerate(msgs):
        bk = engine.train_bucket(m["kind"], m["flags"], m["server"], N)
        perm = np.argsort(bk, kind="stable")
        perms.append(perm)
        bcs[t] = np.bincount(bk, minlength=engine.TRAIN_BUCKETS)
        host[t * stride:t * stride + len(m)] = m[perm]
        srv = m["server"][perm]
        h_st[t * stride:t * stride + len(m)] = sent[srv]
        sent[srv] += 1
    if on_gpu:
        import torch
        d_msgs.t[:host.nbytes].copy_(torch.from_numpy(host.view(np.uint8)))
        d_stamps.t[:h_st.nbytes].copy_(torch.from_numpy(h_st))
    else:
        d_msgs.a[:host.nbytes] = host.view(np.uint8)
        d_stamps.a[:h_st.nbytes] = h_st
    plan = eng.train_plan(bcs)
    eng.train_run_device(plan, 0, T, d_msgs.ptr, d_stamps.ptr, stride, d_dec.ptr, d_rpcs.ptr, rpc_ring=4)
    eng.synchronize()
    assert eng.train_status()[0] == 0
    got = abi.expand_decisions(d_dec.host()[:T * stride * 64].view(abi.DECISION_DTYPE).copy())
    for t in range(T):
        g = got[t * stride:t * stride + len(msgs[t])]
        assert g.tobytes() == decs[t][perms[t]].tobytes(), f"tick {t}"
    assert eng.get_state().tobytes() == want_final.tobytes()
    plan.close()
    eng.close()


def test_repair_stream_as_a_train_on_the_block_emulation(emulated_engine, oracle_lib):
    check_repair_stream_as_a_train(emulated_engine, oracle_lib, 96, 7, 8, False)


@pytest.mark.gpu
def test_repair_stream_as_a_train_on_the_gpu(oracle_lib):
    from ra_amd import engine
    check_repair_stream_as_a_train(engine, oracle_lib, 4096, 7, 12, True)


def test_generator_stamps_on_the_block_emulation(emulated_engine):
    check_generator_stamps(emulated_engine, 192, 5, 12, 0x5EED0003, False)


@pytest.mark.gpu
def test_generator_stamps_on_the_gpu():
    from ra_amd import engine
    check_generator_stamps(engine, 4096, 5, 24, 0x5EED0003, True)


def check_failed_train_is_repaired(engine, oracle_lib, G, N, seed, faults):
    """A train launch of rgb_submit that fails -- a stamp that never comes up, a message bucketed under another
    shard -- is repaired by the engine: the undo logs of the batches in flight go back, the batches run again with one
    launch per round, rgb_collect hands out exactly what the oracle computes, the batch submitted BEHIND the failed
    one included (reference: a member's messages apply in order, exactly once, src/ra_server_proc.erl:1356-1397)."""
    import fuzz
    from test_gpu_parity import assert_same
    rng = np.random.default_rng(seed)
    st = fuzz.random_states(rng, G, N, max_runs=6, backlog=24)
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st)
    with engine.RaGpuBatch(G, N, ring_capacity=65536, ring_slots=4, max_runs=16, flags=abi.CFG_SUBMIT_TRAINS) as gpu:
        gpu.set_state(0, st)
        for fault in faults:
            batches, want = [], []
            for b in range(3):                              # the faulty batch, then two more behind it in the ring
                parts = [fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.9) for _ in range(3 if b != 1 else 1)]
                msgs = np.concatenate(parts)
                msgs = msgs[msgs["kind"] != abi.MSG_NOP]
                rng.shuffle(msgs)
                batches.append(msgs)
                want.append(cpu.step(msgs))
            before = gpu.train_recoveries()
            gpu.inject_train_fault(fault)
            for b, msgs in enumerate(batches):
                gpu.submit(msgs, tick=b)
            for b in range(3):
                dg, rg, tick = gpu.collect()
                assert tick == b
                do, ro = want[b]
                assert dg.tobytes() == do.tobytes(), f"fault {fault}: decisions of batch {b}"
                assert fuzz.sort_rpcs(rg).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"fault {fault}: rpcs of batch {b}"
            assert gpu.train_recoveries() == before + (1 if fault else 0), \
                f"fault {fault}: recoveries {gpu.train_recoveries()} (before {before}), batches as trains so far " \
                f"{gpu.submit_trains()}, form {gpu.train_form()}"

            assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), f"fault {fault}: state"
        assert gpu.submit_trains() >= len(faults)
    cpu.close()


def test_failed_train_is_repaired_on_the_block_emulation(emulated_engine, oracle_lib):
    check_failed_train_is_repaired(emulated_engine, oracle_lib, 1100, 5, 41, faults=(1, 0, 2))


@pytest.mark.gpu
def test_failed_train_is_repaired_on_the_gpu(oracle_lib):
    from ra_amd import engine
    check_failed_train_is_repaired(engine, oracle_lib, 4096, 5, 43, faults=(1, 0, 2))


def test_train_bucket_matches_the_c_function(emulated_engine):
    engine = emulated_engine
    L = engine.lib()
    rng = np.random.default_rng(3)
    for _ in range(300):
        kind, flags, n = int(rng.integers(0, 16)), int(rng.integers(0, 32)), int(rng.integers(1, 9))
        server = int(rng.integers(0, 1 << 20))
        assert L.rgb_train_bucket(kind, flags, server, n) == int(engine.train_bucket(kind, flags, server, n))


@pytest.mark.gpu
@pytest.mark.parametrize("G,N,T,seed,age", [(2048, 5, 24, 0x5EED0003, 0), (1024, 3, 16, 7, 0), (1024, 7, 12, 11, 0),
                                            (16384, 5, 48, 0x5EED0003, 64), (1000, 6, 12, 13, 8), (1536, 8, 12, 17, 8),
                                            (8192, 7, 24, 19, 32)])
def test_train_on_the_gpu(oracle_lib, G, N, T, seed, age):
    """Real races: thousands of wavefronts of neighbouring ticks in flight together, every decision compared."""
    from ra_amd import engine
    check_train_equals_per_tick_launches(engine, oracle_lib, G, N, T, seed, True, chunks=(None, 16, 5), age=age)


@pytest.mark.gpu
@pytest.mark.parametrize("G,N,T,seed,age", [(4096, 5, 24, 0x5EED0003, 32), (2048, 7, 12, 29, 16)])
def test_fused_pipelining_inside_a_train_on_the_gpu(G, N, T, seed, age):
    """RGB_CFG_FUSE_PIPELINE inside train launches on the device (see the emulation test of the same name)."""
    from ra_amd import engine
    check_train_equals_per_tick_launches(engine, None, G, N, T, seed, True, chunks=(None, 5), age=age, flags=abi.CFG_FUSE_PIPELINE)
