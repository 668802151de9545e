"""Full-size, full-length parity (BASELINE.json sizes) -- VERDICT round 1, row x1:
 (i)   configs[2] (65 536 groups x 5): EVERY tick of a 512-tick device-generated stream -- the bench's stream, so
       the state ages into the regime bench.py times (term-run tables of 2..16 runs, compaction, overflows) -- is
       replayed through the oracle: every decision of every tick compared, the state checksum of checksums
       compared every 32 ticks and the whole state at the end;
 (ii)  configs[4] (7 members, 1 024-entry uncommitted backlogs over 3-6 term boundaries) at 16 384 groups: the
       log-matching repair stream of SURVEY 8(d) config 5, every decision, every rpc record, the whole final state;
 (iii) 7 members at 65 536 groups, 64 ticks of the device-generated stream.
Reference cases these streams keep hitting: test/ra_server_SUITE.erl:934-1001 (follower catch-up / term mismatch)
and :1419-1447 (leader repair on failed replies).  Runs only on the GPU box (-m gpu), through the C ABI."""
import os

import numpy as np
import pytest

import fuzz
from ra_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_mod():
    from ra_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    engine.lib()          # raises if the HIP library is missing: no fallback
    return engine


def checksum_of_checksums(per_server: np.ndarray) -> int:
    """rgb_state_checksum's host side: sum_k c_k * (2k + 1) mod 2^64 (numpy uint64 arithmetic wraps)."""
    k = np.arange(len(per_server), dtype=np.uint64)
    with np.errstate(over="ignore"):
        return int((per_server.astype(np.uint64) * (k * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def replay_device_stream(engine_mod, oracle_lib, G, N, seed, ticks, checksum_every):
    import torch
    from ra_amd import workload as W
    S = G * N
    st0 = W.initial_states(G, N, seed)
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st0)
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream
    flags_seen, n_dec, kinds = 0, 0, np.zeros(abi.N_KINDS, dtype=np.int64)
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64) as gpu:
        gpu.set_state(0, st0)
        dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dr = torch.zeros(S * max(N - 1, 1) * 56, dtype=torch.uint8, device="cuda")
        dn = torch.zeros(1, dtype=torch.int32, device="cuda")
        for t in range(ticks):
            gpu.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
            gpu.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
            stream.synchronize()
            n = int(dn.item())
            assert 0 < n <= S
            msgs = dm[:n * 64].cpu().numpy().view(abi.MSG_DTYPE)
            got = abi.expand_decisions(dd[:n * 64].cpu().numpy().view(abi.DECISION_DTYPE))
            srv = msgs["server"]
            assert len(np.unique(srv)) == n, f"tick {t}: two messages for one server"
            want, _ = cpu.step_parallel(msgs)
            if got.tobytes() != want.tobytes():
                bad = int(np.flatnonzero((got.view(np.uint8).reshape(n, 64) != want.view(np.uint8).reshape(n, 64)).any(axis=1))[0])
                raise AssertionError(f"tick {t} slot {bad}: msg={msgs[bad]}\n gpu={got[bad]}\n cpu={want[bad]}")
            flags_seen |= int(np.bitwise_or.reduce(want["flags"]))
            kinds += np.bincount(msgs["kind"], minlength=abi.N_KINDS)[:abi.N_KINDS]
            n_dec += n
            if (t + 1) % checksum_every == 0 or t + 1 == ticks:
                want_sum = checksum_of_checksums(oracle_lib.server_checksums(cpu.get_state()))
                assert gpu.state_checksum() == want_sum, f"state checksum differs after tick {t}"
        final_gpu, final_cpu = gpu.get_state(), cpu.get_state()
    cpu.close()
    assert final_gpu.tobytes() == final_cpu.tobytes(), "final state differs"
    return n_dec, flags_seen, kinds, final_cpu


def test_config3_every_tick_of_a_512_tick_stream(engine_mod, oracle_lib):
    G, N, seed, ticks = 65536, 5, 0x5EED0003, 512
    n_dec, flags, kinds, st = replay_device_stream(engine_mod, oracle_lib, G, N, seed, ticks, 32)
    assert n_dec > ticks * 180_000                                   # ~212 k decisions per tick
    # the aged regime was really reached and exercised: long run tables, overflow / compaction, elections, repairs
    assert (st["n_runs"] >= 8).sum() > 1000
    for kind in (abi.MSG_AER, abi.MSG_AER_REPLY, abi.MSG_REQUEST_VOTE, abi.MSG_WRITTEN, abi.MSG_APPEND,
                 abi.MSG_ELECTION_TIMEOUT, abi.MSG_PRE_VOTE_RESULT, abi.MSG_VOTE_RESULT, abi.MSG_SNAPSHOT_WRITTEN,
                 abi.MSG_HEARTBEAT_RPC, abi.MSG_HEARTBEAT_REPLY, abi.MSG_CONSISTENT_QUERY):
        assert kinds[kind] > 0, f"kind {kind} never generated"
    for f in (abi.F_BECAME_LEADER, abi.F_ROLE_CHANGED, abi.F_WROTE, abi.F_APPLIED, abi.F_PIPELINE, abi.F_PERSIST,
              abi.F_REPROCESSED, abi.F_SEND_VOTE_REQUESTS, abi.F_PRE_VOTE_REQS):
        assert flags & f, f"flag {f:#x} never seen"
    print("flags seen", hex(flags), "kinds", kinds.tolist(), "n_runs hist", np.bincount(st["n_runs"], minlength=17).tolist())


def test_seven_members_64_ticks_at_65536_groups(engine_mod, oracle_lib):
    G, N, seed, ticks = 65536, 7, 0x5EED0003, 64
    n_dec, flags, kinds, _ = replay_device_stream(engine_mod, oracle_lib, G, N, seed, ticks, 16)
    assert n_dec > ticks * 250_000
    assert flags & abi.F_BECAME_LEADER and flags & abi.F_PIPELINE and kinds[abi.MSG_AER_REPLY] > 0


def test_config5_repair_backlogs_at_16384_groups(engine_mod, oracle_lib):
    """SURVEY 8(d) config 5 at a quarter of its BASELINE size (the checker's per-index logs of the full size need
    ~30 s just to build; bench.py runs and oracle-checks the FULL 65 536 x 7 configuration every time it runs):
    append_entries with prev_log_index inside 1 024-entry backlogs, wrong prev_log_term half of the time, failed
    replies driving the leader's repair, over 24 ticks."""
    from ra_amd import workload as W
    G, N, seed, ticks = 16384, 7, 0x5EED0005, 24
    st = W.initial_states(G, N, seed, backlog=1024, boundaries=(3, 6))
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st)
    seen, n_dec = 0, 0
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_capacity=G * N, ring_slots=2) as gpu:
        gpu.set_state(0, st)
        for t in range(ticks):
            m = W.gen_tick(cpu.get_state(), N, t, seed, W.MIX_CONFIG5, backlog_mode=True)
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            if dg.tobytes() != do.tobytes():
                bad = int(np.flatnonzero((dg.view(np.uint8).reshape(-1, 64) != do.view(np.uint8).reshape(-1, 64)).any(axis=1))[0])
                raise AssertionError(f"tick {t} slot {bad}: msg={m[bad]}\n gpu={dg[bad]}\n cpu={do[bad]}")
            assert fuzz.sort_rpcs(rg).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"tick {t}: rpcs differ"
            seen |= int(np.bitwise_or.reduce(do["flags"]))
            n_dec += len(m)
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "final state differs"
        assert gpu.state_checksum() == checksum_of_checksums(oracle_lib.server_checksums(cpu.get_state()))
    assert (cpu.get_state()["role"] == abi.ROLE_AWAIT_CONDITION).sum() > G // 4
    assert seen & abi.F_REPLY and seen & abi.F_PIPELINE and n_dec > ticks * 20_000
    cpu.close()


def test_config5_repair_backlogs_at_full_size(engine_mod, oracle_lib):
    """VERDICT round 4, "What's weak" 1 (i): SURVEY 8(d) config 5 at its BASELINE size -- 65 536 groups x 7 -- as a
    test of its own (bench.py checks the same configuration in the driver's run): 8 ticks of the log-matching repair
    stream, every decision, every rpc record, the whole final state and the checksum of checksums."""
    from ra_amd import workload as W
    G, N, seed, ticks = 65536, 7, 0x5EED0005, 8
    st = W.initial_states(G, N, seed, backlog=1024, boundaries=(3, 6))
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st)
    n_dec = 0
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_capacity=G * N, ring_slots=2) as gpu:
        gpu.set_state(0, st)
        for t in range(ticks):
            m = W.gen_tick(cpu.get_state(), N, t, seed, W.MIX_CONFIG5, backlog_mode=True)
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            if dg.tobytes() != do.tobytes():
                bad = int(np.flatnonzero((dg.view(np.uint8).reshape(-1, 64) != do.view(np.uint8).reshape(-1, 64)).any(axis=1))[0])
                raise AssertionError(f"tick {t} slot {bad}: msg={m[bad]}\n gpu={dg[bad]}\n cpu={do[bad]}")
            assert fuzz.sort_rpcs(rg).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"tick {t}: rpcs differ"
            n_dec += len(m)
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "final state differs"
        assert gpu.state_checksum() == checksum_of_checksums(oracle_lib.server_checksums(cpu.get_state()))
    assert n_dec > ticks * 80_000
    cpu.close()


def test_config4_one_rank_of_262144_groups_hashed_over_eight(engine_mod, oracle_lib):
    """VERDICT round 4, "What's weak" 1 (ii): BASELINE configs[3] -- 262 144 groups x 5 hashed over 8 GPUs -- has one
    rank's share on one GPU: the groups rgb_route (= shard.owner, splitmix64) gives rank 3 of 8, the closed-loop stream
    over them for 32 ticks against the oracle (every decision, checksums, the final state), and the rank's leaderboard
    rows at the two 16-tick boundaries against the host restatement over the oracle's state."""
    import torch
    from ra_amd import shard
    from ra_amd import workload as W
    total, world, rank, N, seed = 262144, 8, 3, 5, 0x5EED0004
    mine = shard.local_group_ids(total, world, rank)
    lib = engine_mod.lib()
    assert all(lib.rgb_route(int(g), world) == rank for g in mine[:2000])
    assert 0.9 * total / world < len(mine) < 1.1 * total / world
    G = len(mine)
    S = G * N
    st0 = W.initial_states(G, N, seed)
    cpu = oracle_lib.Oracle(G, N, max_runs=16)
    cpu.set_state(0, st0)
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64) as gpu:
        gpu.set_state(0, st0)
        dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dr = torch.zeros(S * (N - 1) * 56, dtype=torch.uint8, device="cuda")
        dn = torch.zeros(1, dtype=torch.int32, device="cuda")
        for t in range(32):
            gpu.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), sp)
            gpu.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
            stream.synchronize()
            n = int(dn.item())
            msgs = dm[:n * 64].cpu().numpy().view(abi.MSG_DTYPE)
            got = abi.expand_decisions(dd[:n * 64].cpu().numpy().view(abi.DECISION_DTYPE))
            want, _ = cpu.step_parallel(msgs)
            assert got.tobytes() == want.tobytes(), f"tick {t}: decisions differ"
            if (t + 1) % 16 == 0:
                rows = gpu.snapshot()
                assert rows.tobytes() == shard.leaderboard_rows_from_states(cpu.get_state(), N).tobytes(), f"leaderboard after tick {t}"
                assert gpu.state_checksum() == checksum_of_checksums(oracle_lib.server_checksums(cpu.get_state()))
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "final state differs"
    cpu.close()


def test_config3_trains_against_the_oracle_at_full_size(engine_mod, oracle_lib):
    """VERDICT round 3, "What's weak" 1: rgb_train_kernel at 65 536 x 5 against the ORACLE itself (not through the
    per-tick launches): 64 ageing ticks, then 64 ticks generated by the stamping load generator and replayed from the
    aged state as four 16-tick train launches -- every decision of every tick and the whole final state compared with
    Oracle.step_parallel on the same stream."""
    import torch
    from ra_amd import workload as W
    G, N, seed, age, T, per = 65536, 5, 0x5EED0003, 64, 64, 16
    S, tb = G * N, G * N * 64
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64) as gpu:
        gpu.set_state(0, W.initial_states(G, N, seed))
        dm = torch.zeros(T * tb, dtype=torch.uint8, device="cuda")
        dd = torch.zeros(T * tb, dtype=torch.uint8, device="cuda")
        ds = torch.zeros(T * S, dtype=torch.uint8, device="cuda")
        dr = torch.zeros(4 * S * (N - 1) * 56, dtype=torch.uint8, device="cuda")
        dn = torch.zeros(T, dtype=torch.int32, device="cuda")
        bc = torch.zeros(T * engine_mod.TRAIN_BUCKETS, dtype=torch.int32, device="cuda")
        for t in range(age):                                 # ageing: applied, not kept
            gpu.synth_tick_device(seed, t, dm.data_ptr(), 0, 0, sp)
            gpu.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
        stream.synchronize()
        st_aged = gpu.get_state()
        for t in range(T):                                   # the stream: generated (and stamped) tick by tick
            gpu.synth_tick_stamped_device(seed, age + t, dm.data_ptr() + t * tb, 0, dn.data_ptr() + t * 4,
                                          bc.data_ptr() + t * engine_mod.TRAIN_BUCKETS * 4, ds.data_ptr() + t * S, sp)
            gpu.synth_apply_tick_device(dm.data_ptr() + t * tb, S, dd.data_ptr(), dr.data_ptr(), sp)
        stream.synchronize()
        counts = dn.cpu().numpy().astype(np.int64)
        buckets = bc.cpu().numpy().reshape(T, engine_mod.TRAIN_BUCKETS).astype(np.uint32)
        plan = gpu.train_plan(buckets)
        gpu.set_state(0, st_aged)
        dd.zero_()
        for t in range(0, T, per):
            gpu.train_run_device(plan, t, per, dm.data_ptr(), ds.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), 4, sp)
        stream.synchronize()
        assert gpu.train_status()[0] == 0
        cpu = oracle_lib.Oracle(G, N, max_runs=16)
        cpu.set_state(0, st_aged)
        n_dec = 0
        for t in range(T):
            n = int(counts[t])
            msgs = dm[t * tb:t * tb + n * 64].cpu().numpy().view(abi.MSG_DTYPE)
            got = abi.expand_decisions(dd[t * tb:t * tb + n * 64].cpu().numpy().view(abi.DECISION_DTYPE))
            want, _ = cpu.step_parallel(msgs)
            if got.tobytes() != want.tobytes():
                bad = int(np.flatnonzero((got.view(np.uint8).reshape(n, 64) != want.view(np.uint8).reshape(n, 64)).any(axis=1))[0])
                raise AssertionError(f"train, tick {t} slot {bad}: msg={msgs[bad]}\n gpu={got[bad]}\n cpu={want[bad]}")
            n_dec += n
        assert n_dec > T * 180_000
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "final state differs from the oracle's"
        assert gpu.state_checksum() == checksum_of_checksums(oracle_lib.server_checksums(cpu.get_state()))
        plan.close()
        cpu.close()
