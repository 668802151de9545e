"""Parity of the HIP path (through the C ABI) with the CPU checker and with the vectors
transcribed from the reference's own tests.  Bit-exact: decisions, pipelined rpcs and the full
final state must be identical.  Runs only on the GPU box (-m gpu)."""
import os

import numpy as np
import pytest

import fuzz
import vector_runner as VR
from ra_amd import abi

pytestmark = pytest.mark.gpu

DATA = VR.load()


@pytest.fixture(scope="module")
def engine_mod():
    from ra_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()    # a fresh checkout on the GPU box: hipcc is there, the .so is not in git
    engine.lib()          # raises if the HIP library is missing: no fallback
    return engine


def assert_same(tag, dg, rg, sg, do, ro, so):
    if dg.tobytes() != do.tobytes():
        for i in range(len(dg)):
            if dg[i].tobytes() != do[i].tobytes():
                raise AssertionError(f"{tag}: decision {i} differs\n gpu={dg[i]}\n cpu={do[i]}")
    rg, ro = fuzz.sort_rpcs(rg), fuzz.sort_rpcs(ro)
    assert len(rg) == len(ro), f"{tag}: rpc count gpu={len(rg)} cpu={len(ro)}"
    if rg.tobytes() != ro.tobytes():
        for i in range(len(rg)):
            if rg[i].tobytes() != ro[i].tobytes():
                raise AssertionError(f"{tag}: rpc {i} differs\n gpu={rg[i]}\n cpu={ro[i]}")
    if sg.tobytes() != so.tobytes():
        for i in range(len(sg)):
            if sg[i].tobytes() != so[i].tobytes():
                diff = [n for n in sg.dtype.names if np.any(sg[i][n] != so[i][n])]
                raise AssertionError(f"{tag}: state of server {i} differs in {diff}\n gpu={sg[i]}\n cpu={so[i]}")


@pytest.mark.parametrize("v", DATA["vectors"], ids=[v["id"] for v in DATA["vectors"]])
def test_hip_matches_reference_vector(engine_mod, v):
    VR.run_vector(lambda g, n: engine_mod.RaGpuBatch(g, n, ring_capacity=64, ring_slots=2), v)


# the cases from seed 107 on carry >= 4096 messages per round: rgb_submit then takes the class-dispatch kernel
# (one compile-time specialised path per message kind) instead of the kind-generic one
@pytest.mark.parametrize("n_members,seed,groups", [(3, 101, 300), (5, 102, 400), (7, 103, 300),
                                                   (8, 104, 150), (1, 105, 50), (2, 106, 100),
                                                   (5, 107, 1300), (3, 108, 2200), (7, 109, 900),
                                                   (8, 110, 600), (1, 111, 4300), (2, 112, 2200),
                                                   (4, 113, 1100), (6, 114, 800)])
def test_hip_equals_oracle_on_random_ticks(engine_mod, oracle_lib, n_members, seed, groups):
    rng = np.random.default_rng(seed)
    st = fuzz.random_states(rng, groups, n_members, max_runs=6)
    cpu = oracle_lib.Oracle(groups, n_members)
    cpu.set_state(0, st)
    # max_runs=16 and 6 ticks (<= 2 new runs each on top of <= 4): the run table cannot overflow,
    # which the checker (explicit per-index log) does not model
    with engine_mod.RaGpuBatch(groups, n_members, ring_capacity=max(4096, groups * n_members), ring_slots=2,
                               max_runs=16) as gpu:
        gpu.set_state(0, st)
        assert gpu.get_state().tobytes() == st.tobytes(), "upload/download is not the identity"
        seen_flags = 0
        seen_inv = set()
        for tick in range(6):
            cur = cpu.get_state()
            msgs = fuzz.random_msgs(rng, cur, n_members)
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"N={n_members} tick {tick}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            seen_flags |= int(np.bitwise_or.reduce(do["flags"])) if len(do) else 0
            seen_inv |= set(do["invariant"].tolist())
            # checksum of checksums agrees with the checker's per-server checksums
            want = engine_mod.combine_checksums(oracle_lib.server_checksums(cpu.get_state()))
            assert gpu.state_checksum() == want
        if n_members >= 3:
            for f in ("REPLY", "PERSIST", "LEADER_MSG", "APPLIED", "WROTE", "TRUNCATED", "PIPELINE",
                      "REPROCESSED", "ROLE_CHANGED", "UNHANDLED", "INVARIANT", "REPLY_PRE_VOTE",
                      "SEND_VOTE_REQUESTS", "PRE_VOTE_REQS", "BECAME_LEADER", "START_ELECTION_TIMEOUT",
                      "REPLY_HEARTBEAT", "SEND_HEARTBEATS", "QUERY_QUORUM", "RESEND_PENDING"):
                assert seen_flags & VR.FLAG[f], f"fuzz never produced {f}"


def test_same_server_messages_are_serialised_in_submission_order(engine_mod, oracle_lib):
    """Several messages for one server inside one submit are applied in order (sub-ticks),
    exactly like the sequential checker (= the gen_statem mailbox)."""
    rng = np.random.default_rng(7)
    G, N = 64, 5
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=8192, ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st)
        for rnd in range(2):
            parts = [fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.6) for _ in range(3)]
            msgs = np.concatenate(parts)
            rng.shuffle(msgs)
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"round {rnd}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())


@pytest.mark.parametrize("table_runs", [6, 16], ids=["shallow_run_tables", "deep_run_tables"])
def test_rounds_of_one_batch_run_as_one_train_launch(engine_mod, oracle_lib, table_runs, G=1500, N=5, batches=3,
                                                      wal_down_share=0.0):
    """The normal shape of a real batch: a leader's N-1 replies arrive together, so does a follower's append and its
    written event -- four messages per leader, two or three per follower in ONE rgb_submit.  Rounds 2..16 of a big
    batch run as one train launch (rgb_submit_trains counts them); the result is the sequential checker's."""
    rng = np.random.default_rng(21 + table_runs + 1000 * int(os.environ.get("RGB_FUZZ_SEED_OFFSET", "0")))   # more seeds: tools/gpu_fuzz_rounds.py
    # deep tables (up to 13 in-memory runs): a leader-side train wavefront serves runs 0..7 from LDS and the rest from
    # memory (run_pair in rgb_kernels.hip) -- both sides of that border are walked
    st = fuzz.random_states(rng, G, N, max_runs=table_runs, backlog=24 if table_runs <= 8 else 60)
    rng2 = np.random.default_rng(4242 + table_runs)
    if wal_down_share:
        # a share of the servers waits in one of the two wal_down conditions (the follower's, the leader's): inside a
        # train their messages are dropped, or -- with RGB_MF_CAN_WRITE -- re-processed by the role they return to
        pick = rng2.random(G * N) < wal_down_share
        st["role"][pick] = abi.ROLE_AWAIT_CONDITION
        st["cond_reason"][pick] = rng2.choice([abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER], size=int(pick.sum()))
    cpu = oracle_lib.Oracle(G, N, max_runs=16)              # the device's bound: deep tables overflow it now and then
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=65536, ring_slots=2, max_runs=16, flags=abi.CFG_SUBMIT_TRAINS) as gpu:
        gpu.set_state(0, st)
        fused = 0
        for b in range(batches):
            parts = [fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.9) for _ in range(4)]
            msgs = np.concatenate(parts)
            msgs = msgs[msgs["kind"] != abi.MSG_NOP]
            rng.shuffle(msgs)
            if wal_down_share:
                msgs["flags"] |= np.where(rng2.random(len(msgs)) < 0.5, abi.MF_CAN_WRITE, 0).astype(msgs["flags"].dtype)
                waiting = cpu.get_state()["role"][msgs["server"]] == abi.ROLE_AWAIT_CONDITION
                msgs["kind"][waiting & (rng2.random(len(msgs)) < 0.2)] = abi.MSG_AWAIT_TIMEOUT
            assert len(msgs) >= 4096
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"batch {b}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            fused = gpu.submit_trains()
        assert fused == batches, f"{fused} of {batches} batches took the train path"
        # a device-side train in between makes the host's copy of the sequence bytes stale: it is refreshed
        small = fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.2)
        do, ro = cpu.step(small); dg, rg = gpu.step(small)
        assert_same("small batch (one launch per round)", dg, rg, gpu.get_state(), do, ro, cpu.get_state())


def test_wal_down_conditions_inside_a_train(engine_mod, oracle_lib):
    """A quarter of the servers waits in one of the two wal_down conditions while the rounds of a batch run as one
    train launch: dropped messages, re-processing by handle_follower/2 and by handle_leader/2 (and from there once more
    by handle_follower/2), the leader's timeout effect -- per-server order by the sequence bytes."""
    test_rounds_of_one_batch_run_as_one_train_launch(engine_mod, oracle_lib, 6, G=1500, N=5, batches=2, wal_down_share=0.25)


def wal_down_reupload(before: np.ndarray, dec: np.ndarray) -> np.ndarray:
    """INTEGRATION.md's host recipe for `ra_log:write/2 -> {error, wal_down}` (src/ra_server.erl:1377-1385): the state
    to re-upload for ONE server, from its state before the append_entries_rpc and the decision that carried WROTE.
    The reference keeps State1 (term, leader_id, commit_index := LeaderCommit) and returns await_condition WITHOUT
    evaluate_commit_index_follower/2: the log cursors, `pending` and last_applied are what they were before the
    message, and the APPLIED / AUX_EVAL flags of that decision are ignored (its entries are not in the log)."""
    new = before.copy()
    new["current_term"] = dec["reply_term"] if dec["flags"] & abi.F_REPLY else max(int(before["current_term"]), 0)
    new["commit_index"] = dec["commit_index"]
    new["role"] = abi.ROLE_AWAIT_CONDITION
    new["cond_reason"] = abi.COND_WAL_DOWN
    return new


def test_wal_down_host_recipe_keeps_last_applied_and_the_log(engine_mod, oracle_lib):
    """the append_entries_rpc decision may carry APPLIED together with WROTE (the device applied up to
    min(last_index, leader_commit), entries included that the WAL then refused).  The host recipe drops that: the
    re-uploaded server has the OLD last_applied and log, the new term / leader / commit_index, role await_condition
    (wal_down) -- and from there engine and checker agree on what the next messages do (W1 pins the state itself)."""
    rng = np.random.default_rng(33)
    G, N = 8, 3
    st = fuzz.random_states(rng, G, N, max_runs=4)
    # a follower whose leader is member 0, log [1..10] in term 3, everything written and applied up to 6
    s = 1
    f = st[s:s + 1].copy()
    f["role"] = abi.ROLE_FOLLOWER; f["cond_reason"] = abi.COND_NONE; f["current_term"] = 3; f["leader_id"] = 0
    f["voted_for"] = 0; f["first_index"] = 1; f["last_index"] = 10; f["last_term"] = 3
    f["last_written_index"] = 10; f["last_written_term"] = 3; f["commit_index"] = 6; f["last_applied"] = 6
    f["snapshot_index"] = abi.UNDEF; f["snapshot_term"] = abi.UNDEF
    f["n_runs"] = 1; f["run_start"][0][0] = 1; f["run_term"][0][0] = 3; f["pending_first"] = 11; f["n_pending_old"] = 0
    st[s] = f[0]
    cpu = oracle_lib.Oracle(G, N); cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=64, ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st)
        m = np.zeros(1, dtype=abi.MSG_DTYPE)
        m["server"] = s; m["kind"] = abi.MSG_AER; m["from"] = 0; m["term"] = 3
        m["a"] = 10; m["b"] = 3; m["c"] = 12; m["n_entries"] = 2; m["n_run0"] = 2; m["run0_term"] = 3
        dg, _ = gpu.step(m); do, _ = cpu.step(m)
        assert dg.tobytes() == do.tobytes()
        d = dg[0]
        assert d["flags"] & abi.F_WROTE and d["flags"] & abi.F_APPLIED and d["last_applied"] == 12
        # ra_log:write/2 answered {error, wal_down}: the host recipe
        new = wal_down_reupload(st[s], d)
        assert new["last_applied"] == 6 and new["last_index"] == 10 and new["commit_index"] == 12
        for target in (gpu, cpu):
            target.set_state(s, new.reshape(1))
        assert gpu.get_state(s, 1).tobytes() == cpu.get_state(s, 1).tobytes()
        # while the WAL is down the resend is dropped; once ra_log:can_write/1 is true it is re-processed as a follower
        for flags, wrote in ((0, False), (abi.MF_CAN_WRITE, True)):
            m["flags"] = flags
            dg, _ = gpu.step(m); do, _ = cpu.step(m)
            assert dg.tobytes() == do.tobytes()
            assert bool(dg[0]["flags"] & abi.F_WROTE) == wrote
            assert gpu.get_state(s, 1).tobytes() == cpu.get_state(s, 1).tobytes()
        assert gpu.get_state(s, 1)["last_applied"][0] == 12 and gpu.get_state(s, 1)["role"][0] == abi.ROLE_FOLLOWER
    cpu.close()


@pytest.mark.parametrize("n_members,seed,groups,ticks", [(3, 71, 160, 5), (5, 72, 160, 5), (7, 73, 160, 5), (1, 74, 160, 5),
                                                         (5, 75, 1400, 2)])
def test_wal_down_conditions_follower_and_leader_match_oracle(engine_mod, oracle_lib, n_members, seed, groups, ticks):
    """await_condition with wal_down_condition/2 in both forms: the follower's (src/ra_server.erl:1377-1385: back to
    follower, no timeout effects) and the leader's (:660-668: transition_to => leader, the timeout hands out
    {transfer_leadership, Peer} while the log still cannot be written).  A third of the servers start in one of the
    two, every message kind arrives at them with and without RGB_MF_CAN_WRITE; a message re-processed by
    handle_leader/2 may step down and be re-processed once more by handle_follower/2."""
    rng = np.random.default_rng(seed)
    G, N = groups, n_members
    st = fuzz.random_states(rng, G, N, max_runs=6)
    pick = rng.random(G * N) < 0.34
    st["role"][pick] = abi.ROLE_AWAIT_CONDITION
    st["cond_reason"][pick] = rng.choice([abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER], size=int(pick.sum()))
    st["cond_reason"][~pick & (st["role"] != abi.ROLE_AWAIT_CONDITION)] = abi.COND_NONE
    cpu = oracle_lib.Oracle(G, N); cpu.set_state(0, st)
    seen_transfer = seen_leader_reprocess = 0
    with engine_mod.RaGpuBatch(G, N, ring_capacity=4096, ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st)
        # (5, 75, 1400, 2): >= 4096 messages per round, rgb_submit takes the class-dispatch kernel
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()      # cond_reason 4 survives the packed word
        assert gpu.state_checksum() == engine_mod.combine_checksums(oracle_lib.server_checksums(cpu.get_state()))
        for t in range(ticks):
            cur = cpu.get_state()
            msgs = fuzz.random_msgs(rng, cur, N)
            msgs["flags"] |= np.where(rng.random(len(msgs)) < 0.5, abi.MF_CAN_WRITE, 0).astype(msgs["flags"].dtype)
            waiting = cur["role"][msgs["server"]] == abi.ROLE_AWAIT_CONDITION
            timeout = waiting & (rng.random(len(msgs)) < 0.25)
            msgs["kind"][timeout] = abi.MSG_AWAIT_TIMEOUT
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert dg.tobytes() == do.tobytes(), f"tick {t}: decisions differ"
            assert fuzz.sort_rpcs(rg).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"tick {t}: rpcs differ"
            assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), f"tick {t}: state differs"
            seen_transfer += int(((dg["flags"] & abi.F_TRANSFER_LEADERSHIP) != 0).sum())
            was_ldr_cond = cur["cond_reason"][msgs["server"]] == abi.COND_WAL_DOWN_LEADER
            seen_leader_reprocess += int((was_ldr_cond & ((dg["flags"] & abi.F_REPROCESSED) != 0)).sum())
            # the effect exists only for the leader's condition, on its timeout, with the WAL still down and a peer to name
            tl = (dg["flags"] & abi.F_TRANSFER_LEADERSHIP) != 0
            assert not (tl & ~(was_ldr_cond & (msgs["kind"] == abi.MSG_AWAIT_TIMEOUT) &
                               ((msgs["flags"] & abi.MF_CAN_WRITE) == 0))).any()
            assert (dg["role"][tl] == abi.ROLE_LEADER).all()
            # keep a share of the servers in the two conditions for the next tick
            if t + 1 < ticks:
                cur = cpu.get_state()
                again = rng.random(G * N) < 0.2
                cur["role"][again] = abi.ROLE_AWAIT_CONDITION
                cur["cond_reason"][again] = rng.choice([abi.COND_WAL_DOWN, abi.COND_WAL_DOWN_LEADER], size=int(again.sum()))
                cpu.set_state(0, cur); gpu.set_state(0, cur)
        assert gpu.state_checksum() == engine_mod.combine_checksums(oracle_lib.server_checksums(cpu.get_state()))
    cpu.close()
    if N > 1:
        assert seen_transfer > 0 and seen_leader_reprocess > 0
    else:
        assert seen_transfer == 0          # maps:remove(Self, Cluster) is empty: no effect (src/ra_server.erl:661-662)


def test_leader_wal_down_host_recipe(engine_mod, oracle_lib):
    """The host recipe of INTEGRATION.md for `ra_log:append/2 -> error(wal_down)` on a leader's {command, _}
    (src/ra_server.erl:655-672): the leader is re-uploaded AS IT WAS BEFORE the command in await_condition /
    RGB_COND_WAL_DOWN_LEADER.  While the WAL is down a reply is dropped; the timeout returns to leader with the
    transfer_leadership effect; with the WAL back the reply is re-processed by handle_leader/2 and commits."""
    rng = np.random.default_rng(34)
    G, N = 4, 3
    st = fuzz.random_states(rng, G, N, max_runs=4)
    s = 0
    f = st[s:s + 1].copy()
    f["role"] = abi.ROLE_LEADER; f["cond_reason"] = abi.COND_NONE; f["current_term"] = 3; f["leader_id"] = 0
    f["voted_for"] = 0; f["first_index"] = 1; f["last_index"] = 10; f["last_term"] = 3
    f["last_written_index"] = 10; f["last_written_term"] = 3; f["commit_index"] = 6; f["last_applied"] = 6
    f["snapshot_index"] = abi.UNDEF; f["snapshot_term"] = abi.UNDEF
    f["n_runs"] = 1; f["run_start"][0][0] = 1; f["run_term"][0][0] = 3; f["pending_first"] = 11; f["n_pending_old"] = 0
    f["present_mask"] = 7; f["voter_mask"] = 7; f["status_mask"] = 0xFF; f["backoff_mask"] = 0; f["self_nonvoter"] = 0
    f["match_index"][0][:3] = (10, 6, 6); f["next_index"][0][:3] = (11, 11, 11); f["commit_index_sent"][0][:3] = (6, 6, 6)
    before = f[0].copy()
    st[s] = before
    cpu = oracle_lib.Oracle(G, N); cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=64, ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st)
        m = np.zeros(1, dtype=abi.MSG_DTYPE)
        m["server"] = s; m["kind"] = abi.MSG_APPEND; m["from"] = abi.NONE; m["n_entries"] = 1
        dg, _ = gpu.step(m); do, _ = cpu.step(m)
        assert dg.tobytes() == do.tobytes() and gpu.get_state(s, 1)["last_index"][0] == 11
        # ra_log:append/2 raised wal_down: the leader as it was, waiting
        new = before.copy()
        new["role"] = abi.ROLE_AWAIT_CONDITION; new["cond_reason"] = abi.COND_WAL_DOWN_LEADER
        for target in (gpu, cpu):
            target.set_state(s, new.reshape(1))
        saved = new.copy()
        r = np.zeros(1, dtype=abi.MSG_DTYPE)
        r["server"] = s; r["kind"] = abi.MSG_AER_REPLY; r["from"] = 1; r["term"] = 3; r["flags"] = abi.MF_SUCCESS
        r["a"] = 11; r["b"] = 10; r["c"] = 3
        dg, _ = gpu.step(r); do, _ = cpu.step(r)                                 # WAL down: dropped
        assert dg.tobytes() == do.tobytes() and dg[0]["role"] == abi.ROLE_AWAIT_CONDITION and dg[0]["flags"] == 0
        assert gpu.get_state(s, 1).tobytes() == saved.reshape(1).tobytes()
        t = np.zeros(1, dtype=abi.MSG_DTYPE)
        t["server"] = s; t["kind"] = abi.MSG_AWAIT_TIMEOUT; t["from"] = abi.NONE
        dg, _ = gpu.step(t); do, _ = cpu.step(t)                                 # timeout, WAL still down
        assert dg.tobytes() == do.tobytes() and dg[0]["role"] == abi.ROLE_LEADER
        assert dg[0]["flags"] & abi.F_TRANSFER_LEADERSHIP
        back = gpu.get_state(s, 1)
        assert back["cond_reason"][0] == abi.COND_NONE and back["last_index"][0] == 10
        for target in (gpu, cpu):
            target.set_state(s, saved.reshape(1))
        r["flags"] = abi.MF_SUCCESS | abi.MF_CAN_WRITE
        dg, _ = gpu.step(r); do, _ = cpu.step(r)                                 # WAL back: handle_leader/2 takes the reply
        assert dg.tobytes() == do.tobytes() and dg[0]["role"] == abi.ROLE_LEADER
        assert dg[0]["flags"] & abi.F_REPROCESSED and dg[0]["commit_index"] == 10
        assert gpu.get_state(s, 1).tobytes() == cpu.get_state(s, 1).tobytes()
    cpu.close()


def test_concurrent_producers_and_consumers_on_one_context(engine_mod, oracle_lib, G=256, N=5, P=4, C_=2, per=6):
    """The boundary's threading contract (include/ra_gpu_batch.h): P threads in rgb_submit and C_ threads in rgb_collect
    on ONE context.  Every producer owns a disjoint range of groups, so the order in which the producers' batches
    interleave does not matter: each server sees its own producer's batches in that producer's order, and the result
    must be the checker's -- every batch's decisions (identified by its tick number) and the final state."""
    import threading
    rng = np.random.default_rng(57)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    gp = G // P
    batches, want = {}, {}
    for k in range(P):                                          # producer k: groups [k gp, (k+1) gp)
        for b in range(per):
            cur = cpu.get_state()
            m = fuzz.random_msgs(rng, cur[k * gp * N:(k + 1) * gp * N], N, frac=0.8)
            m["server"] += k * gp * N
            parts = [m]
            if b % 2:                                           # every other batch carries a second round
                m2 = fuzz.random_msgs(rng, cur[k * gp * N:(k + 1) * gp * N], N, frac=0.3)
                m2["server"] += k * gp * N
                parts.append(m2)
            msgs = np.concatenate(parts)
            d, r = cpu.step(msgs)
            batches[(k, b)] = msgs; want[1000 * k + b] = (d, r)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=8192, ring_slots=3, max_runs=16) as gpu:
        gpu.set_state(0, st)
        total = P * per
        got, errs = {}, []
        lock = threading.Lock()
        taken = [0]

        def producer(k):
            try:
                for b in range(per):
                    while True:
                        try:
                            gpu.submit(batches[(k, b)], tick=1000 * k + b); break
                        except engine_mod.RgbError as e:
                            if e.code != abi.E_FULL: raise
            except Exception as e:                               # noqa: BLE001
                errs.append(e)

        def consumer():
            try:
                while True:
                    with lock:
                        if taken[0] >= total: return
                        taken[0] += 1
                    while True:
                        try:
                            d, r, tick = gpu.collect(); break
                        except engine_mod.RgbError as e:
                            if e.code != abi.E_EMPTY: raise
                            gpu.wait(20)
                    with lock:
                        got[tick] = (d.copy(), r.copy())
            except Exception as e:                               # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=producer, args=(k,)) for k in range(P)] + [threading.Thread(target=consumer) for _ in range(C_)]
        for t in ths: t.start()
        for t in ths: t.join()
        assert not errs, errs
        assert sorted(got) == sorted(want)
        for tick, (d, r) in want.items():
            assert got[tick][0].tobytes() == d.tobytes(), f"batch {tick}: decisions"
            assert fuzz.sort_rpcs(got[tick][1].copy()).tobytes() == fuzz.sort_rpcs(r.copy()).tobytes(), f"batch {tick}: rpcs"
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()
    cpu.close()


def test_pipelined_ring_keeps_batches_in_order(engine_mod, oracle_lib):
    """submit/submit/submit then collect x3: the staging ring returns batches oldest first."""
    rng = np.random.default_rng(9)
    G, N = 128, 3
    st = fuzz.random_states(rng, G, N)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=1024, ring_slots=3) as gpu:
        gpu.set_state(0, st)
        batches = []
        for t in range(3):
            # independent of state evolution on purpose: all built from the initial state
            batches.append(fuzz.random_msgs(rng, st, N, frac=0.5))
        for t, b in enumerate(batches):
            gpu.submit(b, tick=100 + t)
        with pytest.raises(engine_mod.RgbError):
            gpu.submit(batches[0])                      # ring full
        for t, b in enumerate(batches):
            do, _ = cpu.step(b)
            dg, _, tick = gpu.collect()
            assert tick == 100 + t
            assert dg.tobytes() == do.tobytes()
        with pytest.raises(engine_mod.RgbError):
            gpu.collect()                               # nothing submitted


def test_collect_view_hands_out_the_slot_in_place(engine_mod, oracle_lib):
    """rgb_collect_view (ABI v9): the oldest batch as pointers into the pinned slot the device wrote -- decisions in
    submission order, rpc records compacted by (msg_index, peer) -- equal to the sequential checker's; the slot stays
    the caller's until rgb_release: the ring is full while it is held, a second release is refused."""
    rng = np.random.default_rng(77)
    G, N = 96, 5
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, ring_capacity=4096, ring_slots=2, max_runs=16) as gpu:
        gpu.set_state(0, st)
        total_rpcs = 0
        for t in range(4):
            # several messages per server in one batch (sub-tick rounds), shuffled: the view is in SUBMISSION order
            msgs = np.concatenate([fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.6) for _ in range(3)])
            rng.shuffle(msgs)
            do, ro = cpu.step(msgs)
            gpu.submit(msgs, tick=7 + t)
            dv, rv, tick, slot = gpu.collect_view()
            assert tick == 7 + t and len(dv) == len(msgs)
            assert dv.tobytes() == do.tobytes()
            assert len(rv) == len(ro) and fuzz.sort_rpcs(rv.copy()).tobytes() == fuzz.sort_rpcs(ro).tobytes()
            assert np.all(np.diff(rv["msg_index"].astype(np.int64)) >= 0), "records are ordered by msg_index"
            total_rpcs += len(rv)
            if t == 0:
                # the held slot is not free: the ring (two slots) takes one more batch, then it is full
                nop = np.zeros(4, dtype=abi.MSG_DTYPE)
                gpu.submit(nop)
                with pytest.raises(engine_mod.RgbError) as e:
                    gpu.submit(nop)
                assert e.value.code == abi.E_FULL
                assert dv.tobytes() == do.tobytes()          # ... and its contents are untouched
                gpu.release(slot)
                gpu.collect()                                # the NOP batch
            else:
                gpu.release(slot)
            with pytest.raises(engine_mod.RgbError) as e:
                gpu.release(slot)                            # not held any more
            assert e.value.code == abi.E_STATE
        assert total_rpcs > 0
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()
        # nothing in flight: the view form reports it like rgb_collect; a slot the ring does not have is refused
        with pytest.raises(engine_mod.RgbError) as e:
            gpu.collect_view()
        assert e.value.code == abi.E_EMPTY
        with pytest.raises(engine_mod.RgbError) as e:
            gpu.release(99)
        assert e.value.code == abi.E_INVAL
        # an empty batch is a batch: zero decisions, zero records, its tick
        gpu.submit(np.zeros(0, dtype=abi.MSG_DTYPE), tick=41)
        dv, rv, tick, slot = gpu.collect_view()
        assert (len(dv), len(rv), tick) == (0, 0, 41)
        gpu.release(slot)
    cpu.close()


def test_big_batches_pipelined_through_the_copy_stream(engine_mod, oracle_lib, G=8192, N=5, ticks=4):
    """Batches of 2 MB and more copy in on their own stream (batch k + 1 under batch k's kernels and results stores):
    three of them in flight at once, every decision, rpc record and the final state equal to the sequential checker's."""
    rng = np.random.default_rng(2026)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    batches, want = [], []
    for t in range(ticks):
        m = fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.95)
        m = m[m["kind"] != abi.MSG_NOP]
        assert len(m) * 68 >= (2 << 20), "the batch must be big enough for the copy stream"
        batches.append(m)
        want.append(cpu.step(m))
    with engine_mod.RaGpuBatch(G, N, ring_capacity=G * N, ring_slots=4, max_runs=16) as gpu:
        gpu.set_state(0, st)
        got, pending = [], 0
        for t, m in enumerate(batches):
            while pending >= 3:
                got.append(gpu.collect()); pending -= 1
            gpu.submit(m, tick=t); pending += 1
        while pending:
            got.append(gpu.collect()); pending -= 1
        for t, ((dg, rg, tick), (do, ro)) in enumerate(zip(got, want)):
            assert tick == t
            assert dg.tobytes() == do.tobytes(), f"batch {t}: decisions differ"
            assert fuzz.sort_rpcs(rg.copy()).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"batch {t}: rpc records differ"
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()
    cpu.close()


def test_device_resident_ticks_match_host_path(engine_mod, oracle_lib):
    import torch
    rng = np.random.default_rng(21)
    G, N = 512, 5
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    ticks = []
    decs = []
    for t in range(4):
        m = fuzz.random_msgs(rng, cpu.get_state(), N, frac=0.5)
        pad = np.zeros(G * N - len(m), dtype=abi.MSG_DTYPE)     # NOP padding to a fixed width
        m = np.concatenate([m, pad])
        d, _ = cpu.step(m)
        ticks.append(m)
        decs.append(d)
    allm = np.concatenate(ticks)
    with engine_mod.RaGpuBatch(G, N, max_runs=16) as gpu:
        gpu.set_state(0, st)
        dm = torch.from_numpy(allm.view(np.uint8).reshape(-1)).cuda()
        dd = torch.zeros(len(allm) * 64, dtype=torch.uint8, device="cuda")
        dr = torch.zeros(G * N * (N - 1) * 56, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            gpu.run_ticks_device(dm.data_ptr(), G * N, 4, dd.data_ptr(), dr.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        # the rpc slots hold the LAST tick's pipelined rpcs
        last = decs[-1]
        slots = dr.cpu().numpy().view(abi.RPC_DTYPE).reshape(G * N, N - 1)
        got_r = np.concatenate([slots[i, :int(last["n_rpcs"][i])] for i in range(G * N)]
                               + [np.zeros(0, dtype=abi.RPC_DTYPE)])
        assert int(last["n_rpcs"].sum()) == len(got_r)
        assert np.all(got_r["msg_index"] >= 3 * G * N)
        got = abi.expand_decisions(dd.cpu().numpy().view(abi.DECISION_DTYPE))
        want = np.concatenate(decs)
        assert got.tobytes() == want.tobytes()
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()


def test_leaderboard_snapshot(engine_mod):
    G, N = 16, 5
    st = abi.empty_server_states(G, N)
    st["current_term"] = np.repeat(np.arange(G, dtype=np.uint64) + 1, N)
    for g in range(G):
        if g % 4 != 3:
            l = g % N
            st["role"][g * N + l] = abi.ROLE_LEADER
            st["commit_index"][g * N + l] = 10 + g
            st["last_applied"][g * N + l] = 5 + g
    with engine_mod.RaGpuBatch(G, N) as gpu:
        gpu.set_state(0, st)
        rows = gpu.snapshot()
    for g in range(G):
        if g % 4 != 3:
            assert int(rows["leader"][g]) == g % N and int(rows["n_leaders"][g]) == 1
            assert int(rows["commit_index"][g]) == 10 + g and int(rows["last_applied"][g]) == 5 + g
        else:
            assert int(rows["leader"][g]) == abi.NONE and int(rows["n_leaders"][g]) == 0
        assert int(rows["term"][g]) == g + 1


def test_run_table_overflow_is_flagged(engine_mod):
    """More term changes than max_runs: the oldest run is dropped, first_index raised, flag set."""
    with engine_mod.RaGpuBatch(1, 3, max_runs=3, ring_capacity=16, ring_slots=1) as gpu:
        flags = 0
        for t in range(1, 6):
            m = np.zeros(1, dtype=abi.MSG_DTYPE)
            m["server"] = 1
            m["kind"] = abi.MSG_AER
            m["from"] = 0
            m["term"] = t
            m["a"], m["b"] = t - 1, (t - 1)
            m["c"] = 0
            m["n_entries"], m["n_run0"], m["run0_term"] = 1, 1, t
            d, _ = gpu.step(m)
            flags |= int(d["flags"][0])
            assert int(d["flags"][0]) & abi.F_WROTE
        st = gpu.get_state()[1]
        assert flags & abi.F_RUNS_OVERFLOW
        assert int(st["n_runs"]) == 3 and int(st["last_index"]) == 5 and int(st["last_term"]) == 5
        assert int(st["first_index"]) == int(st["run_start"][0]) == 3


@pytest.mark.parametrize("n_run0", [3, 1, 2])
def test_write_below_first_index_keeps_the_range_start(engine_mod, oracle_lib, n_run0):
    """A sparse log (snapshot at 44, only entry 47 in the range) overwritten from 45: the range
    keeps its Start (src/ra_log.erl:1617-1622 `{Start, _} -> ra_range:new(Start, LastIdx)`), so the
    exported term runs begin at first_index, wherever the message's own term boundary lies."""
    st = abi.empty_server_states(1, 3)
    for i in range(3):
        st["current_term"][i] = 3
        st["role"][i] = abi.ROLE_FOLLOWER
        abi.set_log(st, i, [(47, 2)], last_written=(44, 2), snapshot=(44, 2))
        st["commit_index"][i] = st["last_applied"][i] = 44
        st["pending_first"][i] = 45
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"], m["kind"], m["from"], m["term"] = 1, abi.MSG_AER, 0, 3
    m["a"], m["b"], m["c"] = 44, 2, 47
    m["n_entries"], m["n_run0"] = 3, n_run0
    m["run0_term"], m["run1_term"] = (3, 3) if n_run0 == 3 else (2, 3)
    cpu = oracle_lib.Oracle(1, 3)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(1, 3, ring_capacity=16, ring_slots=1) as gpu:
        gpu.set_state(0, st)
        do, ro = cpu.step(m)
        dg, rg = gpu.step(m)
        so = cpu.get_state()
        assert_same(f"sparse overwrite n_run0={n_run0}", dg, rg, gpu.get_state(), do, ro, so)
        assert int(do["flags"][0]) & abi.F_WROTE
        assert int(so["first_index"][1]) == 47 == int(so["run_start"][1][0]) and int(so["last_index"][1]) == 47


def test_write_that_ends_below_a_sparse_range_leaves_no_range(engine_mod, oracle_lib):
    """A sparse range [47..48] behind a snapshot at 44, overwritten by ONE entry at 45: the reference computes
    `ra_range:new(Start = 47, LastIdx = 45)` (src/ra_log.erl:1617-1622) and new/2 with Start > End is `undefined`
    (src/ra_range.erl:41-50) -- no range is left, last_index_term/1 (:831-835) is the snapshot's again, nothing above
    the snapshot is applied, and the decision still names the entry that went to the WAL.  (Found by the fused-rounds
    fuzz test; device and checker both kept a half-defined range here.)"""
    st = abi.empty_server_states(1, 3)
    for i in range(3):
        st["current_term"][i] = 3
        st["role"][i] = abi.ROLE_FOLLOWER
        abi.set_log(st, i, [(47, 2), (48, 2)], last_written=(44, 2), snapshot=(44, 2))
        st["commit_index"][i] = st["last_applied"][i] = 44
        st["pending_first"][i] = 45
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"], m["kind"], m["from"], m["term"] = 1, abi.MSG_AER, 0, 3
    m["a"], m["b"], m["c"] = 44, 2, 49
    m["n_entries"], m["n_run0"], m["run0_term"], m["run1_term"] = 1, 1, 3, 3
    cpu = oracle_lib.Oracle(1, 3)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(1, 3, ring_capacity=16, ring_slots=1) as gpu:
        gpu.set_state(0, st)
        do, ro = cpu.step(m)
        dg, rg = gpu.step(m)
        so = cpu.get_state()
        assert_same("write below a sparse range", dg, rg, gpu.get_state(), do, ro, so)
        assert int(do["flags"][0]) & abi.F_WROTE and not int(do["flags"][0]) & abi.F_APPLIED
        assert (int(do["reply_next_index"][0]), int(do["reply_last_index"][0])) == (45, 45)      # written range
        s1 = so[1]
        assert (int(s1["last_index"]), int(s1["last_term"]), int(s1["first_index"]), int(s1["n_runs"])) == (44, 2, 45, 0)
        assert int(s1["last_applied"]) == 44 and int(s1["commit_index"]) == 49
        # the next append_entries_rpc continues from the snapshot
        m2 = m.copy()
        m2["n_entries"], m2["n_run0"] = 2, 2
        do, ro = cpu.step(m2); dg, rg = gpu.step(m2)
        assert_same("append after the range was lost", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
        assert int(cpu.get_state()["last_index"][1]) == 46


def _reply_ok(server, peer, term, next_index, last_index):
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"], m["kind"], m["from"], m["term"], m["flags"] = server, abi.MSG_AER_REPLY, peer, term, abi.MF_SUCCESS
    m["a"], m["b"] = next_index, last_index
    return m[0]


@pytest.mark.parametrize("groups,seed", [(4200, 231), (64, 232)])
def test_quorum_term_gate_on_any_run_table(engine_mod, oracle_lib, groups, seed):
    """evaluate_quorum's Raft 5.4.2 gate `fetch_term(Agreed) == current_term` (src/ra_server.erl:3633-3646) on run tables
    Raft can produce and on ones it cannot (terms in any order, at, above and below current_term, the agreed index in
    any run, so the lookup goes through the mirrored runs and through the table walk): leaders take success replies
    for several ticks (4 200 groups: rgb_submit takes the class-dispatch kernel from 4 096 messages per round on -- its
    fast path and its general path; 64 groups: the kind-generic kernel),
    decisions and states equal the checker's.  Written for a shortcut that answered the gate without the walk from a
    derived "older runs are below current_term" flag (measured: no gain, removed); a mutant that
    trusted the flag blindly fails here."""
    rng = np.random.default_rng(seed)
    N = 5
    st = abi.empty_server_states(groups, N)
    for s in range(groups * N):
        n_runs = int(rng.integers(1, 7))
        monotone = rng.random() < 0.5
        terms, t = [], int(rng.integers(1, 4))
        for r in range(n_runs):
            if monotone:
                t += int(rng.integers(1, 3)) if r else 0
            else:
                t = int(rng.choice([x for x in range(1, 8) if not terms or x != terms[-1]]))
            terms.append(t)
        first = int(rng.integers(1, 30))
        entries, idx = [], first
        for t in terms:
            for _ in range(int(rng.integers(1, 6))):
                entries.append((idx, t)); idx += 1
        li = entries[-1][0]
        lwi = int(rng.integers(first, li + 1))
        abi.set_log(st, s, entries, last_written=(lwi, dict(entries)[lwi]))
        st["pending_first"][s] = lwi + 1
        ct = max(terms) if monotone and rng.random() < 0.7 else int(rng.integers(1, 9))
        st["current_term"][s] = ct
        la = int(rng.integers(first - 1, li + 1))
        st["last_applied"][s] = la
        st["commit_index"][s] = min(li, la + int(rng.integers(0, 4)))
        st["role"][s] = abi.ROLE_LEADER if s % N == 0 else abi.ROLE_FOLLOWER
        st["leader_id"][s] = 0
        st["voted_for"][s] = 0
        st["present_mask"][s] = st["voter_mask"][s] = 0x1F
        st["status_mask"][s] = 0xFF
        for j in range(N):
            mi = int(rng.integers(max(first - 3, 0), li + 1))
            st["match_index"][s, j] = mi
            st["next_index"][s, j] = mi + 1 + int(rng.integers(0, 3))
            st["commit_index_sent"][s, j] = st["commit_index"][s]
    cpu = oracle_lib.Oracle(groups, N)
    cpu.set_state(0, st)
    moved = 0
    with engine_mod.RaGpuBatch(groups, N, ring_capacity=8192, ring_slots=2) as gpu:
        gpu.set_state(0, st)
        for tick in range(6):
            cur = cpu.get_state()
            msgs = []
            for g in range(groups):
                s = g * N
                li, first = int(cur["last_index"][s]), int(cur["first_index"][s])
                last = int(rng.integers(first, li + 1))
                msgs.append(_reply_ok(s, int(rng.integers(1, N)), int(cur["current_term"][s]), last + 1, last))
            msgs = np.array(msgs, dtype=abi.MSG_DTYPE)
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"term gate tick {tick}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            moved += int(np.count_nonzero(cpu.get_state()["commit_index"][::N] != cur["commit_index"][::N]))
    assert moved > groups // 4          # the gate opened often enough to matter (up AND down: no max())


@pytest.mark.parametrize("n_members,seed,groups,max_runs", [(5, 211, 300, 4), (3, 212, 300, 3), (7, 213, 200, 5)])
def test_bounded_run_table_matches_oracle(engine_mod, oracle_lib, n_members, seed, groups, max_runs):
    """A run table smaller than the logs' term structure: the engine forgets its oldest runs
    (RGB_F_RUNS_OVERFLOW, first_index raised); the checker models the same bound, so decisions and
    states stay bit-identical through overflows."""
    rng = np.random.default_rng(seed)
    st = fuzz.random_states(rng, groups, n_members, max_runs=max_runs + 2)     # <= max_runs - 1 runs each
    assert int(st["n_runs"].max()) <= max_runs
    cpu = oracle_lib.Oracle(groups, n_members, max_runs=max_runs)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(groups, n_members, ring_capacity=4096, ring_slots=2, max_runs=max_runs) as gpu:
        gpu.set_state(0, st)
        overflows = 0
        for tick in range(12):
            msgs = fuzz.random_msgs(rng, cpu.get_state(), n_members)
            do, ro = cpu.step(msgs)
            dg, rg = gpu.step(msgs)
            assert_same(f"bounded runs N={n_members} tick {tick}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            overflows += int(((do["flags"] & abi.F_RUNS_OVERFLOW) != 0).sum())
        assert overflows > 0, "the fuzz never overflowed the run table"


@pytest.mark.parametrize("n_members,groups,ticks", [(5, 2048, 48), (7, 1024, 32), (3, 1024, 32)])
def test_device_generated_stream_matches_oracle(engine_mod, oracle_lib, n_members, groups, ticks):
    """The bench's tick stream (device-side load generator incl. term churn and on-device
    elections) replayed through the oracle: every decision and the final state are identical."""
    import torch
    from ra_amd import workload as W
    G, N = groups, n_members
    S = G * N
    st0 = W.initial_states(G, N, 0x5EED0003)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st0)
    stream = torch.cuda.Stream()
    with engine_mod.RaGpuBatch(G, N, max_runs=16) as gpu:
        gpu.set_state(0, st0)
        dm = torch.zeros(ticks * S * 64, dtype=torch.uint8, device="cuda")
        dd = torch.zeros(ticks * S * 64, dtype=torch.uint8, device="cuda")
        dr = torch.zeros(S * max(N - 1, 1) * 56, dtype=torch.uint8, device="cuda")
        kc = torch.zeros(ticks * abi.N_KINDS, dtype=torch.int32, device="cuda")
        dn = torch.zeros(ticks, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for t in range(ticks):
                gpu.synth_tick_device(0x5EED0003, t, dm.data_ptr() + t * S * 64, kc.data_ptr() + t * 4 * abi.N_KINDS,
                                      dn.data_ptr() + t * 4, stream.cuda_stream)
                if t % 2 == 0:      # both ways of applying a device-generated tick
                    gpu.synth_apply_tick_device(dm.data_ptr() + t * S * 64, S, dd.data_ptr() + t * S * 64,
                                                dr.data_ptr(), stream.cuda_stream)
                else:
                    gpu.run_ticks_device(dm.data_ptr() + t * S * 64, S, 1, dd.data_ptr() + t * S * 64,
                                         dr.data_ptr(), stream.cuda_stream, d_tick_counts=dn.data_ptr() + t * 4)
        torch.cuda.synchronize()
        msgs = dm.cpu().numpy().view(abi.MSG_DTYPE).reshape(ticks, S)
        decs = abi.expand_decisions(dd.cpu().numpy().view(abi.DECISION_DTYPE)).reshape(ticks, S)
        counts = kc.cpu().numpy().reshape(ticks, abi.N_KINDS)
        ns = dn.cpu().numpy()
        flags_seen = 0
        for t in range(ticks):
            nt = int(ns[t])
            m = msgs[t, :nt]
            assert nt > G and not np.any(m["kind"] == abi.MSG_NOP)
            assert len(np.unique(m["server"])) == nt, "two messages for one server in a tick"
            assert np.array_equal(np.bincount(m["kind"], minlength=abi.N_KINDS), counts[t])
            # bucket order = (class, group mod 8, success flag): classes contiguous, and so is every (class, shard)
            bk = engine_mod.train_bucket(m["kind"], m["flags"], m["server"], N) >> 1
            assert np.all(np.diff(bk.astype(np.int64)) >= 0), "tick is not in (class, shard) order"
            assert np.all(np.diff(abi.family(m) // 2) >= 0), "classes are not contiguous"
            want, _ = cpu.step(m)
            got = decs[t, :nt]
            if got.tobytes() != want.tobytes():
                bad = int(np.flatnonzero((got.view(np.uint8).reshape(nt, 64) !=
                                          want.view(np.uint8).reshape(nt, 64)).any(axis=1))[0])
                raise AssertionError(f"tick {t} slot {bad}: msg={m[bad]}\n gpu={got[bad]}\n cpu={want[bad]}")
            flags_seen |= int(np.bitwise_or.reduce(want["flags"]))
            assert not np.any(want["flags"] & abi.F_INVARIANT), "the generator produced an invariant breach"
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes()
        # term churn really exercises the election path on the device
        for f in ("BECAME_LEADER", "SEND_VOTE_REQUESTS", "PRE_VOTE_REQS", "ROLE_CHANGED", "WROTE", "APPLIED",
                  "PIPELINE", "PERSIST", "REPROCESSED") if ticks >= 32 else ():
            assert flags_seen & VR.FLAG[f], f"stream never produced {f}"
        # and the groups stay healthy: most of them have exactly one leader at the end
        rows = gpu.snapshot()
        assert (rows["n_leaders"] == 1).mean() > 0.6


def test_config2_small_batches_through_the_host_path(engine_mod, oracle_lib):
    """BASELINE configs[1] as specified: 4 096 groups x 5 members, 256 messages per tick (50 % follower
    append_entries, 50 % leader success replies, seed 0x5EED0002) through rgb_submit / rgb_collect --
    the path the NIF binds; every decision, every rpc and the full final state equal the oracle's."""
    from ra_amd import workload as W
    G, N, seed = 4096, 5, 0x5EED0002
    st = W.initial_states(G, N, seed)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_capacity=1024, ring_slots=4) as gpu:
        gpu.set_state(0, st)
        n_dec = 0
        for t in range(200):
            m = W.gen_tick(cpu.get_state(), N, t, seed, W.MIX_CONFIG2, groups_per_tick=256, housekeeping=False)
            assert 0 < len(m) <= 256
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            assert dg.tobytes() == do.tobytes(), f"tick {t}: decisions differ"
            assert fuzz.sort_rpcs(rg).tobytes() == fuzz.sort_rpcs(ro).tobytes(), f"tick {t}: rpcs differ"
            n_dec += len(m)
        assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "final state differs"
        assert gpu.state_checksum() == engine_mod.combine_checksums(oracle_lib.server_checksums(cpu.get_state()))
        assert n_dec > 200 * 200


def test_config5_log_matching_repair_matches_oracle(engine_mod, oracle_lib):
    """BASELINE config 5 shape at a size the oracle handles in seconds: 7 members, 1024-entry
    uncommitted backlogs crossing 3-6 term boundaries, AERs with prev_log_index inside the backlog
    and a wrong prev_log_term half of the time, failed replies driving the leader's a8 repair."""
    from ra_amd import workload as W
    G, N, seed = 1024, 7, 0x5EED0005
    st = W.initial_states(G, N, seed, backlog=1024, boundaries=(3, 6))
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    with engine_mod.RaGpuBatch(G, N, max_runs=16, ring_capacity=G * N, ring_slots=2) as gpu:
        gpu.set_state(0, st)
        seen = 0
        for t in range(10):
            cur = cpu.get_state()
            m = W.gen_tick(cur, N, t, seed, W.MIX_CONFIG5, backlog_mode=True)
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            assert_same(f"config5 tick {t}", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            seen |= int(np.bitwise_or.reduce(do["flags"]))
        assert (cpu.get_state()["role"] == abi.ROLE_AWAIT_CONDITION).sum() > G // 4   # mismatches happened
        assert seen & abi.F_REPLY and seen & abi.F_PIPELINE


def test_full_size_properties_config3(engine_mod):
    """BASELINE config 3 at full size (65 536 groups x 5 members), no oracle in the loop:
    (1) determinism: two engines fed the same device-generated stream end on the same checksum of
    checksums; (2) batch-split invariance: a tick applied as ONE batch (class kernel) and the same
    tick applied through the host path in four chunks (sub-batches, generic/class kernels, different
    order of servers) give the same state; (3) every decision of a tick names a distinct server."""
    import torch
    from ra_amd import workload as W
    G, N, seed, ticks = 65536, 5, 0x5EED0003, 6
    S = G * N
    st0 = W.initial_states(G, N, seed)
    stream = torch.cuda.Stream()
    a = engine_mod.RaGpuBatch(G, N, max_runs=16)
    b = engine_mod.RaGpuBatch(G, N, max_runs=16, ring_capacity=65536, ring_slots=2)
    try:
        a.set_state(0, st0)
        b.set_state(0, st0)
        dm = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dd = torch.zeros(S * 64, dtype=torch.uint8, device="cuda")
        dn = torch.zeros(1, dtype=torch.int32, device="cuda")
        for t in range(ticks):
            with torch.cuda.stream(stream):
                a.synth_tick_device(seed, t, dm.data_ptr(), 0, dn.data_ptr(), stream.cuda_stream)
                a.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            n = int(dn.item())
            msgs = dm[:n * 64].cpu().numpy().view(abi.MSG_DTYPE)
            dec_a = abi.expand_decisions(dd[:n * 64].cpu().numpy().view(abi.DECISION_DTYPE))
            assert len(np.unique(msgs["server"])) == n
            assert not np.any(dec_a["flags"] & abi.F_INVARIANT)
            # the same messages, shuffled, through the host path in 65536-message chunks
            perm = np.random.default_rng(t).permutation(n)
            dec_b, _ = b.step(msgs[perm])
            assert dec_b.tobytes() == dec_a[perm].tobytes(), f"tick {t}: host path decisions differ"
            assert a.state_checksum() == b.state_checksum(), f"tick {t}: states differ"
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_sparse_pending_and_two_range_written_events(engine_mod, oracle_lib, seed):
    """SURVEY 8(f) #2 on the device: `pending` with gaps (the host's re-upload after ra_log:write_sparse/3 +
    install_snapshot, src/ra_log.erl:601-635) and written events whose ra_seq has two ranges (RGB_MF_SEQ2), in lock
    step with the checker AND with the literal ra_seq model of tests/ra_log_model.py: the same last_written, the same
    pending set, RGB_F_RESEND_PENDING exactly where the reference calls resend_pending/2 (:917-919), the
    {ok, Pend} = ra_seq:remove_prefix(..) badmatch of the snapshot clause (:929) as an invariant."""
    import test_pending_model as PM
    cpu = oracle_lib.Oracle(1, 3)
    with engine_mod.RaGpuBatch(1, 3, ring_capacity=64, ring_slots=2, max_runs=16) as gpu:
        def step(m):
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            assert_same("sparse pending", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            return do

        def set_state(st):
            cpu.set_state(0, st)
            gpu.set_state(0, st)
        PM.sparse_history(step, cpu.get_state, set_state, seed, steps=120)
        assert gpu.state_checksum() == engine_mod.combine_checksums(oracle_lib.server_checksums(cpu.get_state()))
    cpu.close()


def test_written_event_for_a_live_index_below_the_snapshot(engine_mod, oracle_lib):
    """test/ra_log_2_SUITE.erl:1840-1880 snapshot_installation_with_live_indexes, the cursor part: indexes 1..9
    written, live index 14 written sparsely (write_sparse({14,2,_}, 9, _)), snapshot {15,2} installed -- the host
    re-uploads the server with the range emptied and 14 still pending -- then the suite asserts last_written = {15,_},
    writes index 16 and waits for last_written = {16,2}: the late written event of 14 only trims `pending`
    (handle_event's snapshot clause, src/ra_log.erl:921-930), the one of 16 moves last_written."""
    import test_pending_model as PM
    cpu = oracle_lib.Oracle(1, 3)
    with engine_mod.RaGpuBatch(1, 3, ring_capacity=64, ring_slots=2, max_runs=16) as gpu:
        st = cpu.get_state()
        st[1] = PM.sparse_state(st[1:2], 15, 2, [14])[0]
        cpu.set_state(0, st); gpu.set_state(0, st)
        assert PM.pending_of(gpu.get_state()[1]) == [14]

        def step(m):
            do, ro = cpu.step(m)
            dg, rg = gpu.step(m)
            assert_same("live index", dg, rg, gpu.get_state(), do, ro, cpu.get_state())
            return gpu.get_state()[1]
        s = step(PM._msg(abi.MSG_AER, frm=0, term=2, a=15, b=2, c=15, n_entries=1, n_run0=1, run0_term=2))   # write 16
        assert int(s["last_index"]) == 16 and PM.pending_of(s) == [14, 16]
        assert (int(s["last_written_index"]), int(s["last_written_term"])) == (15, 2)
        s = step(PM._msg(abi.MSG_WRITTEN, term=2, a=14, b=14))                   # the WAL confirms the live index
        assert PM.pending_of(s) == [16] and int(s["n_pending_old"]) == 0
        assert (int(s["last_written_index"]), int(s["last_written_term"])) == (15, 2)
        s = step(PM._msg(abi.MSG_WRITTEN, term=2, a=16, b=16))
        assert PM.pending_of(s) == [] and (int(s["last_written_index"]), int(s["last_written_term"])) == (16, 2)
    cpu.close()


def test_forced_spill_class_kernel_is_bit_exact():
    """The per-tick class kernel compiled against a 96-register budget -- FORCED spills: ~1 500 spill instructions, 196
    bytes of scratch per lane -- must return the oracle's decisions on the device (round 5 saw a spilling build of this
    kernel return wrong decisions in lanes 0-15; round 6 could not make it happen again: tests/test_kernel_resources.py).
    The variant library is built here (one group size: about a minute) and run in its own process (one library per
    process): three ticks of the generator's stream over 16 384 five-member groups, every decision against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "ra_amd", "csrc", "variants", "spill5.so")
    srcs = [os.path.join(root, "ra_amd", "csrc", f) for f in ("rgb_kernels.hip", "rgb_api.hip", "rgb_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        r = subprocess.run(["bash", os.path.join(root, "tools", "build_variants.sh"), "spill5:-DRGB_CLASS_MIN_WAVES(N)=5"],
                           capture_output=True, text=True, env=dict(os.environ, ONLY_N="5"))
        assert os.path.exists(so) and "built spill5" in r.stdout, r.stdout + r.stderr
    env = dict(os.environ, RGB_LIB=so, G="16384", T="3")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "parity_tick0.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("tick ")]
    assert len(lines) == 3 and all(ln.endswith(" 0 mismatches") for ln in lines), r.stdout[-2000:]
