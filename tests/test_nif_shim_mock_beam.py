"""The Erlang NIF shim (ra_amd/csrc/ra_gpu_batch_nif.c) EXECUTED without OTP: compiled against a functional mock
of the erl_nif subset it uses (tests/native/mock_beam: heap terms, owned binaries, reference-counted resources
with destructors, enif_send into a queue, pthreads) and linked to the CPU-emulated library.  The test plays the
Erlang caller of erlang/ra_gpu_batch.erl -- open/4, register_groups/3, upload_state/3, submit/3, collect/1,
download_state/3, snapshot/2, start_collector/2, the WAL calls -- and checks what comes back, record for record,
against the checker.  What it cannot show is the BEAM itself (scheduling, real binaries' reference counts)."""
import ctypes as C
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

import fuzz
from ra_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T_ATOM, T_INT, T_BIN, T_TUPLE, T_RES, T_PID, T_BADARG = 1, 2, 3, 4, 5, 6, 7


@pytest.fixture(scope="module")
def beam(emulated_kernels_so, tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    out = tmp_path_factory.mktemp("nif") / "ra_gpu_batch_nif_mock.so"
    emu_dir, emu_name = os.path.split(emulated_kernels_so)
    cmd = ["gcc", "-std=c11", "-D_GNU_SOURCE", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "tests", "native", "mock_beam"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "ra_amd", "csrc", "ra_gpu_batch_nif.c"),
           os.path.join(ROOT, "tests", "native", "mock_beam", "mock_beam.c"),
           "-L", emu_dir, "-l:" + emu_name, "-Wl,-rpath," + emu_dir, "-lpthread", "-o", str(out)]
    san = os.environ.get("RGB_EMU_SANITIZE")
    if san:                                   # same opt-in as the emulation build (tests/conftest.py): one runtime
        clang = shutil.which("clang", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang")
        cmd[0] = clang
        cmd[1:1] = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-shared-libsan"] + \
                   (["-fno-sanitize-recover=undefined"] if san == "undefined" else [])
        # RGB_EMU_SANITIZE=thread: the concurrent-producer test below under ThreadSanitizer, run as
        #   RGB_EMU_SANITIZE=thread LD_PRELOAD=<clang lib dir>/libclang_rt.tsan-x86_64.so \
        #       TSAN_OPTIONS=halt_on_error=1 pytest tests/test_nif_shim_mock_beam.py -k "concurrent or fans"
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    L = C.CDLL(str(out))
    vp, u64 = C.c_void_p, C.c_uint64
    for name, res, args in [("mock_load", C.c_int, []), ("mock_call", vp, [C.c_char_p, C.c_int, C.POINTER(vp)]),
                            ("mock_func_flags", C.c_uint, [C.c_char_p, C.c_int]),
                            ("mock_uint", vp, [u64]), ("mock_int", vp, [C.c_int]), ("mock_atom", vp, [C.c_char_p]),
                            ("mock_pid", vp, [u64]), ("mock_binary", vp, [vp, C.c_size_t]), ("mock_tag", C.c_int, [vp]),
                            ("mock_atom_name", C.c_char_p, [vp]), ("mock_arity", C.c_int, [vp]),
                            ("mock_elem", vp, [vp, C.c_int]), ("mock_uint_value", u64, [vp]),
                            ("mock_int_value", C.c_int64, [vp]), ("mock_bin_data", vp, [vp]),
                            ("mock_bin_size", C.c_size_t, [vp]), ("mock_gc_resource_term", None, [vp]),
                            ("mock_live_resources", C.c_long, []), ("mock_dtor_calls", C.c_long, []),
                            ("mock_recv", vp, [C.c_int, C.POINTER(u64)]), ("mock_dirty_reschedules", C.c_long, [])]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    assert L.mock_load() == 0, "on_load: resource type or ABI version"
    return Beam(L)


class Beam:
    """Term construction / inspection on the Python side of the mock."""

    def __init__(self, L):
        self.L = L

    def to_term(self, x):
        if isinstance(x, Opaque):
            return x.t
        if isinstance(x, (bytes, bytearray)):
            buf = (C.c_char * max(len(x), 1)).from_buffer_copy(bytes(x) or b"\0")
            return self.L.mock_binary(C.cast(buf, C.c_void_p), len(x))
        if isinstance(x, str):
            return self.L.mock_atom(x.encode())
        if isinstance(x, int):
            return self.L.mock_int(x) if x < 0 else self.L.mock_uint(x)
        raise TypeError(x)

    def from_term(self, t):
        tag = self.L.mock_tag(t)
        if tag == T_ATOM:
            return self.L.mock_atom_name(t).decode()
        if tag == T_INT:
            return int(self.L.mock_uint_value(t))
        if tag == T_BIN:
            return C.string_at(self.L.mock_bin_data(t), self.L.mock_bin_size(t))
        if tag == T_TUPLE:
            return tuple(self.from_term(self.L.mock_elem(t, k)) for k in range(self.L.mock_arity(t)))
        if tag == T_BADARG:
            return "badarg"
        if tag in (T_RES, T_PID):
            return Opaque(t)
        raise AssertionError(f"term tag {tag}")

    def call(self, name, *args):
        argv = (C.c_void_p * max(len(args), 1))(*[self.to_term(a) for a in args])
        t = self.L.mock_call(name.encode(), len(args), argv)
        assert t, f"undef: {name}/{len(args)}"
        return self.from_term(t)

    def recv(self, timeout_ms=20000):
        to = C.c_uint64(0)
        t = self.L.mock_recv(timeout_ms, C.byref(to))
        return (None, None) if not t else (int(to.value), self.from_term(t))


class Opaque:
    def __init__(self, t):
        self.t = t


def test_shim_round_trip_against_the_checker(beam, oracle_lib):
    """open -> register_groups -> upload_state -> (submit, collect) x ticks -> download_state / snapshot, every
    record compared with the checker; then the destructor path."""
    G, N = 96, 5
    rng = np.random.default_rng(77)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    live0, dtors0 = beam.L.mock_live_resources(), beam.L.mock_dtor_calls()

    ok, ctx = beam.call("open", 0, 16, 2, 1024)
    assert ok == "ok" and isinstance(ctx, Opaque)
    assert beam.L.mock_live_resources() == live0 + 1
    assert beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    # malformed arguments are badarg, not crashes (badarg only for malformed binaries)
    assert beam.call("upload_state", ctx, 0, st.tobytes()[:-1]) == "badarg"
    assert beam.call("submit", ctx, b"\0" * 63, 1) == "badarg"
    assert beam.call("submit", 17, b"", 1) == "badarg"
    assert beam.call("collect", ctx) == ("error", "empty")

    for tick in range(1, 5):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), N)
        want_d, want_r = cpu.step(msgs)
        assert beam.call("submit", ctx, msgs.tobytes(), tick) == "ok"
        ok, got_tick, n, dec_bin, rpc_bin = beam.call("collect", ctx)
        assert (ok, got_tick, n) == ("ok", tick, len(msgs))
        assert len(dec_bin) == n * abi.DECISION_DTYPE.itemsize          # binaries carry exactly the records
        assert dec_bin == want_d.tobytes(), f"tick {tick}: decisions differ"
        got_r = np.frombuffer(rpc_bin, dtype=abi.RPC_DTYPE)
        assert len(got_r) == len(want_r)
        assert fuzz.sort_rpcs(got_r.copy()).tobytes() == fuzz.sort_rpcs(want_r).tobytes(), f"tick {tick}: rpcs"

    ok, state_bin = beam.call("download_state", ctx, 0, G * N)
    assert ok == "ok" and state_bin == cpu.get_state().tobytes()
    ok, lb = beam.call("snapshot", ctx, G)
    assert ok == "ok" and len(lb) == G * 32
    # the binary is sized from the registration: a caller's smaller (or larger) NGroups is badarg, never an overrun
    assert beam.call("snapshot", ctx, G - 1) == "badarg"
    assert beam.call("snapshot", ctx, G + 1) == "badarg"
    # an error code from the library comes back as {error, Atom}
    assert beam.call("download_state", ctx, G * N, 8) == ("error", "invalid")
    # the node's leaderboard all-gather through the shim (one rank here): the gathered binary = this context's rows,
    # padded to NRows; before comm_init/4 and with too few rows it is badarg
    assert beam.call("allgather_leaderboard", ctx, G) == "badarg"
    ok, comm_id = beam.call("comm_unique_id")
    assert ok == "ok" and len(comm_id) == abi.COMM_ID_BYTES
    assert beam.call("comm_init", ctx, comm_id[:-1], 1, 0) == "badarg"
    assert beam.call("comm_init", ctx, comm_id, 1, 0) == "ok"
    assert beam.call("comm_init", ctx, comm_id, 1, 0) == "badarg"          # once per context
    assert beam.call("allgather_leaderboard", ctx, G - 1) == "badarg"
    ok, gathered = beam.call("allgather_leaderboard", ctx, G + 3)
    assert ok == "ok" and len(gathered) == (G + 3) * 32
    assert gathered[:G * 32] == lb and gathered[G * 32:] == b"\0" * (3 * 32)

    # the last reference goes away: the destructor runs rgb_close exactly once
    beam.L.mock_gc_resource_term(ctx.t)
    assert beam.L.mock_live_resources() == live0 and beam.L.mock_dtor_calls() == dtors0 + 1
    cpu.close()


def test_collector_thread_fans_batches_back_in_order(beam, oracle_lib):
    """start_collector/2: a thread owns rgb_collect and enif_send()s {ra_gpu_batch, Tick, N, Decisions, Rpcs}
    to the owner; submit/3 stays non-blocking.  The resource stays alive while the thread holds it."""
    G, N = 64, 3
    rng = np.random.default_rng(78)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    dtors0 = beam.L.mock_dtor_calls()
    ok, ctx = beam.call("open", 0, 16, 4, 512)
    assert ok == "ok"
    assert beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    owner = Opaque(beam.L.mock_pid(4242))
    assert beam.call("start_collector", ctx, 99) == "badarg"           # not a pid
    assert beam.call("start_collector", ctx, owner) == "ok"
    assert beam.call("start_collector", ctx, owner) == "badarg"        # already running
    assert beam.call("collect", ctx) == ("error", "collector_running")  # one consumer at a time
    wants = []
    for tick in range(10, 16):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), N)
        wants.append((tick, len(msgs), cpu.step(msgs)[0].tobytes()))
        assert beam.call("submit", ctx, msgs.tobytes(), tick) == "ok"
        to, msg = beam.recv()                      # one batch in flight at a time keeps this test deterministic
        assert msg is not None, "collector sent nothing"
        tag, got_tick, n, dec_bin, _rpc_bin = msg
        assert to == 4242 and tag == "ra_gpu_batch"
        assert (got_tick, n, dec_bin) == wants[-1]
    # several batches in flight (ring_slots = 4): submit/3 on this thread races the collector's rgb_collect on
    # its own; the batches still come back whole and in submission order
    burst = []
    for tick in range(20, 23):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), N)
        burst.append((tick, len(msgs), cpu.step(msgs)[0].tobytes()))
        assert beam.call("submit", ctx, msgs.tobytes(), tick) == "ok"
    for want in burst:
        to, msg = beam.recv()
        assert msg is not None and to == 4242
        assert (msg[1], msg[2], msg[3]) == want
    # the collector thread holds its own reference: stop_collector/1 joins it, and only then does dropping the
    # term run the destructor
    assert beam.call("stop_collector", ctx) == "ok"
    assert beam.call("stop_collector", ctx) == "badarg"
    assert beam.L.mock_dtor_calls() == dtors0
    beam.L.mock_gc_resource_term(ctx.t)
    assert beam.L.mock_dtor_calls() == dtors0 + 1
    cpu.close()


def test_collector_fans_decisions_back_to_the_owning_processes(beam, oracle_lib):
    """register_owner/4: every gen_statem gets ONE message per batch holding only its servers' decisions, in
    submission order, with their rpc records re-indexed; unregistered servers go to the default owner
    (reference interception point: per process, src/ra_server_proc.erl:1356-1397)."""
    G, N = 40, 5
    rng = np.random.default_rng(79)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    ok, ctx = beam.call("open", 0, 16, 4, 512)
    assert ok == "ok"
    assert beam.call("register_owner", ctx, 0, 1, Opaque(beam.L.mock_pid(1))) == "badarg"   # before register_groups
    assert beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    # pid 1000+g owns the five servers of group g for g < 30; the last ten groups stay with the default owner
    for g in range(30):
        assert beam.call("register_owner", ctx, g * N, N, Opaque(beam.L.mock_pid(1000 + g))) == "ok"
    assert beam.call("register_owner", ctx, G * N - 1, 2, Opaque(beam.L.mock_pid(7))) == "badarg"   # out of range
    assert beam.call("start_collector", ctx, Opaque(beam.L.mock_pid(4242))) == "ok"
    for tick in range(1, 4):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), N)
        want_d, want_r = cpu.step(msgs)
        assert beam.call("submit", ctx, msgs.tobytes(), tick) == "ok"
        owner_of = lambda srv: 1000 + srv // N if srv // N < 30 else 4242
        owners = []
        for d in want_d:                                       # first-appearance order
            if owner_of(int(d["server"])) not in owners:
                owners.append(owner_of(int(d["server"])))
        got = {}
        for _ in owners:
            to, msg = beam.recv()
            assert msg is not None, "an owner got nothing"
            assert msg[0] == "ra_gpu_batch" and msg[1] == tick and to not in got
            got[to] = msg
        assert sorted(got) == sorted(owners)
        for o in owners:
            idx = [i for i, d in enumerate(want_d) if owner_of(int(d["server"])) == o]
            _tag, _t, n, dec_bin, rpc_bin = got[o]
            assert n == len(idx) and dec_bin == want_d[idx].tobytes(), f"owner {o}: decisions"
            rp = np.frombuffer(rpc_bin, dtype=abi.RPC_DTYPE)
            exp = []
            for pos, i in enumerate(idx):
                for r in want_r[want_r["msg_index"] == i]:
                    r = r.copy(); r["msg_index"] = pos; exp.append(r)
            exp = np.array(exp, dtype=abi.RPC_DTYPE) if exp else np.zeros(0, dtype=abi.RPC_DTYPE)
            assert fuzz.sort_rpcs(rp.copy()).tobytes() == fuzz.sort_rpcs(exp).tobytes(), f"owner {o}: rpcs"
    assert beam.recv(timeout_ms=200) == (None, None)           # nothing else was sent
    assert beam.call("stop_collector", ctx) == "ok"
    beam.L.mock_gc_resource_term(ctx.t)
    cpu.close()


def test_owner_table_is_bounded_and_the_collector_starts_once(beam, oracle_lib):
    """(1) a pid that registers again keeps its slot, unregister_owner/2 frees a slot for the next process: a thousand
    restarts leave the table as big as the live owners; the servers of an unregistered process go to the default
    owner.  (2) start_collector/2 from many threads at once: exactly one starts a thread (compare-exchange)."""
    import threading
    G, N = 16, 5
    rng = np.random.default_rng(83)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    ok, ctx = beam.call("open", 0, 16, 4, 512)
    assert ok == "ok" and beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    for g in range(G):
        assert beam.call("register_owner", ctx, g * N, N, Opaque(beam.L.mock_pid(2000 + g))) == "ok"
    assert beam.call("owner_slots", ctx) == (G, G)
    for g in range(G):                                          # the same processes again: no growth
        assert beam.call("register_owner", ctx, g * N, N, Opaque(beam.L.mock_pid(2000 + g))) == "ok"
    assert beam.call("owner_slots", ctx) == (G, G)
    for restart in range(1000):                                 # group 3's process dies and comes back under a new pid
        assert beam.call("unregister_owner", ctx, Opaque(beam.L.mock_pid(2003 if restart == 0 else 50000 + restart - 1))) == "ok"
        assert beam.call("register_owner", ctx, 3 * N, N, Opaque(beam.L.mock_pid(50000 + restart))) == "ok"
    assert beam.call("owner_slots", ctx) == (G, G)
    assert beam.call("unregister_owner", ctx, Opaque(beam.L.mock_pid(2005))) == "ok"     # group 5: nobody takes over
    assert beam.call("owner_slots", ctx) == (G, G - 1)
    results = []
    def start():
        results.append(beam.call("start_collector", ctx, Opaque(beam.L.mock_pid(4242))))
    ths = [threading.Thread(target=start) for _ in range(8)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert sorted(results) == ["badarg"] * 7 + ["ok"], results
    msgs = fuzz.random_msgs(rng, cpu.get_state(), N, frac=1.0)
    want_d, _ = cpu.step(msgs)
    assert beam.call("submit", ctx, msgs.tobytes(), 1) == "ok"
    owner_of = lambda srv: 4242 if srv // N == 5 else (50999 if srv // N == 3 else 2000 + srv // N)
    expect = {}
    for d in want_d:
        expect.setdefault(owner_of(int(d["server"])), []).append(d)
    got = {}
    for _ in expect:
        to, msg = beam.recv()
        assert msg is not None and msg[0] == "ra_gpu_batch"
        got[to] = msg
    assert sorted(got) == sorted(expect)
    for o, ds in expect.items():
        assert got[o][3] == np.array(ds, dtype=abi.DECISION_DTYPE).tobytes(), f"owner {o}"
        assert len(got[o][3]) == 64 * got[o][2]                 # the binary is exactly the batch's decisions
    assert beam.call("stop_collector", ctx) == "ok"
    assert beam.call("start_collector", ctx, Opaque(beam.L.mock_pid(4242))) == "ok"      # and again
    assert beam.call("stop_collector", ctx) == "ok"
    beam.L.mock_gc_resource_term(ctx.t)
    cpu.close()


@pytest.mark.skipif(not os.environ.get("RGB_FANBACK_REPORT"), reason="a 12-minute measurement: RGB_FANBACK_REPORT=1 pytest -s -k fan_back_rate")
@pytest.mark.parametrize("G,owners", [(8192, 1), (8192, 4096), (65536, 327680)])
def test_fan_back_rate_report(beam, G, owners, capsys):
    """What an Erlang caller pays for the per-process fan-back (reference interception point: one gen_statem per
    server, src/ra_server_proc.erl:1382-1397): wall time from submit/3 until every owner has its message, for ONE
    default owner (one send per batch), 4 096 owners and one owner PER SERVER (327 680: two binaries + one send per
    decision).  The kernels run on the CPU emulation here, so the single-owner time is the baseline that the others
    are read against; the numbers go to DESIGN.md section 5 (run with -s).  Asserts only that everything arrives."""
    import time
    N = 5
    S = G * N
    rng = np.random.default_rng(91)
    from ra_amd import workload as W
    st = W.initial_states(G, N, 91)
    ok, ctx = beam.call("open", 0, 16, 2, S)
    assert ok == "ok" and beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    per = max(S // owners, 1)
    n_own = 0
    if owners > 1:
        for k in range(0, S, per):
            assert beam.call("register_owner", ctx, k, min(per, S - k), Opaque(beam.L.mock_pid(10000 + k // per))) == "ok"
            n_own += 1
    assert beam.call("start_collector", ctx, Opaque(beam.L.mock_pid(4242))) == "ok"
    msgs = W.gen_tick(st, N, 0, 91)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        assert beam.call("submit", ctx, msgs.tobytes(), rep + 1) == "ok"
        want = len(np.unique(msgs["server"] // per)) if owners > 1 else 1
        got, n_dec = 0, 0
        while got < want:
            to, msg = beam.recv(timeout_ms=120000)
            assert msg is not None, f"{got} of {want} owner messages arrived"
            got += 1; n_dec += msg[2]
        dt = time.perf_counter() - t0
        assert n_dec == len(msgs)
        best = dt if best is None else min(best, dt)
    nb, nd, ns = beam.call("fan_back_stats", ctx)
    with capsys.disabled():
        print(f"\n[fan-back] in the collector thread: {nb} batches, {nd} decisions, {ns / max(nd, 1):.0f} ns per decision "
              f"({nd / max(ns, 1) * 1e3:.1f} M decisions/s) with {max(n_own, 1)} owners")
        print(f"[fan-back] {len(msgs)} decisions, {max(n_own, 1)} registered owners -> {want} messages per batch: "
              f"{best * 1e3:.1f} ms per batch (CPU-emulated kernels + collector + fan_back + mock enif_send + Python recv), "
              f"{len(msgs) / best / 1e6:.2f} M decisions/s")
    assert beam.call("stop_collector", ctx) == "ok"
    beam.L.mock_gc_resource_term(ctx.t)


def test_concurrent_producers_and_the_collector_thread(beam, oracle_lib):
    """Several processes submit at once (submit is thread-safe) while the collector thread consumes:
    every batch comes back exactly once and whole, per-producer order is kept, the engine ends in the state the
    checker reaches for the same batches in the order they were accepted.  (The sanitizer build of this module
    runs it under ThreadSanitizer.)"""
    import threading
    G, N, P, ROUNDS = 64, 3, 4, 12
    rng = np.random.default_rng(80)
    st = fuzz.random_states(rng, G, N, max_runs=6)
    cpu = oracle_lib.Oracle(G, N)
    cpu.set_state(0, st)
    ok, ctx = beam.call("open", 0, 16, 4, 512)
    assert ok == "ok"
    assert beam.call("register_groups", ctx, G, N) == "ok"
    assert beam.call("upload_state", ctx, 0, st.tobytes()) == "ok"
    assert beam.call("start_collector", ctx, Opaque(beam.L.mock_pid(4242))) == "ok"
    # producer p owns the groups g with g % P == p: batches of different producers touch disjoint servers, so the
    # final state does not depend on how their submits interleave
    per = [[] for _ in range(P)]
    states = cpu.get_state()
    for p in range(P):
        for r in range(ROUNDS):
            m = fuzz.random_msgs(rng, states, N)
            m = m[(m["server"] // N) % P == p]
            per[p].append(m)
    errors, full = [], [0]

    def producer(p):
        try:
            for r, m in enumerate(per[p]):
                while True:
                    res = beam.call("submit", ctx, m.tobytes(), p * 1000 + r)
                    if res == "ok":
                        break
                    assert res == ("error", "full"), res
                    full[0] += 1
        except Exception as e:                                  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=producer, args=(p,)) for p in range(P)]
    for t in threads:
        t.start()
    seen = []
    for _ in range(P * ROUNDS):
        to, msg = beam.recv(timeout_ms=60000)
        assert msg is not None, f"lost a batch after {len(seen)}"
        assert to == 4242 and msg[0] == "ra_gpu_batch"
        seen.append((int(msg[1]), int(msg[2]), msg[3]))
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(t for t, _, _ in seen) == sorted(p * 1000 + r for p in range(P) for r in range(ROUNDS))
    for p in range(P):                                          # per-producer FIFO
        ticks = [t for t, _, _ in seen if t // 1000 == p]
        assert ticks == sorted(ticks)
    # replay in acceptance order through the checker: same decisions batch by batch, same final state
    for tick, n, dec_bin in seen:
        m = per[tick // 1000][tick % 1000]
        want_d, _ = cpu.step(m)
        assert n == len(m) and dec_bin == want_d.tobytes(), f"batch {tick}"
    assert beam.call("stop_collector", ctx) == "ok"
    ok, state_bin = beam.call("download_state", ctx, 0, G * N)
    assert ok == "ok" and state_bin == cpu.get_state().tobytes()
    beam.L.mock_gc_resource_term(ctx.t)
    cpu.close()


def test_big_batches_go_to_a_dirty_scheduler(beam):
    """submit/3 above 2048 messages reschedules itself as a dirty CPU-bound NIF (enif_schedule_nif); small ones
    run inline."""
    G, N = 1024, 3
    ok, ctx = beam.call("open", 0, 16, 2, 4096)
    assert ok == "ok" and beam.call("register_groups", ctx, G, N) == "ok"
    m = np.zeros(64, dtype=abi.MSG_DTYPE)
    m["kind"], m["server"] = abi.MSG_AER_REPLY, np.arange(64)
    before = beam.L.mock_dirty_reschedules()
    assert beam.call("submit", ctx, m.tobytes(), 1) == "ok"
    assert beam.L.mock_dirty_reschedules() == before
    big = np.zeros(3000, dtype=abi.MSG_DTYPE)
    big["kind"], big["server"] = abi.MSG_AER_REPLY, np.arange(3000)
    assert beam.call("submit", ctx, big.tobytes(), 2) == "ok"
    assert beam.L.mock_dirty_reschedules() == before + 1
    for want in (64, 3000):
        ok, _t, n, _d, _r = beam.call("collect", ctx)
        assert (ok, n) == ("ok", want)
    beam.L.mock_gc_resource_term(ctx.t)


def test_route_nif_is_the_c_partition(beam):
    from ra_amd import shard
    ids = np.arange(300, dtype=np.uint64)
    for world in (2, 8):
        got = [beam.call("route", int(g), world) for g in ids]
        assert got == shard.owner(ids, world).tolist()
    assert beam.call("route", 5, 0) == "badarg"


def test_nif_table_matches_the_erlang_stub(beam):
    """Every NIF the Erlang module declares (erlang/ra_gpu_batch.erl: `Name(_Args) -> erlang:nif_error(not_loaded)`)
    is in the shim's table with the same arity, and the blocking ones are dirty-scheduler NIFs."""
    import re
    src = open(os.path.join(ROOT, "erlang", "ra_gpu_batch.erl")).read()
    stubs = re.findall(r"^(\w+)\(([^)]*)\)\s*->\s*erlang:nif_error\(not_loaded\)\.", src, flags=re.M)
    assert len(stubs) >= 11
    for name, args in stubs:
        arity = len([a for a in args.split(",") if a.strip()])
        flags = beam.L.mock_func_flags(name.encode(), arity)
        assert flags != 0xFFFFFFFF, f"{name}/{arity} is not in the NIF table"
        if name in ("submit", "open", "start_collector", "register_owner", "route", "owner_slots", "fan_back_stats"):
            assert flags == 0, f"{name} must not be a dirty NIF (non-blocking)"
        elif name == "unregister_owner":
            assert flags == 1, "unregister_owner scans the owner map: dirty CPU-bound"
        else:
            assert flags == 2, f"{name} waits on the GPU / copies: dirty IO-bound"


def test_wal_nifs(beam):
    """wal_checksums/3, wal_frame/4 and wal_recover_check/2 through the shim: checksums against zlib, the framed
    batch round-trips through the recovery check, a flipped byte in the middle is `corrupt`."""
    ok, ctx = beam.call("open", 0, 16, 2, 256)
    assert ok == "ok"
    rng = np.random.default_rng(5)
    n = 40
    payloads = [rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8).tobytes() for _ in range(n)]
    entries = np.zeros(n, dtype=abi.WAL_ENTRY_DTYPE)
    off = 0
    for i, p in enumerate(payloads):
        entries[i]["index"], entries[i]["term"] = 1000 + i, 7
        entries[i]["data_offset"], entries[i]["data_len"] = off, len(p)
        off += len(p)
    data = b"".join(payloads)
    ok, sums = beam.call("wal_checksums", ctx, entries.tobytes(), data)
    assert ok == "ok"
    got = np.frombuffer(sums, dtype="<u4")
    for i, p in enumerate(payloads):
        want = zlib.adler32((1000 + i).to_bytes(8, "big") + (7).to_bytes(8, "big") + p) & 0xFFFFFFFF
        assert int(got[i]) == want, f"entry {i}"
    assert beam.call("wal_checksums", ctx, entries.tobytes()[:-1], data) == "badarg"

    # wal_frame/4: the batch's on-disk bytes equal struct.pack + zlib, written independently
    import test_wal_framing as WF
    lens = [0, 1, 15, 16, 17, 1000] + [int(x) for x in rng.integers(0, 900, size=30)]
    specs = WF.random_specs(rng, len(lens), lens)
    recs, rdata, rpayloads = WF.make_batch(rng, specs)
    ok, framed = beam.call("wal_frame", ctx, recs.tobytes(), rdata.tobytes(), 0)
    assert ok == "ok" and framed == WF.python_frame(specs, rpayloads)
    assert beam.call("wal_frame", ctx, recs.tobytes()[:-3], rdata.tobytes(), 0) == "badarg"

    # wal_recover_check/2 over a file made of that batch: every record back, status clean; one flipped payload
    # byte in the middle is `corrupt` (wal_checksum_validation_failure), in the last record `dropped_last`
    file_bytes = abi.WAL_FILE_HEADER + framed
    ok, scanned_bin, n_ok, status = beam.call("wal_recover_check", ctx, file_bytes)
    scanned = np.frombuffer(scanned_bin, dtype=abi.WAL_SCANNED_DTYPE)
    assert (ok, n_ok, status) == ("ok", len(specs), "clean") and len(scanned) == len(specs)
    assert [(int(r["index"]), int(r["term"]), int(r["data_len"])) for r in scanned] == \
           [(idx, term, ln) for (_t, _w, _u, idx, term, ln) in specs]
    mid = next(i for i in range(len(specs) // 2, len(specs)) if specs[i][5] > 0)
    bad = bytearray(file_bytes)
    bad[int(scanned[mid]["data_offset"])] ^= 0x40
    ok, _sb, n_ok, status = beam.call("wal_recover_check", ctx, bytes(bad))
    assert (ok, n_ok, status) == ("ok", mid, "corrupt")
    last = len(specs) - 1
    assert specs[last][5] > 0
    bad = bytearray(file_bytes)
    bad[int(scanned[last]["data_offset"])] ^= 0x40
    ok, _sb, n_ok, status = beam.call("wal_recover_check", ctx, bytes(bad))
    assert (ok, n_ok, status) == ("ok", last, "dropped_last")
    beam.L.mock_gc_resource_term(ctx.t)
