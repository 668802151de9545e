"""The device transition code itself, without a GPU: ra_amd/csrc/rgb_kernels.hip compiled as x86 C++
(tests/native/kernel_on_cpu.cpp + tests/native/fake_hip) and run one lane per message -- the generic path
process_message<N, -1> and every clause-folded specialisation process_message<N, KIND> the class-dispatch
kernel instantiates, plus the pack/unpack kernels -- bit for bit against the checker on the same random
ticks, bounded run tables and closed-loop cluster streams the -m gpu tests use.  The kernels' launch
structure (LDS staging, cooperative fetch, class dispatch) only runs on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from ra_amd import abi, engine
import fuzz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu_lib(emulated_kernels_so):
    out = emulated_kernels_so
    L = C.CDLL(str(out))
    L.emu_new.restype = C.c_void_p
    L.emu_new.argtypes = [C.c_uint32] * 5
    L.emu_free.argtypes = [C.c_void_p]
    L.emu_set_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.emu_get_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.emu_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                           C.POINTER(C.c_uint32), C.c_int]
    return L


class Emu:
    """Same step/get_state/set_state interface as the engine and the checker."""

    def __init__(self, L, n_groups, n_members, max_runs=16, max_pipeline_count=0, max_aer_batch=0, specialised=False):
        self.L, self.S, self.N, self.spec = L, n_groups * n_members, n_members, int(specialised)
        self.h = L.emu_new(n_groups, n_members, max_runs, max_pipeline_count, max_aer_batch)

    def set_state(self, first, states):
        st = np.ascontiguousarray(states, dtype=abi.SERVER_STATE_DTYPE)
        self.L.emu_set_state(self.h, first, len(st), st.ctypes.data)

    def get_state(self, first=0, n=None):
        n = self.S - first if n is None else n
        out = np.zeros(n, dtype=abi.SERVER_STATE_DTYPE)
        self.L.emu_get_state(self.h, first, n, out.ctypes.data)
        return out

    def step(self, msgs):
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        dec = np.zeros(len(m), dtype=abi.DECISION_DTYPE)
        cap = max(1, len(m) * abi.MAX_MEMBERS)
        rpcs = np.zeros(cap, dtype=abi.RPC_DTYPE)
        n = C.c_uint32(0)
        assert self.L.emu_step(self.h, m.ctypes.data, len(m), dec.ctypes.data, rpcs.ctypes.data, cap, C.byref(n),
                               self.spec) == 0
        return dec, rpcs[:n.value]

    def close(self):
        self.L.emu_free(self.h)


def assert_same(tag, dg, rg, sg, do, ro, so):
    for name, a, b in (("decisions", dg, do), ("rpcs", fuzz.sort_rpcs(rg), fuzz.sort_rpcs(ro))):
        assert len(a) == len(b), f"{tag}: {name} count {len(a)} vs {len(b)}"
        bad = [i for i in range(len(a)) if a[i].tobytes() != b[i].tobytes()]
        assert not bad, f"{tag}: {name}[{bad[0]}] emulated kernel {a[bad[0]]} checker {b[bad[0]]}"
    bad = [i for i in range(len(sg)) if sg[i].tobytes() != so[i].tobytes()]
    if bad:
        i = bad[0]
        diff = [f for f in abi.SERVER_STATE_DTYPE.names if sg[i][f].tobytes() != so[i][f].tobytes()]
        raise AssertionError(f"{tag}: state of server {i} differs in {diff}: kernel {[sg[i][f] for f in diff]} "
                             f"checker {[so[i][f] for f in diff]}")


@pytest.mark.parametrize("specialised", [False, True], ids=["generic", "per_kind"])
@pytest.mark.parametrize("n_members,seed", [(1, 301), (2, 302), (3, 303), (5, 304), (7, 305), (8, 306)])
def test_kernel_code_equals_checker_on_random_ticks(emu_lib, oracle_lib, n_members, seed, specialised):
    rng = np.random.default_rng(seed)
    G = 300
    st = fuzz.random_states(rng, G, n_members, max_runs=6)
    cpu = oracle_lib.Oracle(G, n_members); cpu.set_state(0, st)
    emu = Emu(emu_lib, G, n_members, specialised=specialised); emu.set_state(0, st)
    assert_same("upload/download", [], np.zeros(0, dtype=abi.RPC_DTYPE), emu.get_state(), [], np.zeros(0, dtype=abi.RPC_DTYPE),
                cpu.get_state())
    seen = 0
    for tick in range(10):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), n_members)
        do, ro = cpu.step(msgs)
        dg, rg = emu.step(msgs)
        assert_same(f"N={n_members} tick {tick}", dg, rg, emu.get_state(), do, ro, cpu.get_state())
        seen |= int(np.bitwise_or.reduce(do["flags"]))
    emu.close()
    for f in (abi.F_REPLY, abi.F_WROTE, abi.F_TRUNCATED, abi.F_PIPELINE, abi.F_INVARIANT, abi.F_REPROCESSED, abi.F_APPLIED):
        assert seen & f, hex(f)


@pytest.mark.parametrize("n_members,seed,max_runs", [(5, 311, 4), (3, 312, 3)])
def test_kernel_code_with_bounded_run_tables(emu_lib, oracle_lib, n_members, seed, max_runs):
    rng = np.random.default_rng(seed)
    G = 300
    st = fuzz.random_states(rng, G, n_members, max_runs=max_runs + 2)
    cpu = oracle_lib.Oracle(G, n_members, max_runs=max_runs); cpu.set_state(0, st)
    emu = Emu(emu_lib, G, n_members, max_runs=max_runs, specialised=True); emu.set_state(0, st)
    overflows = 0
    for tick in range(12):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), n_members)
        do, ro = cpu.step(msgs)
        dg, rg = emu.step(msgs)
        assert_same(f"bounded runs tick {tick}", dg, rg, emu.get_state(), do, ro, cpu.get_state())
        overflows += int(((do["flags"] & abi.F_RUNS_OVERFLOW) != 0).sum())
    emu.close()
    assert overflows > 0


@pytest.mark.parametrize("n_members,seed,snapshots", [(3, 41, False), (5, 42, True)])
def test_kernel_code_on_closed_loop_cluster_streams(emu_lib, oracle_lib, n_members, seed, snapshots):
    from test_cluster_safety import run_lossy_then_heal
    G = 8
    st0 = abi.empty_server_states(G, n_members)
    cpu = oracle_lib.Oracle(G, n_members); cpu.set_state(0, st0)
    kw = dict(p_snapshot=0.03, max_leaders=11, drop=0.15) if snapshots else {}
    sim = run_lossy_then_heal(cpu, G, n_members, seed, lossy_ticks=250, heal_ticks=150, **kw)
    ref = oracle_lib.Oracle(G, n_members); ref.set_state(0, st0)
    emu = Emu(emu_lib, G, n_members, specialised=True); emu.set_state(0, st0)
    for t, h in enumerate(sim.history):
        if isinstance(h, tuple):
            ref.set_state(h[1], h[2].reshape(1)); emu.set_state(h[1], h[2].reshape(1))
            continue
        do, ro = ref.step(h)
        dg, rg = emu.step(h)
        assert_same(f"closed loop tick {t}", dg, rg, emu.get_state(), do, ro, ref.get_state())
    emu.close()


import vector_runner as VR  # noqa: E402

_VECTORS = VR.load()["vectors"]


@pytest.mark.parametrize("specialised", [False, True], ids=["generic", "per_kind"])
def test_kernel_code_passes_the_reference_vectors(emu_lib, specialised):
    """The 80 vectors transcribed from the reference's suites, through the emulated device code."""
    for v in _VECTORS:
        try:
            VR.run_vector(lambda g, n: Emu(emu_lib, g, n, specialised=specialised), v)
        except AssertionError as e:
            raise AssertionError(f"vector {v['id']}: {e}") from e


@pytest.mark.parametrize("name,G,N,kw,mix,bm", [
    ("config3", 256, 5, {}, "MIX_CONFIG3", False),
    ("config5", 96, 7, dict(backlog=1024, boundaries=(3, 6)), "MIX_CONFIG5", True),
])
def test_kernel_code_on_the_baseline_workload_shapes(emu_lib, oracle_lib, name, G, N, kw, mix, bm):
    """BASELINE configs[2] (mixed append/vote traffic) and configs[4] (7 members, 1024-entry backlogs crossing
    term boundaries, wrong prev_log_term half of the time: the log-matching repair path) from the CPU workload
    generator, through the emulated device code."""
    from ra_amd import workload as W
    seed = 0x5EED0005
    st = W.initial_states(G, N, seed, **kw)
    cpu = oracle_lib.Oracle(G, N); cpu.set_state(0, st)
    emu = Emu(emu_lib, G, N, specialised=True); emu.set_state(0, st)
    for t in range(8):
        cur = cpu.get_state()
        m = W.gen_tick(cur, N, t, seed, getattr(W, mix), backlog_mode=bm)
        do, ro = cpu.step(m)
        dg, rg = emu.step(m)
        assert_same(f"{name} tick {t}", dg, rg, emu.get_state(), do, ro, cpu.get_state())
    emu.close()
    if bm:
        assert (cpu.get_state()["role"] == abi.ROLE_AWAIT_CONDITION).sum() > 0


# ---------------------------------------------------------------------------------------------------
# Whole kernels through the product's launchers: every lane of a block is a fiber (tests/native), so the LDS
# staging, the cooperative hot-line fetch, the block -> class mapping and the generator's atomics execute.

def _bind_launchers(L):
    vp, u32 = C.c_void_p, C.c_uint32
    L.emu_launch_tick.argtypes = [vp, C.c_int, vp, u32, vp, vp, u32, C.POINTER(u32)]
    L.emu_launch_classes.argtypes = [vp, vp, vp, u32, vp, vp, u32, C.POINTER(u32)]
    L.emu_launch_classes_dev.argtypes = [vp, vp, vp, u32, vp]
    L.emu_launch_pack.argtypes = [vp, u32, u32, vp]
    L.emu_launch_unpack.argtypes = [vp, u32, u32, vp]
    L.emu_launch_checksum.argtypes = [vp, u32, u32, vp]
    L.emu_launch_leaderboard.argtypes = [vp, vp]
    L.emu_launch_synth.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp]
    L.emu_synth_scratch_words.restype = u32
    L.emu_synth_scratch_words.argtypes = [vp]
    return L


class KernelEmu(Emu):
    """State through the pack/unpack KERNELS; a tick through the class-dispatch kernel (messages ordered by
    clause family like rgb_submit does, NOP slots through the generic kernel) or through the generic kernel."""

    def __init__(self, L, n_groups, n_members, mode="classes", **kw):
        super().__init__(_bind_launchers(L), n_groups, n_members, **kw)
        self.mode = mode

    def set_state(self, first, states):
        st = np.ascontiguousarray(states, dtype=abi.SERVER_STATE_DTYPE)
        assert self.L.emu_launch_pack(self.h, first, len(st), st.ctypes.data) == 0

    def get_state(self, first=0, n=None):
        n = self.S - first if n is None else n
        out = np.zeros(n, dtype=abi.SERVER_STATE_DTYPE)
        assert self.L.emu_launch_unpack(self.h, first, n, out.ctypes.data) == 0
        return out

    def step(self, msgs):
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        n = len(m)
        dec = np.zeros(n, dtype=abi.DECISION_DTYPE)
        cap = max(1, n * abi.MAX_MEMBERS)
        rpcs = np.zeros(cap, dtype=abi.RPC_DTYPE)
        nr = C.c_uint32(0)
        if self.mode == "generic":
            assert self.L.emu_launch_tick(self.h, -1, m.ctypes.data, n, dec.ctypes.data, rpcs.ctypes.data, cap,
                                          C.byref(nr)) == 0
            return dec, rpcs[:nr.value]
        perm = np.argsort(abi.family(m), kind="stable")            # the device order of rgb_submit
        ms = np.ascontiguousarray(m[perm])
        real = int((ms["kind"] != abi.MSG_NOP).sum())               # NOP has the last rank: a tail
        counts = np.bincount(abi.KIND_RANK[ms["kind"][:real]], minlength=15).astype(np.uint32)[:15]
        ds = np.zeros(n, dtype=abi.DECISION_DTYPE)
        assert self.L.emu_launch_classes(self.h, ms.ctypes.data, counts.ctypes.data, real, ds.ctypes.data,
                                         rpcs.ctypes.data, cap, C.byref(nr)) == 0
        if real < n:
            tail = np.zeros(n - real, dtype=abi.DECISION_DTYPE)
            nr2 = C.c_uint32(0)
            assert self.L.emu_launch_tick(self.h, -1, ms[real:].ctypes.data, n - real, tail.ctypes.data,
                                          rpcs[nr.value:].ctypes.data, cap - nr.value, C.byref(nr2)) == 0
            ds[real:] = tail
        dec[perm] = ds
        out = rpcs[:nr.value].copy()
        out["msg_index"] = perm[out["msg_index"]]                   # back to submission order, like rgb_collect
        return dec, out


@pytest.mark.parametrize("mode", ["classes", "generic"])
@pytest.mark.parametrize("n_members,seed", [(3, 321), (5, 322), (8, 323)])
def test_whole_kernels_equal_checker_on_random_ticks(emu_lib, oracle_lib, n_members, seed, mode):
    rng = np.random.default_rng(seed)
    G = 120
    st = fuzz.random_states(rng, G, n_members, max_runs=6)
    cpu = oracle_lib.Oracle(G, n_members); cpu.set_state(0, st)
    emu = KernelEmu(emu_lib, G, n_members, mode=mode); emu.set_state(0, st)
    empty = np.zeros(0, dtype=abi.RPC_DTYPE)
    assert_same("pack/unpack kernels", [], empty, emu.get_state(), [], empty, cpu.get_state())
    for tick in range(5):
        msgs = fuzz.random_msgs(rng, cpu.get_state(), n_members)
        do, ro = cpu.step(msgs)
        dg, rg = emu.step(msgs)
        assert_same(f"{mode} kernel N={n_members} tick {tick}", dg, rg, emu.get_state(), do, ro, cpu.get_state())
    # checksum and leaderboard kernels against their host-side definitions
    sums = np.zeros(emu.S, dtype=np.uint64)
    assert emu.L.emu_launch_checksum(emu.h, 0, emu.S, sums.ctypes.data) == 0
    assert np.array_equal(sums, oracle_lib.server_checksums(cpu.get_state()))
    from ra_amd import shard
    rows = np.zeros(G, dtype=abi.LEADERBOARD_DTYPE)
    assert emu.L.emu_launch_leaderboard(emu.h, rows.ctypes.data) == 0
    assert rows.tobytes() == shard.leaderboard_rows_from_states(cpu.get_state(), n_members).tobytes()
    emu.close()


@pytest.mark.parametrize("n_members,groups,ticks", [(5, 192, 24), (3, 128, 16)])
def test_load_generator_and_device_sized_dispatch(emu_lib, oracle_lib, n_members, groups, ticks):
    """The bench's inner loop on the CPU: rgb_synth_kernel (both passes, block reservations through atomics)
    writes a compacted, family-ordered tick from the device state; the class-dispatch kernel sizes itself from
    the generator's per-family totals; the checker replays every tick."""
    from ra_amd import workload as W
    G, N = groups, n_members
    S = G * N
    seed = 0x5EED0003
    st0 = W.initial_states(G, N, seed)
    cpu = oracle_lib.Oracle(G, N); cpu.set_state(0, st0)
    emu = KernelEmu(emu_lib, G, N); emu.set_state(0, st0)
    seen = 0
    for t in range(ticks):
        msgs = np.zeros(S, dtype=abi.MSG_DTYPE)
        scratch = np.zeros(emu.L.emu_synth_scratch_words(emu.h), dtype=np.uint32)
        kc = np.zeros(abi.N_KINDS, dtype=np.uint32)
        n = np.zeros(1, dtype=np.uint32)
        bc = np.zeros(engine.TRAIN_BUCKETS, dtype=np.uint32)
        assert emu.L.emu_launch_synth(emu.h, seed, t, msgs.ctypes.data, scratch.ctypes.data, kc.ctypes.data,
                                      n.ctypes.data, bc.ctypes.data) == 0
        nt = int(n[0])
        m = msgs[:nt]
        assert nt > G and not np.any(m["kind"] == abi.MSG_NOP)
        assert len(np.unique(m["server"])) == nt
        assert np.array_equal(np.bincount(m["kind"], minlength=abi.N_KINDS), kc)
        # bucket order = (class of the kind, group mod 8, success flag): every class is contiguous (what the class
        # kernel needs) and so is every (class, shard) pair (what a train launch needs)
        # (the sub-bucket -- bit 0 -- is the success flag for the replies and the producer's steady-state HINT for
        # append_entries_rpc / written: the generator's own counts say where a sub-bucket ends)
        bk = engine.train_bucket(m["kind"], m["flags"], m["server"], N)
        assert np.all(np.diff((bk >> 1).astype(np.int64)) >= 0), "tick is not in (class, shard) order"
        assert np.array_equal(np.bincount(bk >> 1, minlength=engine.TRAIN_BUCKETS // 2), bc.reshape(-1, 2).sum(axis=1))
        hinted = (m["kind"] == abi.MSG_AER) | (m["kind"] == abi.MSG_WRITTEN)
        gen_bucket = np.repeat(np.arange(engine.TRAIN_BUCKETS), bc)
        assert np.array_equal(gen_bucket[~hinted], bk[~hinted]), "a sub-bucket of an unhinted kind is not its success flag"
        assert np.array_equal(gen_bucket >> 1, bk >> 1)
        assert np.all(np.diff(abi.family(m) // 2) >= 0), "classes are not contiguous"
        # inside a bucket: group order at the generator's block granularity (64 groups), the same every tick
        key = gen_bucket.astype(np.int64) * (1 << 32) + (m["server"] // N) // 64
        assert np.all(np.diff(key) >= 0), "a bucket is not in group order"
        dec = np.zeros(S, dtype=abi.DECISION_DTYPE)
        assert emu.L.emu_launch_classes_dev(emu.h, msgs.ctypes.data, scratch.ctypes.data, S, dec.ctypes.data) == 0
        want, _ = cpu.step(m)
        bad = [i for i in range(nt) if dec[i].tobytes() != want[i].tobytes()]
        assert not bad, f"tick {t} slot {bad[0]}: msg={m[bad[0]]}\n kernel={dec[bad[0]]}\n checker={want[bad[0]]}"
        assert not np.any(want["flags"] & abi.F_INVARIANT)
        seen |= int(np.bitwise_or.reduce(want["flags"]))
    assert emu.get_state().tobytes() == cpu.get_state().tobytes()
    emu.close()
    for f in (abi.F_WROTE, abi.F_APPLIED, abi.F_PIPELINE, abi.F_REPLY):
        assert seen & f
