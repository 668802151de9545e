"""ctypes binding of libra_gpu_batch.so (the C ABI of include/ra_gpu_batch.h).

This is exactly what the Erlang NIF binds (ra_amd/csrc/ra_gpu_batch_nif.c); Python plays the
role of the gen_statem shell in tests and benchmarks.  No CPU fallback: a missing library or a
missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("RGB_LIB") or os.path.join(_CSRC, "libra_gpu_batch.so")

EXPORTS = [
    "rgb_abi_version", "rgb_struct_size", "rgb_strerror", "rgb_default_config", "rgb_open",
    "rgb_close", "rgb_last_hip_error", "rgb_register_groups", "rgb_n_servers", "rgb_upload_state",
    "rgb_download_state", "rgb_submit", "rgb_collect", "rgb_run_ticks_device", "rgb_snapshot",
    "rgb_snapshot_device", "rgb_state_checksum", "rgb_synchronize", "rgb_wait", "rgb_wake", "rgb_in_flight",
    "rgb_route", "rgb_submit_trains", "rgb_peek",
    "rgb_train_bucket", "rgb_train_plan_create", "rgb_train_plan_destroy", "rgb_train_plan_blocks_per_tick",
    "rgb_train_stamp_device", "rgb_train_run_device", "rgb_train_status", "rgb_train_form", "rgb_train_recoveries",
    "rgb_train_plan_create_snap", "rgb_train_run_snap_device", "rgb_snapshot_train_device", "rgb_train_seq_bytes",
    "rgb_train_plan_create_device", "rgb_train_plan_build_device", "rgb_train_plan_download", "rgb_train_plan_fit",
    "rgb_submit_seq", "rgb_set_seq_ranges_device", "rgb_collect_view", "rgb_release",
]
COMM_EXPORTS = ["rgb_comm_unique_id", "rgb_comm_init_rank", "rgb_comm_destroy", "rgb_comm_n_ranks", "rgb_comm_rank",
                "rgb_leaderboard_allgather", "rgb_leaderboard_allgather_host", "rgb_comm_last_error"]   # the one collective of the path (RCCL)
EXPORTS += COMM_EXPORTS
SYNTH_EXPORTS = ["rgb_synth_tick_device", "rgb_synth_tick_buckets_device", "rgb_synth_apply_tick_device",
                 "rgb_synth_tick_stamped_device", "rgb_synth_stamps_resync_device",
                 "rgb_synth_snapshot_mark_device", "rgb_synth_set_hint"]                                       # include/ra_gpu_batch_synth.h (bench tooling)
OPTIONAL_IN_OLD_BUILDS = {"rgb_synth_tick_stamped_device", "rgb_synth_stamps_resync_device", "rgb_train_form",
                          "rgb_train_recoveries", "rgb_train_plan_create_snap", "rgb_train_run_snap_device",
                          "rgb_snapshot_train_device", "rgb_train_seq_bytes", "rgb_synth_snapshot_mark_device",
                          "rgb_synth_set_hint", "rgb_train_plan_create_device", "rgb_train_plan_build_device",
                          "rgb_train_plan_download", "rgb_train_plan_fit", "rgb_submit_seq", "rgb_set_seq_ranges_device",
                          "rgb_collect_view", "rgb_release"} | set(COMM_EXPORTS)
WAL_EXPORTS = ["rgb_wal_adler32_device", "rgb_wal_adler32", "rgb_wal_layout", "rgb_wal_frame_device",
               "rgb_wal_frame", "rgb_wal_scan", "rgb_wal_validate"]                            # include/ra_gpu_wal.h


class RgbError(RuntimeError):
    def __init__(self, code: int, what: str, hip: int = 0):
        self.code, self.hip = code, hip
        super().__init__(f"{what}: rc={code} ({_strerror(code)})" + (f" hipError={hip}" if hip else ""))


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("rgb_kernels.hip", "rgb_api.hip", "rgb_wal.hip", "rgb_wal_host.cpp", "rgb_comm.cpp", "rgb_internal.h")]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "ra_gpu_wal.h"))
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "ra_gpu_batch.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "libra_gpu_batch.so"])
    return LIB_PATH


_lib = None


def lib():
    """Load the library and verify the ABI.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C ra_amd/csrc).  ra_gpu_batch has no CPU fallback.")
    # One HIP/HSA runtime per process: when PyTorch is installed its bundled libamdhip64.so.7 must
    # be the copy in the process (a second HSA runtime cannot open the device), so import torch
    # BEFORE the library resolves that soname.  The library itself has no torch dependency (the
    # Erlang NIF uses /opt/rocm's runtime).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the binding
        pass
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS + SYNTH_EXPORTS + WAL_EXPORTS:
        if not hasattr(L, name):
            # tools/ only: RGB_LIB=<an older build> for same-box A/B timing of bench.py (tools/gpu_ab.sh --variants)
            if os.environ.get("RGB_LIB") and name in OPTIONAL_IN_OLD_BUILDS:
                continue
            raise RuntimeError(f"libra_gpu_batch.so does not export {name}")
    vp, u32, u64p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)
    L.rgb_abi_version.restype = C.c_uint32
    L.rgb_struct_size.restype = C.c_size_t
    L.rgb_struct_size.argtypes = [C.c_int]
    L.rgb_strerror.restype = C.c_char_p
    L.rgb_strerror.argtypes = [C.c_int]
    L.rgb_default_config.argtypes = [vp]
    L.rgb_open.argtypes = [vp, C.POINTER(vp)]
    L.rgb_close.argtypes = [vp]
    L.rgb_last_hip_error.argtypes = [vp]
    L.rgb_register_groups.argtypes = [vp, u32, u32]
    L.rgb_n_servers.restype = C.c_uint32
    L.rgb_n_servers.argtypes = [vp]
    L.rgb_upload_state.argtypes = [vp, u32, u32, vp]
    L.rgb_download_state.argtypes = [vp, u32, u32, vp]
    L.rgb_submit.argtypes = [vp, vp, u32, C.c_uint64]
    if hasattr(L, "rgb_submit_seq"):
        L.rgb_submit_seq.argtypes = [vp, vp, u32, C.c_uint64, vp, u32]
        L.rgb_set_seq_ranges_device.argtypes = [vp, vp, u32]
    L.rgb_collect.argtypes = [vp, vp, u32, C.POINTER(u32), vp, u32, C.POINTER(u32), u64p]
    if hasattr(L, "rgb_collect_view"):
        L.rgb_collect_view.argtypes = [vp, vp]
        L.rgb_release.argtypes = [vp, u32]
    L.rgb_run_ticks_device.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp, vp]
    L.rgb_snapshot.argtypes = [vp, vp]
    L.rgb_snapshot_device.argtypes = [vp, vp, vp]
    L.rgb_state_checksum.argtypes = [vp, u32, u32, u64p]
    L.rgb_synchronize.argtypes = [vp]
    L.rgb_wait.argtypes = [vp, u32]
    L.rgb_wake.argtypes = [vp]
    L.rgb_wake.restype = None
    L.rgb_in_flight.argtypes = [vp]
    L.rgb_peek.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.rgb_submit_trains.restype = C.c_uint32
    L.rgb_submit_trains.argtypes = [vp]
    L.rgb_in_flight.restype = C.c_uint32
    L.rgb_route.argtypes = [C.c_uint64, u32]
    L.rgb_route.restype = C.c_uint32
    L.rgb_synth_tick_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp]
    L.rgb_synth_tick_buckets_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp]
    if hasattr(L, "rgb_synth_tick_stamped_device"):
        L.rgb_synth_tick_stamped_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp, vp]
        L.rgb_synth_stamps_resync_device.argtypes = [vp, vp]
    if hasattr(L, "rgb_comm_init_rank"):
        L.rgb_comm_unique_id.argtypes = [vp]
        L.rgb_comm_init_rank.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
        L.rgb_comm_destroy.argtypes = [vp]
        L.rgb_comm_destroy.restype = None
        L.rgb_comm_n_ranks.argtypes = [vp]
        L.rgb_comm_n_ranks.restype = C.c_uint32
        L.rgb_comm_rank.argtypes = [vp]
        L.rgb_comm_rank.restype = C.c_uint32
        L.rgb_leaderboard_allgather.argtypes = [vp, vp, vp, u32, vp, vp]
        L.rgb_comm_last_error.restype = C.c_char_p
    if hasattr(L, "rgb_debug_inject_train_fault"):
        L.rgb_debug_inject_train_fault.argtypes = [vp, u32]
        L.rgb_debug_inject_train_fault.restype = None
        L.rgb_train_recoveries.argtypes = [vp]
        L.rgb_train_recoveries.restype = C.c_uint32
        L.rgb_train_form.argtypes = [vp]
        L.rgb_train_form.restype = C.c_uint32
    L.rgb_train_bucket.restype = C.c_uint32
    L.rgb_train_bucket.argtypes = [u32, u32, u32, u32]
    L.rgb_train_plan_create.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.rgb_train_plan_destroy.argtypes = [vp]
    L.rgb_train_plan_destroy.restype = None
    L.rgb_train_plan_blocks_per_tick.restype = C.c_uint32
    L.rgb_train_plan_blocks_per_tick.argtypes = [vp]
    L.rgb_train_stamp_device.argtypes = [vp, vp, vp, u32, vp, u32, vp]
    L.rgb_train_run_device.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp, u32, vp]
    if hasattr(L, "rgb_train_run_snap_device"):
        L.rgb_train_plan_create_snap.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
        L.rgb_train_run_snap_device.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp, u32, vp, vp, vp]
        L.rgb_snapshot_train_device.argtypes = [vp, vp, vp]
        L.rgb_train_seq_bytes.restype = C.c_uint32
        L.rgb_train_seq_bytes.argtypes = [vp]
        L.rgb_synth_snapshot_mark_device.argtypes = [vp, vp, vp]
    if hasattr(L, "rgb_train_plan_create_device"):          # ABI v8
        L.rgb_train_plan_create_device.argtypes = [vp, u32, u32, C.POINTER(vp)]
        L.rgb_train_plan_build_device.argtypes = [vp, vp, u32, u32, vp, vp]
        L.rgb_train_plan_download.argtypes = [vp, vp, u32, vp, vp, u32]
    if hasattr(L, "rgb_train_plan_fit"):
        L.rgb_train_plan_fit.argtypes = [vp, vp, u32, u32, vp]
    L.rgb_train_status.argtypes = [vp, C.POINTER(u32), vp]
    L.rgb_synth_apply_tick_device.argtypes = [vp, vp, u32, vp, vp, vp]
    if hasattr(L, "rgb_synth_set_hint"):
        L.rgb_synth_set_hint.argtypes = [vp, u32]
    L.rgb_wal_adler32_device.argtypes = [vp, vp, u32, vp, C.c_uint64, vp, vp]
    L.rgb_wal_adler32.argtypes = [vp, vp, u32, vp, C.c_uint64, vp]
    L.rgb_wal_layout.restype = C.c_uint64
    L.rgb_wal_layout.argtypes = [vp, u32, C.c_uint64]
    L.rgb_wal_frame_device.argtypes = [vp, vp, u32, vp, C.c_uint64, vp, C.c_uint64, vp, u32, vp]
    L.rgb_wal_frame.argtypes = [vp, vp, u32, vp, C.c_uint64, vp, C.c_uint64, u32]
    L.rgb_wal_scan.argtypes = [vp, C.c_uint64, vp, u32, C.POINTER(C.c_uint32), u64p, C.POINTER(C.c_uint32)]
    L.rgb_wal_validate.argtypes = [vp, vp, C.c_uint64, vp, u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if L.rgb_abi_version() != abi.ABI_VERSION and not (os.environ.get("RGB_LIB") and L.rgb_abi_version() == abi.ABI_VERSION - 1):
        raise RuntimeError("ABI version mismatch")     # (RGB_LIB=<the previous ABI's build>: A/B timing, tools/ only)
    for i, dt in enumerate(abi.STRUCT_DTYPES):
        if os.environ.get("RGB_LIB") and i >= 6 and L.rgb_abi_version() < abi.ABI_VERSION:
            continue                                       # (an older A/B build has no rgb_view)
        if L.rgb_struct_size(i) != dt.itemsize:
            raise RuntimeError(f"struct {i}: C size {L.rgb_struct_size(i)} != numpy {dt.itemsize}")
    _lib = L
    return L


def _strerror(code: int) -> str:
    try:
        return lib().rgb_strerror(code).decode()
    except Exception:
        return "?"


def default_config() -> np.ndarray:
    cfg = np.zeros(1, dtype=abi.CONFIG_DTYPE)
    lib().rgb_default_config(cfg.ctypes.data)
    return cfg


class RaGpuBatch:
    """One rgb_ctx: the device-resident state of n_groups x n_members ra_servers on one GPU."""

    def __init__(self, n_groups: int, n_members: int, device: int = 0, max_runs: int = 8,
                 ring_slots: int = 4, ring_capacity: int = 65536, max_pipeline_count: int = 0,
                 max_aer_batch: int = 0, flags: int = 0):
        self._L = lib()
        cfg = default_config()
        cfg["device"] = device
        cfg["max_runs"] = max_runs
        cfg["ring_slots"] = ring_slots
        cfg["ring_capacity"] = ring_capacity
        if max_pipeline_count:
            cfg["max_pipeline_count"] = max_pipeline_count
        if max_aer_batch:
            cfg["max_aer_batch"] = max_aer_batch
        cfg["flags"] = flags                    # abi.CFG_SUBMIT_TRAINS: fuse the sub-tick rounds of a batch into a train
        h = C.c_void_p()
        rc = self._L.rgb_open(cfg.ctypes.data, C.byref(h))
        if rc:
            raise RgbError(rc, "rgb_open")
        self._h = h
        self.n_groups, self.n_members = n_groups, n_members
        self.n_servers = n_groups * n_members
        self.ring_capacity = ring_capacity
        self._check(self._L.rgb_register_groups(self._h, n_groups, n_members), "rgb_register_groups")

    # -- plumbing ------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc:
            raise RgbError(rc, what, self._L.rgb_last_hip_error(self._h) if self._h else 0)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgb_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state transfer ----------------------------------------------------------------
    def set_state(self, first: int, states: np.ndarray):
        st = np.ascontiguousarray(states, dtype=abi.SERVER_STATE_DTYPE)
        self._check(self._L.rgb_upload_state(self._h, first, len(st), st.ctypes.data), "rgb_upload_state")

    def get_state(self, first: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n_servers - first if n is None else n
        out = np.zeros(n, dtype=abi.SERVER_STATE_DTYPE)
        self._check(self._L.rgb_download_state(self._h, first, n, out.ctypes.data), "rgb_download_state")
        return out

    # -- host path -------------------------------------------------------------------
    def submit(self, msgs: np.ndarray, tick: int = 0, seq_ranges: np.ndarray | None = None):
        """seq_ranges (uint64[n][2]: first, last): the batch's range list -- the lower ranges of written events of more
        than two ranges (abi.MF_SEQX: record fields c = first entry, n_entries = how many) -- rgb_submit_seq."""
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        if seq_ranges is None or len(seq_ranges) == 0:
            self._check(self._L.rgb_submit(self._h, m.ctypes.data, len(m), tick), "rgb_submit")
        else:
            r = np.ascontiguousarray(seq_ranges, dtype=np.uint64).reshape(-1, 2)
            self._check(self._L.rgb_submit_seq(self._h, m.ctypes.data, len(m), tick, r.ctypes.data, len(r)), "rgb_submit_seq")

    def collect(self, cap: int | None = None, rpc_cap: int | None = None, out=None):
        """Wait for the oldest submitted batch.  `out` = (decisions, rpcs) preallocated arrays to
        reuse (the NIF hands BEAM binaries instead); returns views of the filled parts."""
        if out is not None:
            dec, rpcs = out
            cap, rpc_cap = len(dec), len(rpcs)
        else:
            cap = self.ring_capacity if cap is None else cap
            rpc_cap = cap * max(self.n_members - 1, 1) if rpc_cap is None else rpc_cap
            dec = np.empty(cap, dtype=abi.DECISION_DTYPE)
            rpcs = np.empty(max(rpc_cap, 1), dtype=abi.RPC_DTYPE)
        n, nr, tick = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        rc = self._L.rgb_collect(self._h, dec.ctypes.data, cap, C.byref(n), rpcs.ctypes.data,
                                 rpc_cap, C.byref(nr), C.byref(tick))
        if rc in (abi.E_FULL, abi.E_INVAL) and out is None and (n.value > cap or nr.value > rpc_cap):
            # the batch stayed in the ring and reported the sizes it needs: retry once with them
            cap, rpc_cap = max(cap, n.value), max(rpc_cap, nr.value)
            dec = np.empty(max(cap, 1), dtype=abi.DECISION_DTYPE)
            rpcs = np.empty(max(rpc_cap, 1), dtype=abi.RPC_DTYPE)
            rc = self._L.rgb_collect(self._h, dec.ctypes.data, cap, C.byref(n), rpcs.ctypes.data,
                                     rpc_cap, C.byref(nr), C.byref(tick))
        self._check(rc, "rgb_collect")
        return dec[:n.value], rpcs[:nr.value], tick.value

    def collect_view(self):
        """The oldest submitted batch IN PLACE (rgb_collect_view, ABI v9): (decisions, rpcs, tick, slot) where the two
        arrays are numpy views of the pinned slot the device wrote -- no copy.  They are valid until release(slot);
        the slot is not free for rgb_submit before that."""
        v = np.zeros(1, dtype=abi.VIEW_DTYPE)
        self._check(self._L.rgb_collect_view(self._h, v.ctypes.data), "rgb_collect_view")
        n, nr = int(v["n"][0]), int(v["n_rpcs"][0])

        def arr(ptr, count, dt):
            if not count:
                return np.empty(0, dtype=dt)
            buf = (C.c_char * (count * dt.itemsize)).from_address(int(ptr))
            return np.frombuffer(buf, dtype=dt, count=count)
        return arr(v["decisions"][0], n, abi.DECISION_DTYPE), arr(v["rpcs"][0], nr, abi.RPC_DTYPE), int(v["tick"][0]), int(v["slot"][0])

    def release(self, slot: int):
        self._check(self._L.rgb_release(self._h, slot), "rgb_release")

    def wait(self, timeout_ms: int = 1000) -> bool:
        """Park until a batch is in flight (True) or the timeout / a wake() passes (False)."""
        return self._L.rgb_wait(self._h, timeout_ms) == abi.OK

    def wake(self):
        self._L.rgb_wake(self._h)

    def submit_trains(self) -> int:
        """Batches whose sub-tick rounds ran as one train launch (rgb_submit_trains)."""
        return int(self._L.rgb_submit_trains(self._h))

    @property
    def in_flight(self) -> int:
        return int(self._L.rgb_in_flight(self._h))

    def step(self, msgs: np.ndarray, seq_ranges: np.ndarray | None = None):
        """submit + collect: decisions in submission order and the pipelined rpcs."""
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        out_d, out_r = [], []
        for off in range(0, max(len(m), 1), self.ring_capacity):
            chunk = m[off:off + self.ring_capacity]
            self.submit(chunk, seq_ranges=seq_ranges)
            d, r, _ = self.collect(cap=max(len(chunk), 1))
            r = r.copy()
            r["msg_index"] += off
            out_d.append(d)
            out_r.append(r)
        return np.concatenate(out_d), np.concatenate(out_r)

    # -- device-resident path ------------------------------------------------------------
    def run_ticks_device(self, d_msgs: int, tick_stride: int, n_ticks: int, d_decisions: int,
                         d_rpcs: int = 0, stream: int = 0, tick_counts: np.ndarray | None = None,
                         d_tick_counts: int = 0, kind_counts: np.ndarray | None = None):
        """Raw device pointers (e.g. torch tensors' data_ptr()); enqueues and returns.
        tick_counts: messages per tick (uint32, host) or None for tick_stride each;
        d_tick_counts: device uint32 array with the real size of each tick (device producers);
        kind_counts: uint32 [n_ticks, n_kinds] (host) for family-ordered ticks -> specialised,
        concurrent per-kind kernels."""
        cp = None
        if tick_counts is not None:
            tc = np.ascontiguousarray(tick_counts, dtype=np.uint32)
            assert len(tc) >= n_ticks
            cp = tc.ctypes.data
        kp = None
        if kind_counts is not None:
            kcs = np.ascontiguousarray(kind_counts, dtype=np.uint32)
            assert kcs.shape == (n_ticks, abi.N_KINDS), kcs.shape
            kp = kcs.ctypes.data
        self._check(self._L.rgb_run_ticks_device(self._h, d_msgs, tick_stride, cp, d_tick_counts or None, kp,
                                                 n_ticks, d_decisions, d_rpcs or None, stream or None),
                    "rgb_run_ticks_device")

    def synth_tick_device(self, seed: int, tick: int, d_msgs: int, d_kind_counts: int = 0, d_n: int = 0,
                          stream: int = 0):
        """Device-side load generator (include/ra_gpu_batch_synth.h): one compacted, family-ordered
        tick from the current device state into d_msgs (room for n_servers messages)."""
        self._check(self._L.rgb_synth_tick_device(self._h, seed, tick, d_msgs, d_kind_counts or None,
                                                  d_n or None, stream or None), "rgb_synth_tick_device")

    def synth_tick_buckets_device(self, seed: int, tick: int, d_msgs: int, d_kind_counts: int = 0, d_n: int = 0,
                                  d_bucket_counts: int = 0, stream: int = 0):
        """The load generator, also writing uint32[TRAIN_BUCKETS] message counts per train bucket."""
        self._check(self._L.rgb_synth_tick_buckets_device(self._h, seed & (2**64 - 1), tick, d_msgs, d_kind_counts,
                                                         d_n, d_bucket_counts, stream), "rgb_synth_tick_buckets_device")

    def synth_tick_stamped_device(self, seed: int, tick: int, d_msgs: int, d_kind_counts: int = 0, d_n: int = 0,
                                  d_bucket_counts: int = 0, d_stamps: int = 0, stream: int = 0):
        """The load generator, also writing the tick's train stamps (uint8 per message slot): the producer's own
        count of the messages it has addressed to each server."""
        self._check(self._L.rgb_synth_tick_stamped_device(self._h, seed & (2**64 - 1), tick, d_msgs, d_kind_counts,
                                                         d_n, d_bucket_counts, d_stamps, stream),
                    "rgb_synth_tick_stamped_device")

    def synth_stamps_resync_device(self, stream: int = 0):
        """The generator's per-server message counts := the servers' sequence bytes as they are now."""
        self._check(self._L.rgb_synth_stamps_resync_device(self._h, stream), "rgb_synth_stamps_resync_device")

    # ---- train launches: several device-resident ticks in one launch ----
    def train_plan(self, bucket_counts: np.ndarray) -> "TrainPlan":
        return TrainPlan(self, bucket_counts)

    def train_stamp_device(self, d_msgs: int, d_stamps: int, tick_stride: int, tick_counts: np.ndarray, stream: int = 0):
        """d_stamps: uint8[n_ticks * tick_stride] on the device, the sequence value every message must find."""
        tc = np.ascontiguousarray(tick_counts, dtype=np.uint32)
        self._check(self._L.rgb_train_stamp_device(self._h, d_msgs, d_stamps, tick_stride, tc.ctypes.data, len(tc),
                                                   stream), "rgb_train_stamp_device")

    def train_run_device(self, plan: "TrainPlan", first_tick: int, n_ticks: int, d_msgs: int, d_stamps: int,
                         tick_stride: int, d_decisions: int, d_rpcs: int = 0, rpc_ring: int = 1, stream: int = 0):
        self._check(self._L.rgb_train_run_device(self._h, plan.h, first_tick, n_ticks, d_msgs, d_stamps, tick_stride,
                                                 d_decisions, d_rpcs, rpc_ring, stream), "rgb_train_run_device")

    def train_plan_snap(self, bucket_counts: np.ndarray, snapshot_every: int) -> "TrainPlan":
        """A plan whose ticks k * snapshot_every (k >= 1) carry the leaderboard snapshot in front of them."""
        return TrainPlan(self, bucket_counts, snapshot_every)

    def train_run_snap_device(self, plan: "TrainPlan", first_tick: int, n_ticks: int, d_msgs: int, d_stamps: int,
                              tick_stride: int, d_decisions: int, d_rpcs: int, rpc_ring: int, d_snap_stamps: int,
                              d_snap_rows: int, stream: int = 0):
        self._check(self._L.rgb_train_run_snap_device(self._h, plan.h, first_tick, n_ticks, d_msgs, d_stamps, tick_stride,
                                                      d_decisions, d_rpcs, rpc_ring, d_snap_stamps, d_snap_rows, stream),
                    "rgb_train_run_snap_device")

    def snapshot_train_device(self, d_rows: int, stream: int = 0):
        """rgb_snapshot_device + every sequence byte advanced (a snapshot boundary between two train launches)."""
        self._check(self._L.rgb_snapshot_train_device(self._h, d_rows, stream or None), "rgb_snapshot_train_device")

    def train_seq_bytes(self) -> int:
        return int(self._L.rgb_train_seq_bytes(self._h))

    def synth_snapshot_mark_device(self, d_snap_stamps: int = 0, stream: int = 0):
        """The generator's side of a snapshot boundary: its counts -> d_snap_stamps, then + 1."""
        self._check(self._L.rgb_synth_snapshot_mark_device(self._h, d_snap_stamps or None, stream or None),
                    "rgb_synth_snapshot_mark_device")

    def train_status(self, check: bool = True):
        """(error flags, XCD of every shard) of the trains run since the last call; the caller has synchronised."""
        flags = C.c_uint32(0)
        xcc = np.zeros(8, dtype=np.uint32)
        rc = self._L.rgb_train_status(self._h, C.byref(flags), xcc.ctypes.data)
        if check and rc:
            raise RgbError(rc, f"train launch failed (flags={flags.value}: 1 = placement, 2 = spin bound)")
        return int(flags.value), xcc

    def inject_train_fault(self, fault: int):
        """Fail-safe tests: the next train batch of submit() gets a fault (1 = a stamp that never comes up,
        2 = two messages bucketed under each other's shard).  The engine must repair the failed launch itself."""
        self._L.rgb_debug_inject_train_fault(self._h, fault)

    def train_form(self) -> str:
        return {0: "none", 1: "dealt", 2: "persistent"}[int(self._L.rgb_train_form(self._h))] if hasattr(self._L, "rgb_train_form") else "round-3 build"

    def train_recoveries(self) -> int:
        return int(self._L.rgb_train_recoveries(self._h))

    def synth_set_hint(self, level: int):
        """The generator's ordering hint (include/ra_gpu_batch_synth.h): 0 none, 1 state name, 2 + header compare."""
        self._check(self._L.rgb_synth_set_hint(self._h, level), "rgb_synth_set_hint")

    def synth_apply_tick_device(self, d_msgs: int, max_msgs: int, d_decisions: int, d_rpcs: int = 0,
                                stream: int = 0):
        """Apply the tick just generated by synth_tick_device (class-dispatch kernel sized on-device)."""
        self._check(self._L.rgb_synth_apply_tick_device(self._h, d_msgs, max_msgs, d_decisions, d_rpcs or None,
                                                        stream or None), "rgb_synth_apply_tick_device")

    # -- WAL entry checksums (include/ra_gpu_wal.h) -----------------------------------
    def wal_adler32_device(self, d_entries: int, n: int, d_data: int, data_bytes: int, d_checksums: int,
                           stream: int = 0):
        """adler32(<<Idx:64, Term:64, Payload>>) of n rgb_wal_entry records whose payloads are
        resident in device memory (src/ra_log_wal.erl:528-534, 861, 873, 1028); enqueues and returns."""
        self._check(self._L.rgb_wal_adler32_device(self._h, d_entries, n, d_data, data_bytes, d_checksums,
                                                   stream or None), "rgb_wal_adler32_device")

    def wal_adler32(self, entries: np.ndarray, data: np.ndarray) -> np.ndarray:
        """Host-buffer form: checksums of `entries` (abi.WAL_ENTRY_DTYPE) over the packed payload bytes."""
        entries = np.ascontiguousarray(entries, dtype=abi.WAL_ENTRY_DTYPE)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(len(entries), dtype=np.uint32)
        self._check(self._L.rgb_wal_adler32(self._h, entries.ctypes.data, len(entries),
                                            data.ctypes.data if len(data) else None, len(data),
                                            out.ctypes.data), "rgb_wal_adler32")
        return out

    # -- WAL record framing and recovery validation (include/ra_gpu_wal.h) ---------------
    def wal_frame_device(self, d_records: int, n: int, d_data: int, data_bytes: int, d_out: int, out_bytes: int,
                         d_checksums: int = 0, flags: int = 0, stream: int = 0):
        """Frame n rgb_wal_record records into d_out (src/ra_log_wal.erl:513-537); enqueues and returns."""
        self._check(self._L.rgb_wal_frame_device(self._h, d_records, n, d_data, data_bytes, d_out, out_bytes,
                                                 d_checksums or None, flags, stream or None), "rgb_wal_frame_device")

    def wal_frame(self, records: np.ndarray, data: np.ndarray, out_bytes: int, flags: int = 0) -> np.ndarray:
        """Host-buffer form: the framed bytes [0, out_bytes) of `records` (abi.WAL_RECORD_DTYPE)."""
        records = np.ascontiguousarray(records, dtype=abi.WAL_RECORD_DTYPE)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(out_bytes, dtype=np.uint8)
        self._check(self._L.rgb_wal_frame(self._h, records.ctypes.data, len(records),
                                          data.ctypes.data if len(data) else None, len(data),
                                          out.ctypes.data if out_bytes else None, out_bytes, flags), "rgb_wal_frame")
        return out

    def wal_validate(self, file_bytes: np.ndarray, scanned: np.ndarray):
        """(n_ok, status) of the scanned records of a WAL file: validate_checksum/4 on every record
        flagged WAL_REC_VALIDATE, is_last_record/3 on the first failure (src/ra_log_wal.erl:994-1033)."""
        file_bytes = np.ascontiguousarray(file_bytes, dtype=np.uint8)
        scanned = np.ascontiguousarray(scanned, dtype=abi.WAL_SCANNED_DTYPE)
        n_ok, status = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.rgb_wal_validate(self._h, file_bytes.ctypes.data, len(file_bytes),
                                             scanned.ctypes.data if len(scanned) else None, len(scanned),
                                             C.byref(n_ok), C.byref(status)), "rgb_wal_validate")
        return n_ok.value, status.value

    # -- observability -----------------------------------------------------------------
    def snapshot(self) -> np.ndarray:
        rows = np.zeros(self.n_groups, dtype=abi.LEADERBOARD_DTYPE)
        self._check(self._L.rgb_snapshot(self._h, rows.ctypes.data), "rgb_snapshot")
        return rows

    def snapshot_device(self, d_rows: int, stream: int = 0):
        self._check(self._L.rgb_snapshot_device(self._h, d_rows, stream or None), "rgb_snapshot_device")

    def state_checksum(self, first: int = 0, n: int | None = None) -> int:
        n = self.n_servers - first if n is None else n
        out = C.c_uint64(0)
        self._check(self._L.rgb_state_checksum(self._h, first, n, C.byref(out)), "rgb_state_checksum")
        return out.value

    def synchronize(self):
        self._check(self._L.rgb_synchronize(self._h), "rgb_synchronize")


TRAIN_BUCKETS = 256


def comm_unique_id() -> bytes:
    """RGB_COMM_ID_BYTES of a fresh communicator id (one rank creates it, the others receive it from the host)."""
    buf = C.create_string_buffer(abi.COMM_ID_BYTES)
    rc = lib().rgb_comm_unique_id(buf)
    if rc:
        raise RgbError(rc, "rgb_comm_unique_id")
    return buf.raw


class Comm:
    """rgb_comm: this context's rank in the leaderboard all-gather (RCCL behind the C ABI)."""

    def __init__(self, eng: "RaGpuBatch", comm_id: bytes, n_ranks: int, rank: int):
        assert len(comm_id) == abi.COMM_ID_BYTES
        self.eng, self.n_ranks, self.rank = eng, n_ranks, rank
        h = C.c_void_p()
        rc = eng._L.rgb_comm_init_rank(eng._h, comm_id, n_ranks, rank, C.byref(h))
        if rc:
            raise RgbError(rc, f"rgb_comm_init_rank: {eng._L.rgb_comm_last_error().decode()}")
        self.h = h

    def allgather_leaderboard(self, d_rows_local: int, n_rows: int, d_rows_all: int, stream: int = 0):
        """Every rank's n_rows leaderboard rows (device memory) -> d_rows_all[rank * n_rows ..], on `stream`."""
        rc = self.eng._L.rgb_leaderboard_allgather(self.eng._h, self.h, d_rows_local, n_rows, d_rows_all, stream or None)
        if rc:
            raise RgbError(rc, f"rgb_leaderboard_allgather: {self.eng._L.rgb_comm_last_error().decode()}")

    def close(self):
        if self.h:
            self.eng._L.rgb_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrainPlan:
    """Device plan of a train (rgb_train_plan): bucket_counts = uint32[n_ticks][TRAIN_BUCKETS]."""

    def __init__(self, eng: "RaGpuBatch", bucket_counts, snapshot_every: int = 0, device_ticks: int = 0):
        """bucket_counts (host array): the plan is built on the host and uploaded.  device_ticks > 0 (bucket_counts =
        None): an empty plan of that many ticks for build_device() -- rgb_train_plan_create_device."""
        h = C.c_void_p()
        if device_ticks:
            self.eng, self.n_ticks, self.snapshot_every = eng, device_ticks, snapshot_every
            eng._check(eng._L.rgb_train_plan_create_device(eng._h, device_ticks, snapshot_every, C.byref(h)),
                       "rgb_train_plan_create_device")
            self.h = h
            self.blocks_per_tick = int(eng._L.rgb_train_plan_blocks_per_tick(h))
            return
        bc = np.ascontiguousarray(bucket_counts, dtype=np.uint32).reshape(-1, TRAIN_BUCKETS)
        self.eng, self.n_ticks, self.snapshot_every = eng, len(bc), snapshot_every
        if snapshot_every:
            eng._check(eng._L.rgb_train_plan_create_snap(eng._h, bc.ctypes.data, len(bc), snapshot_every, C.byref(h)),
                       "rgb_train_plan_create_snap")
        else:
            eng._check(eng._L.rgb_train_plan_create(eng._h, bc.ctypes.data, len(bc), C.byref(h)), "rgb_train_plan_create")
        self.h = h
        self.blocks_per_tick = int(eng._L.rgb_train_plan_blocks_per_tick(h))

    def download(self, tick: int):
        """(header words uint32[16], off uint32[30][8], cnt uint32[30][8], row table uint32[n_rows]) of one tick as it
        stands on the device (rgb_train_plan_download; synchronises)."""
        raw = np.zeros(496, dtype=np.uint32)                      # sizeof(rgb_train_tick) = 1984
        rows = np.zeros(max(self.blocks_per_tick // 8, 1), dtype=np.uint32)
        n = self.eng._L.rgb_train_plan_download(self.eng._h, self.h, tick, raw.ctypes.data, rows.ctypes.data, len(rows))
        if n < 0:
            raise RgbError(n, "rgb_train_plan_download")
        return raw[:16].copy(), raw[16:256].reshape(30, 8).copy(), raw[256:496].reshape(30, 8).copy(), rows[:n].copy()

    def build_device(self, first_tick: int, n_ticks: int, d_bucket_counts: int, stream: int = 0):
        """Ticks [first_tick, first_tick + n_ticks) from uint32[n_ticks][TRAIN_BUCKETS] in DEVICE memory, one kernel
        on `stream` (rgb_train_plan_build_device): nothing of the plan passes through the host."""
        self.eng._check(self.eng._L.rgb_train_plan_build_device(self.eng._h, self.h, first_tick, n_ticks, d_bucket_counts,
                                                                stream or None), "rgb_train_plan_build_device")

    def fit(self, first_tick: int, n_ticks: int, stream: int = 0):
        """Tell the host the rows of the built ticks (rgb_train_plan_fit: 4 bytes per tick come back; synchronises):
        launches over them take a grid of their rows instead of the rows bound."""
        self.eng._check(self.eng._L.rgb_train_plan_fit(self.eng._h, self.h, first_tick, n_ticks, stream or None), "rgb_train_plan_fit")
        self.blocks_per_tick = int(self.eng._L.rgb_train_plan_blocks_per_tick(self.h))

    def close(self):
        if self.h:
            self.eng._L.rgb_train_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_bucket(kind, flags, server, n_members):
    """rgb_train_bucket over numpy arrays: (class rank of the kind, group mod 8, success flag)."""
    rank = np.array([15, 0, 1, 5, 6, 2, 4, 3, 7, 8, 9, 10, 11, 12, 13, 14], dtype=np.uint32)[np.asarray(kind) & 15]
    shard = (np.asarray(server, dtype=np.uint32) // n_members) & 7
    return (rank * 8 + shard) * 2 + (np.asarray(flags, dtype=np.uint32) & 1)


def wal_layout(records: np.ndarray, base: int = 0) -> int:
    """Fill out_offset for back-to-back records from `base`; returns the end offset (host helper)."""
    assert records.dtype == abi.WAL_RECORD_DTYPE and records.flags["C_CONTIGUOUS"]
    return int(lib().rgb_wal_layout(records.ctypes.data if len(records) else None, len(records), base))


def wal_scan(file_bytes, cap: int | None = None):
    """recover_records/5's record walk over a whole WAL file (host code, no device):
    (records as abi.WAL_SCANNED_DTYPE, consumed bytes, end reason abi.WAL_END_*)."""
    buf = np.frombuffer(bytes(file_bytes), dtype=np.uint8) if not isinstance(file_bytes, np.ndarray) else \
        np.ascontiguousarray(file_bytes, dtype=np.uint8)
    n, consumed, end = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
    if cap is None:                                   # count first, then size the array
        rc = lib().rgb_wal_scan(buf.ctypes.data if len(buf) else None, len(buf), None, 0,
                                C.byref(n), C.byref(consumed), C.byref(end))
        if rc != 0:
            raise RgbError(rc, "rgb_wal_scan")
        cap = max(1, n.value)
    out = np.zeros(cap, dtype=abi.WAL_SCANNED_DTYPE)
    rc = lib().rgb_wal_scan(buf.ctypes.data if len(buf) else None, len(buf), out.ctypes.data, cap,
                            C.byref(n), C.byref(consumed), C.byref(end))
    if rc != 0:
        raise RgbError(rc, "rgb_wal_scan")
    return out[:n.value].copy(), consumed.value, end.value


def combine_checksums(per_server: np.ndarray, first: int = 0) -> int:
    """The host-side 'checksum of checksums' rgb_state_checksum returns."""
    acc = 0
    for k, c in enumerate(per_server.tolist()):
        acc = (acc + c * (2 * (first + k) + 1)) & 0xFFFFFFFFFFFFFFFF
    return acc
