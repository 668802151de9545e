"""ctypes binding of the CPU checker (oracle/libra_oracle.so).  TEST INFRASTRUCTURE: may be
imported only by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from ra_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libra_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ra_oracle.c", "wal_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libra_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ora_new.restype = C.c_void_p
        L.ora_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_n_servers.restype = C.c_uint32
        L.ora_n_servers.argtypes = [C.c_void_p]
        L.ora_set_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ora_get_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ora_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                               C.c_uint32, C.c_void_p]
        L.ora_step_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int,
                                        C.c_void_p]
        L.ora_max_threads.restype = C.c_int
        L.ora_set_max_runs.argtypes = [C.c_void_p, C.c_uint32]
        L.ora_agreed_commit.restype = C.c_uint64
        L.ora_agreed_commit.argtypes = [C.c_void_p, C.c_uint32]
        L.ora_server_checksum.restype = C.c_uint64
        L.ora_server_checksum.argtypes = [C.c_void_p]
        L.ora_struct_size.restype = C.c_size_t
        L.ora_struct_size.argtypes = [C.c_int]
        for i, dt in enumerate(abi.STRUCT_DTYPES):
            assert L.ora_struct_size(i) == dt.itemsize, (i, L.ora_struct_size(i), dt.itemsize)
        L.ora_adler32_update.restype = C.c_uint32
        L.ora_adler32_update.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.ora_wal_entry_checksum.restype = C.c_uint32
        L.ora_wal_entry_checksum.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


def adler32(data: bytes, start: int = 1) -> int:
    """RFC 1950 Adler-32 (wal_oracle.c)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    return int(lib().ora_adler32_update(start, buf.ctypes.data if len(buf) else None, len(buf)))


def wal_entry_checksums(entries: np.ndarray, data: np.ndarray) -> np.ndarray:
    """erlang:adler32([<<Idx:64, Term:64>> | EntryData]) per rgb_wal_entry (src/ra_log_wal.erl:528-534)."""
    L = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(len(entries), dtype=np.uint32)
    for i, e in enumerate(entries):
        off, ln = int(e["data_offset"]), int(e["data_len"])
        out[i] = L.ora_wal_entry_checksum(int(e["index"]), int(e["term"]),
                                          data.ctypes.data + off if ln else None, ln)
    return out


def agreed_commit(indexes) -> int:
    a = np.ascontiguousarray(indexes, dtype=np.uint64)
    return int(lib().ora_agreed_commit(a.ctypes.data, len(a)))


class Oracle:
    """Sequential CPU restatement of the reference transition over n_groups x n_members servers."""

    def __init__(self, n_groups: int, n_members: int, max_pipeline_count: int = 0,
                 max_aer_batch: int = 0, max_runs: int = 0):
        self._L = lib()
        self._h = self._L.ora_new(n_groups, n_members, max_pipeline_count, max_aer_batch)
        if not self._h:
            raise MemoryError("ora_new failed")
        if max_runs:      # model the engine's bounded term-run table (RGB_F_RUNS_OVERFLOW)
            self._L.ora_set_max_runs(self._h, max_runs)
        self.n_groups, self.n_members = n_groups, n_members
        self.n_servers = n_groups * n_members

    def close(self):
        if self._h:
            self._L.ora_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, first: int, states: np.ndarray):
        st = np.ascontiguousarray(states, dtype=abi.SERVER_STATE_DTYPE)
        rc = self._L.ora_set_state(self._h, first, len(st), st.ctypes.data)
        if rc:
            raise ValueError(f"ora_set_state rc={rc}")

    def get_state(self, first: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n_servers - first if n is None else n
        out = np.zeros(n, dtype=abi.SERVER_STATE_DTYPE)
        rc = self._L.ora_get_state(self._h, first, n, out.ctypes.data)
        if rc:
            raise ValueError(f"ora_get_state rc={rc}")
        return out

    def step(self, msgs: np.ndarray, rpc_cap: int | None = None):
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        n = len(m)
        dec = np.zeros(n, dtype=abi.DECISION_DTYPE)
        cap = n * abi.MAX_MEMBERS if rpc_cap is None else rpc_cap
        rpcs = np.zeros(max(cap, 1), dtype=abi.RPC_DTYPE)
        nr = C.c_uint32(0)
        self._L.ora_step(self._h, m.ctypes.data, n, dec.ctypes.data, rpcs.ctypes.data, cap,
                         C.byref(nr))
        return dec, rpcs[:min(nr.value, cap)]

    def step_parallel(self, msgs: np.ndarray, n_threads: int = 0):
        m = np.ascontiguousarray(msgs, dtype=abi.MSG_DTYPE)
        dec = np.zeros(len(m), dtype=abi.DECISION_DTYPE)
        nr = C.c_uint64(0)
        self._L.ora_step_parallel(self._h, m.ctypes.data, len(m), dec.ctypes.data, n_threads,
                                  C.byref(nr))
        return dec, nr.value

    def max_threads(self) -> int:
        return int(self._L.ora_max_threads())


def server_checksums(states: np.ndarray) -> np.ndarray:
    st = np.ascontiguousarray(states, dtype=abi.SERVER_STATE_DTYPE)
    L = lib()
    base = st.ctypes.data
    return np.array([L.ora_server_checksum(base + i * st.itemsize) for i in range(len(st))],
                    dtype=np.uint64)
