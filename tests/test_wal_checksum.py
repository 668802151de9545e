"""WAL entry checksums (include/ra_gpu_wal.h): the CPU checker against
the published Adler-32 vectors and zlib, the HIP kernel against both."""
import struct
import zlib

import numpy as np
import pytest

from ra_amd import abi
from oracle import oracle as O


def frame(index, term, payload: bytes) -> bytes:
    """Entry = [<<Idx:64/unsigned, Term:64/unsigned>> | EntryData]  (src/ra_log_wal.erl:528-530)"""
    return struct.pack(">QQ", index, term) + payload


def _engine():
    import os
    from ra_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    return engine


def make_batch(rng, lens, misalign=True):
    """Payloads packed back to back (arbitrary alignment) into one buffer, 16 spare bytes behind."""
    entries = np.zeros(len(lens), dtype=abi.WAL_ENTRY_DTYPE)
    off = int(rng.integers(0, 16)) if misalign else 0
    chunks = []
    pos = 0
    for i, ln in enumerate(lens):
        pad = int(rng.integers(0, 5)) if misalign else (-pos) % 16
        chunks.append(bytes(pad)); pos += pad
        entries["index"][i] = int(rng.integers(0, 1 << 62))
        entries["term"][i] = int(rng.integers(0, 1 << 40))
        entries["data_offset"][i] = off + pos
        entries["data_len"][i] = ln
        chunks.append(rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()); pos += ln
    data = np.frombuffer(bytes(off) + b"".join(chunks) + bytes(32), dtype=np.uint8).copy()
    return entries, data


def zlib_checksums(entries, data):
    out = np.zeros(len(entries), dtype=np.uint32)
    for i, e in enumerate(entries):
        o, n = int(e["data_offset"]), int(e["data_len"])
        out[i] = zlib.adler32(frame(int(e["index"]), int(e["term"]), data[o:o + n].tobytes()))
    return out


def test_oracle_adler32_known_answers():
    assert O.adler32(b"") == 1                                   # RFC 1950: s1 = 1, s2 = 0
    assert O.adler32(b"Wikipedia") == 0x11E60398                 # the textbook vector
    assert O.adler32(b"a") == 0x00620062
    assert O.adler32(bytes(range(256)) * 300) == zlib.adler32(bytes(range(256)) * 300)  # wraps 65521 often


def test_oracle_wal_entry_checksum_matches_zlib():
    rng = np.random.default_rng(7)
    lens = [0, 1, 15, 16, 17, 255, 4096, 70001] + [int(x) for x in rng.integers(0, 3000, size=40)]
    entries, data = make_batch(rng, lens)
    assert np.array_equal(O.wal_entry_checksums(entries, data), zlib_checksums(entries, data))
    # the reference's validation is the same function on the read path (src/ra_log_wal.erl:861)
    e0 = entries[3]
    payload = data[int(e0["data_offset"]):int(e0["data_offset"]) + int(e0["data_len"])].tobytes()
    assert O.wal_entry_checksums(entries[3:4], data)[0] == zlib.adler32(frame(int(e0["index"]), int(e0["term"]), payload))


def test_oracle_detects_overwritten_bytes():
    """test/ra_log_wal_SUITE.erl:1500-1528 checksum_failure_in_middle_of_file_should_fail: 1000-byte
    entries, ten bytes of one record overwritten with zeros -> validation must fail for that record
    (and only for it)."""
    rng = np.random.default_rng(3)
    entries, data = make_batch(rng, [1000] * 100, misalign=False)
    entries["term"] = 1
    entries["index"] = np.arange(1, 101)
    stored = O.wal_entry_checksums(entries, data)
    victim = 42
    o = int(entries["data_offset"][victim])
    data[o + 500:o + 510] = np.where(data[o + 500:o + 510] == 0, 1, 0)     # guaranteed to differ
    again = O.wal_entry_checksums(entries, data)
    assert again[victim] != stored[victim]
    assert np.array_equal(np.delete(again, victim), np.delete(stored, victim))


@pytest.mark.gpu
@pytest.mark.parametrize("misalign", [True, False])
def test_gpu_wal_checksums_match_oracle_and_zlib(misalign):
    import torch
    engine = _engine()
    rng = np.random.default_rng(11 + misalign)
    lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536, 70001,
            1 << 20] + [int(x) for x in rng.integers(0, 20000, size=300)]
    entries, data = make_batch(rng, lens, misalign)
    eng = engine.RaGpuBatch(1, 1)
    d_e = torch.from_numpy(entries.view(np.uint8)).cuda()
    d_d = torch.from_numpy(data).cuda()
    d_o = torch.zeros(len(entries), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                  # the library launches on its own stream
    eng.wal_adler32_device(d_e.data_ptr(), len(entries), d_d.data_ptr(), len(data), d_o.data_ptr())
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint32)
    want = O.wal_entry_checksums(entries, data)
    assert np.array_equal(want, zlib_checksums(entries, data))
    bad = np.flatnonzero(got != want)
    assert len(bad) == 0, f"entry {bad[0]} len {lens[bad[0]]}: gpu {got[bad[0]]:#x} oracle {want[bad[0]]:#x}"
    eng.close()


@pytest.mark.gpu
def test_gpu_wal_host_buffer_form():
    """rgb_wal_adler32: host buffers in, host checksums out (what the NIF binds); grows its staging."""
    engine = _engine()
    rng = np.random.default_rng(31)
    eng = engine.RaGpuBatch(1, 1)
    for lens in ([0, 5, 100, 4096], [int(x) for x in rng.integers(0, 9000, size=500)], [1 << 20, 3]):
        entries, data = make_batch(rng, lens, True)
        assert np.array_equal(eng.wal_adler32(entries, data), zlib_checksums(entries, data))
    bad = np.zeros(1, dtype=abi.WAL_ENTRY_DTYPE)
    bad["data_offset"] = 10; bad["data_len"] = 100
    with pytest.raises(engine.RgbError):
        eng.wal_adler32(bad, np.zeros(50, dtype=np.uint8))          # payload outside the buffer
    eng.close()


@pytest.mark.gpu
def test_gpu_wal_checksums_small_entries():
    """Mean payload below 1 KiB: the four-entries-per-wavefront variant."""
    import torch
    engine = _engine()
    rng = np.random.default_rng(23)
    lens = [0, 1, 15, 16, 17, 240, 255, 256, 257, 1023] + [int(x) for x in rng.integers(0, 700, size=1013)]
    entries, data = make_batch(rng, lens, True)
    assert len(data) / len(lens) < 1024
    eng = engine.RaGpuBatch(1, 1)
    d_e = torch.from_numpy(entries.view(np.uint8)).cuda()
    d_d = torch.from_numpy(data).cuda()
    d_o = torch.zeros(len(entries), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.wal_adler32_device(d_e.data_ptr(), len(entries), d_d.data_ptr(), len(data), d_o.data_ptr())
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint32)
    want = zlib_checksums(entries, data)
    bad = np.flatnonzero(got != want)
    assert len(bad) == 0, f"entry {bad[0]} len {lens[bad[0]]}: gpu {got[bad[0]]:#x} zlib {want[bad[0]]:#x}"
    eng.close()


@pytest.mark.gpu
def test_gpu_wal_checksums_full_size_properties():
    """4 GiB-class batches cannot be checked byte by byte on the CPU in seconds: check the property
    Adler-32 offers -- the checksum of a record is a function of (A, B) sums that are additive, so
    flipping one byte by +1 at position p from the end changes A by 1 and B by p (mod 65521)."""
    import torch
    engine = _engine()
    rng = np.random.default_rng(5)
    n, ln = 4096, 65536
    entries = np.zeros(n, dtype=abi.WAL_ENTRY_DTYPE)
    entries["index"] = np.arange(n); entries["term"] = 7
    entries["data_offset"] = np.arange(n, dtype=np.uint64) * ln
    entries["data_len"] = ln
    data = rng.integers(0, 255, size=n * ln + 16, dtype=np.uint8)      # < 255 so +1 never wraps a byte
    eng = engine.RaGpuBatch(1, 1)
    d_e = torch.from_numpy(entries.view(np.uint8)).cuda()
    d_d = torch.from_numpy(data).cuda()
    d_o = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                  # the library launches on its own stream
    eng.wal_adler32_device(d_e.data_ptr(), n, d_d.data_ptr(), len(data), d_o.data_ptr())
    torch.cuda.synchronize()
    before = d_o.cpu().numpy().view(np.uint32).copy()
    # spot-check a sample against the CPU
    sample = rng.choice(n, size=24, replace=False)
    assert np.array_equal(before[sample], O.wal_entry_checksums(entries[sample], data))
    pos = rng.integers(0, ln, size=n)                                    # byte position inside each record
    flat = torch.from_numpy((np.arange(n, dtype=np.int64) * ln + pos)).cuda()
    d_d[flat] += 1
    torch.cuda.synchronize()
    eng.wal_adler32_device(d_e.data_ptr(), n, d_d.data_ptr(), len(data), d_o.data_ptr())
    torch.cuda.synchronize()
    after = d_o.cpu().numpy().view(np.uint32)
    a0, b0 = before & 0xFFFF, before >> 16
    a1, b1 = after & 0xFFFF, after >> 16
    assert np.array_equal((a0.astype(np.int64) + 1) % 65521, a1)
    assert np.array_equal((b0.astype(np.int64) + (ln - pos)) % 65521, b1)
    eng.close()
