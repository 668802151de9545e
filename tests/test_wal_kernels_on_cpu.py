"""The WAL kernels without a GPU: ra_amd/csrc/rgb_wal.hip (Adler-32 and record framing kernels with their
staging and validation code) compiled as x86 C++ and executed by the fiber-per-lane block emulation of
tests/native (wal_on_cpu.cpp + kernel_on_cpu.cpp), against zlib / struct.pack and the reference's corruption
scenarios -- the same checks the -m gpu tests make on the real kernels."""
import os
import struct
import zlib

import numpy as np
import pytest

from ra_amd import abi
from oracle import oracle as O
from test_wal_framing import (make_batch as make_records, python_frame, random_specs, scanned_as_tuples,
                               phase_sweep_batch, check_phase_sweep, SWEEP_SMALL, SWEEP_MID, SWEEP_LARGE)
from test_wal_checksum import make_batch as make_entries, zlib_checksums

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class WalEmu:
    """The WAL entry points through ra_amd.engine (bound to the emulated library by the fixture)."""

    def __init__(self, engine):
        self.engine = engine
        self.eng = engine.RaGpuBatch(1, 1)

    def wal_adler32(self, entries, data):
        return self.eng.wal_adler32(entries, data)

    def wal_frame(self, records, data, out_bytes, flags=0):
        try:
            return 0, self.eng.wal_frame(records, data, out_bytes, flags)
        except self.engine.RgbError as e:
            return e.code, None

    def wal_layout(self, records, base=0):
        return self.engine.wal_layout(records, base)

    def wal_scan(self, f):
        return self.engine.wal_scan(f)[0]

    def wal_validate(self, f, scanned):
        return self.eng.wal_validate(np.frombuffer(bytes(f), dtype=np.uint8), scanned)


@pytest.fixture(scope="module")
def wal(emulated_engine):
    w = WalEmu(emulated_engine)
    yield w
    w.eng.close()


@pytest.mark.parametrize("small", [False, True], ids=["wave_per_entry", "four_per_wave"])
def test_checksum_kernel_matches_zlib(wal, small):
    rng = np.random.default_rng(50 + small)
    if small:
        lens = [0, 1, 15, 16, 17, 240, 255, 256, 257, 1023] + [int(x) for x in rng.integers(0, 700, size=150)]
    else:
        lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536, 70001] + \
               [int(x) for x in rng.integers(0, 20000, size=40)]
    entries, data = make_entries(rng, lens, True)
    assert (len(data) / len(lens) < 1024) == small
    got = wal.wal_adler32(entries, data)
    want = zlib_checksums(entries, data)
    bad = np.flatnonzero(got != want)
    assert len(bad) == 0, f"entry {bad[0]} len {lens[bad[0]]}: kernel {got[bad[0]]:#x} zlib {want[bad[0]]:#x}"


@pytest.mark.parametrize("small", [0, 1, 2], ids=["wave_per_record", "four_per_wave", "eight_per_wave"])
@pytest.mark.parametrize("flags", [0, abi.WAL_NO_CHECKSUMS])
def test_framing_kernel_matches_struct_pack(wal, small, flags):
    rng = np.random.default_rng(60 + small)
    if small == 2:
        lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 255, 256, 257] + [int(x) for x in rng.integers(0, 600, size=150)]
    elif small == 1:
        lens = [0, 1, 16, 255, 256, 257, 511, 512, 513, 1023] + [int(x) for x in rng.integers(300, 1000, size=80)]
    else:
        lens = [0, 1, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536, 70001] + \
               [int(x) for x in rng.integers(0, 20000, size=40)]
    specs = random_specs(rng, len(lens), lens, n_writers=7)
    recs, data, payloads = make_records(rng, specs)
    total = wal.wal_layout(recs, 0)
    assert (len(data) / len(lens) < 1024) == bool(small) and (len(data) / len(lens) <= 320) == (small == 2)
    rc, out = wal.wal_frame(recs, data, total, flags)
    assert rc == 0
    want = python_frame(specs, payloads, not flags)
    assert len(want) == total
    bad = np.flatnonzero(out != np.frombuffer(want, dtype=np.uint8))
    assert len(bad) == 0, f"first differing output byte {bad[0]} of {total}"


@pytest.mark.parametrize("small", [False, True], ids=["wave_per_record", "four_per_wave"])
def test_framing_kernel_with_header_data_longer_than_a_lane_group(wal, small):
    """A new writer's record carries its uid (src/ra_log_wal.erl:520-526): 200-byte uids make HeaderData longer than
    the 16 / 64 lanes that copy it, next to payloads shorter than one 16-byte chunk (head and tail in one chunk)."""
    rng = np.random.default_rng(80 + small)
    lens = ([0, 1, 2, 3, 5, 7, 11, 13, 15, 16, 17, 40, 100, 300] if small else [0, 1, 5, 15, 2000, 4096, 9000, 33000])
    uids = [bytes(rng.integers(97, 123, size=200, dtype=np.uint8)) for _ in lens]
    specs = [(i & 1, i, uids[i], 10 + i, 3, ln) for i, ln in enumerate(lens)]
    recs, data, payloads = make_records(rng, specs)
    assert int(recs["hdr_len"].max()) >= 202 and (len(data) / len(lens) < 1024) == small
    for base in (0, 5, 15):
        total = wal.wal_layout(recs, base)
        rc, out = wal.wal_frame(recs, data, total)
        assert rc == 0 and out[base:].tobytes() == python_frame(specs, payloads)


def test_recovery_scenarios_of_the_reference(wal):
    """test/ra_log_wal_SUITE.erl:1439-1528 on 100 entries of 1006 bytes, framed by the kernel."""
    rng = np.random.default_rng(70)
    uid = b"recover_with_last_entry_corruption_pre_allocate"
    specs = [(int(i == 0), 0, uid if i == 0 else None, i + 1, 1, 1006) for i in range(100)]
    recs, data, payloads = make_records(rng, specs)
    total = wal.wal_layout(recs, 0)
    rc, body = wal.wal_frame(recs, data, total)
    assert rc == 0 and body.tobytes() == python_frame(specs, payloads)
    clean = abi.WAL_FILE_HEADER + body.tobytes()
    assert len(clean) == 103354

    def recover(f):
        scanned = wal.wal_scan(f)
        n_ok, status = wal.wal_validate(f, scanned)
        want, outcome = O.wal_recover_records(f)
        assert scanned_as_tuples(f, scanned[:n_ok]) == want
        return n_ok, status, outcome

    assert recover(clean) == (100, abi.WAL_CLEAN, "eof")
    f = bytearray(clean); f[-10:] = bytes(10)
    assert recover(bytes(f)) == (99, abi.WAL_DROPPED_LAST, "dropped_last")
    f = bytearray(clean + bytes(4096)); f[103331:103341] = bytes(10)
    assert recover(bytes(f)) == (99, abi.WAL_DROPPED_LAST, "dropped_last")
    f = bytearray(clean); f[1000:1010] = bytes(10)
    assert recover(bytes(f)) == (0, abi.WAL_CORRUPT, "corrupt")
    bad = recs.copy(); bad["out_offset"][5] = bad["out_offset"][4]
    assert wal.wal_frame(bad, data, total)[0] == abi.E_INVAL          # overlapping records


def test_descriptors_whose_offset_plus_length_wraps_are_refused(wal):
    """Descriptors come from the caller (a NIF): `offset + length` is checked without the u64 wrap-around of the
    sum, so a wrapped slice is RGB_E_INVAL and nothing is read or written out of bounds."""
    rng = np.random.default_rng(71)
    specs = [(0, 0, None, i + 1, 1, 64) for i in range(4)]
    recs, data, _ = make_records(rng, specs)
    total = wal.wal_layout(recs, 0)
    assert wal.wal_frame(recs, data, total)[0] == 0
    for field, value in (("data_offset", 2 ** 64 - 8), ("hdr_offset", 2 ** 64 - 1), ("out_offset", 2 ** 64 - 16)):
        bad = recs.copy(); bad[field][3] = value
        assert wal.wal_frame(bad, data, total)[0] == abi.E_INVAL, field
    entries, edata = make_entries(rng, [16, 32, 48])
    bad = entries.copy(); bad["data_offset"][2] = 2 ** 64 - 4
    with pytest.raises(wal.engine.RgbError) as e:
        wal.wal_adler32(bad, edata)
    assert e.value.code == abi.E_INVAL
    # a scanned record whose `Rest` offset points past the file
    uid = b"writer"
    specs = [(int(i == 0), 0, uid if i == 0 else None, i + 1, 1, 100) for i in range(3)]
    recs, data, _ = make_records(rng, specs)
    total = wal.wal_layout(recs, 0)
    body = wal.wal_frame(recs, data, total)[1]
    f = bytearray(abi.WAL_FILE_HEADER + body.tobytes()); f[-5] ^= 0xFF       # the last record fails its checksum
    scanned = wal.wal_scan(bytes(f))
    assert wal.wal_validate(bytes(f), scanned) == (2, abi.WAL_DROPPED_LAST)
    scanned["next_offset"][2] = len(f) + 1
    with pytest.raises(wal.engine.RgbError) as e:
        wal.wal_validate(bytes(f), scanned)
    assert e.value.code == abi.E_INVAL


@pytest.mark.parametrize("small", [0, 1, 2], ids=["wave_per_record", "four_per_wave", "eight_per_wave"])
def test_framing_kernel_at_every_source_and_destination_phase(wal, small):
    """The -m gpu test of the same name on the emulated kernel (fewer phases for the larger records: the fiber
    emulation runs them at a few MB/s)."""
    rng = np.random.default_rng(95 + small)
    if small == 2:
        recs, data, out_bytes, want = phase_sweep_batch(rng, SWEEP_SMALL)
    elif small == 1:
        recs, data, out_bytes, want = phase_sweep_batch(rng, SWEEP_MID, (0, 1, 5, 9, 15), (0, 3, 5, 8, 13))
    else:
        recs, data, out_bytes, want = phase_sweep_batch(rng, SWEEP_LARGE, (0, 5, 15), (0, 3, 8, 13))
    assert (len(data) / len(recs) < 1024) == bool(small) and (len(data) / len(recs) <= 320) == (small == 2)
    rc, out = wal.wal_frame(recs, data, out_bytes)
    assert rc == 0
    check_phase_sweep(out.tobytes(), want, out_bytes)
