"""WAL record framing (write path) and recovery validation (read path) -- include/ra_gpu_wal.h,
reference src/ra_log_wal.erl:482-537 (serialize_header, write_data) and :877-1033 (recover_records,
is_last_record, validate_checksum); scenarios of test/ra_log_wal_SUITE.erl:1439-1528."""
import struct
import zlib

import numpy as np
import pytest

from ra_amd import abi, engine
from oracle import oracle as O


def header_bytes(trunc: int, id_ref: int, uid: bytes | None) -> bytes:
    """serialize_header/3 (src/ra_log_wal.erl:482-499): uid given = first appearance in this file."""
    if uid is None:
        return ((trunc << 23) | (1 << 22) | id_ref).to_bytes(3, "big")
    return ((trunc << 23) | id_ref).to_bytes(3, "big") + struct.pack(">H", len(uid)) + uid


def make_batch(rng, specs, misalign=True):
    """specs: [(trunc, id_ref, uid-or-None, index, term, payload_len)] -> (records, data, payloads).
    HeaderData and payload bytes are packed into one data buffer at arbitrary alignment."""
    recs = np.zeros(len(specs), dtype=abi.WAL_RECORD_DTYPE)
    chunks, pos, payloads = [], 0, []
    for i, (trunc, id_ref, uid, idx, term, ln) in enumerate(specs):
        h = header_bytes(trunc, id_ref, uid)
        pad = int(rng.integers(0, 7)) if misalign else 0
        chunks.append(bytes(pad)); pos += pad
        recs["hdr_offset"][i], recs["hdr_len"][i] = pos, len(h)
        chunks.append(h); pos += len(h)
        pad = int(rng.integers(0, 7)) if misalign else 0
        chunks.append(bytes(pad)); pos += pad
        payload = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        payloads.append(payload)
        recs["index"][i], recs["term"][i] = idx, term
        recs["data_offset"][i], recs["data_len"][i] = pos, ln
        chunks.append(payload); pos += ln
    data = np.frombuffer(b"".join(chunks) + bytes(16), dtype=np.uint8).copy()
    return recs, data, payloads


def python_frame(specs, payloads, compute_checksums=True) -> bytes:
    """The record bytes written independently of the checker: struct.pack + zlib."""
    out = []
    for (trunc, id_ref, uid, idx, term, ln), payload in zip(specs, payloads):
        entry = struct.pack(">QQ", idx, term) + payload
        cs = zlib.adler32(entry) if compute_checksums else 0
        out.append(header_bytes(trunc, id_ref, uid) + struct.pack(">II", cs, ln) + entry)
    return b"".join(out)


def random_specs(rng, n, lens, n_writers=5):
    uids = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 60)), dtype=np.uint8)) for _ in range(n_writers)]
    seen, nxt, specs = set(), {}, []
    for i in range(n):
        w = int(rng.integers(0, n_writers))
        idx = nxt.get(w, int(rng.integers(1, 1 << 40)))
        nxt[w] = idx + 1
        first = w not in seen
        seen.add(w)
        specs.append((int(rng.integers(0, 2)), w, uids[w] if first else None, idx, int(rng.integers(1, 1 << 30)),
                      int(lens[i])))
    return specs


def phase_sweep_batch(rng, lens, src_phases=range(16), dst_phases=range(16)):
    """Every payload length of `lens` at every source phase x destination phase (address mod 16, for 16-byte aligned
    buffers): records with gaps between them in the output so that each starts at the phase wanted.  Returns
    (records with out_offset set, data, out_bytes, [(out_offset, record bytes)])."""
    hdr = header_bytes(0, 3, None)
    specs = [(ln, sp, dp) for ln in lens for sp in src_phases for dp in dst_phases]
    recs = np.zeros(len(specs), dtype=abi.WAL_RECORD_DTYPE)
    chunks, pos, want, out = [hdr + bytes(13)], 16, [], 0
    for i, (ln, sp, dp) in enumerate(specs):
        pad = (sp - pos) % 16
        chunks.append(bytes(pad)); pos += pad
        payload = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        recs["index"][i], recs["term"][i] = i + 1, 7
        recs["hdr_offset"][i], recs["hdr_len"][i] = 0, 3
        recs["data_offset"][i], recs["data_len"][i] = pos, ln
        chunks.append(payload); pos += ln
        out += (dp - out) % 16
        recs["out_offset"][i] = out
        entry = struct.pack(">QQ", i + 1, 7) + payload
        want.append((out, hdr + struct.pack(">II", zlib.adler32(entry), ln) + entry))
        out += 27 + ln
    data = np.frombuffer(b"".join(chunks) + bytes(16), dtype=np.uint8).copy()
    return recs, data, out, want


def check_phase_sweep(framed: bytes, want, out_bytes):
    """Every record where it belongs, and nothing written in the gaps (the host forms zero the output first)."""
    ref = bytearray(out_bytes)
    for off, rec in want:
        ref[off:off + len(rec)] = rec
    if framed != bytes(ref):
        got = np.frombuffer(framed, dtype=np.uint8); exp = np.frombuffer(bytes(ref), dtype=np.uint8)
        bad = int(np.flatnonzero(got != exp)[0])
        k = max(i for i, (off, _) in enumerate(want) if off <= bad) if bad >= want[0][0] else -1
        raise AssertionError(f"first difference at output byte {bad} (record {k}, starts at {want[k][0]}, "
                             f"{len(want[k][1])} bytes)")


SWEEP_SMALL = [0, 1, 2, 3, 5, 8, 13, 15, 16, 17, 31, 32, 33, 47, 48, 49, 240, 255, 256, 257, 272]
SWEEP_MID = [321, 400, 496, 511, 512, 513, 527, 528, 529, 767, 768, 769, 1000]     # sixteen lanes per record
SWEEP_LARGE = [1007, 1008, 1023, 1024, 1025, 1040, 2047, 2048, 2049, 4096]


# ------------------------------------------------------------------------------------------ CPU

def test_oracle_frame_matches_struct_pack_and_zlib():
    rng = np.random.default_rng(1)
    lens = [0, 1, 15, 16, 17, 1000, 4096] + [int(x) for x in rng.integers(0, 3000, size=60)]
    specs = random_specs(rng, len(lens), lens)
    recs, data, payloads = make_batch(rng, specs)
    total = engine.wal_layout(recs, 0)
    want = python_frame(specs, payloads)
    assert total == len(want)
    assert O.wal_frame(recs, data, total).tobytes() == want
    assert O.wal_frame(recs, data, total, compute_checksums=False).tobytes() == python_frame(specs, payloads, False)


def test_layout_reproduces_the_file_offset_the_reference_test_hard_codes():
    """test/ra_log_wal_SUITE.erl:1469-1498 writes 100 entries of 1000 random bytes (term_to_iovec of a
    1000-byte binary = 1006 bytes) as writer <<"recover_with_last_entry_corruption_pre_allocate">> and
    then pokes file offset 103331 "if the internal WAL format changes this will be wrong": with
    DataSize = HeaderLen + 24 + EntryDataLen (src/ra_log_wal.erl:526) behind the 5-byte file header
    that offset must lie inside the last record's payload."""
    uid = b"recover_with_last_entry_corruption_pre_allocate"
    assert len(uid) == 47
    recs = np.zeros(100, dtype=abi.WAL_RECORD_DTYPE)
    recs["data_len"] = 1006
    recs["hdr_len"] = 3
    recs["hdr_len"][0] = 5 + len(uid)
    end = engine.wal_layout(recs, 5)
    assert end == 5 + (5 + 47 + 24 + 1006) + 99 * (3 + 24 + 1006) == 103354
    last_payload = int(recs["out_offset"][99]) + 3 + 24
    assert last_payload <= 103331 and 103331 + 10 <= end


def build_file(rng, n=40, tail=b"", lens=None, n_writers=4):
    lens = [int(x) for x in rng.integers(0, 700, size=n)] if lens is None else lens
    specs = random_specs(rng, len(lens), lens, n_writers)
    recs, data, payloads = make_batch(rng, specs)
    total = engine.wal_layout(recs, 0)
    body = O.wal_frame(recs, data, total).tobytes()
    return abi.WAL_FILE_HEADER + body + tail, specs, payloads


def scanned_as_tuples(file_bytes, scanned):
    out, names = [], {}
    for r in scanned:
        if int(r["flags"]) & abi.WAL_REC_FIRST:
            names[int(r["id_ref"])] = file_bytes[int(r["uid_offset"]):int(r["uid_offset"]) + int(r["uid_len"])]
        if int(r["flags"]) & abi.WAL_REC_UNKNOWN:
            continue
        o, n = int(r["data_offset"]), int(r["data_len"])
        out.append((names[int(r["id_ref"])], int(r["trunc"]), int(r["index"]), int(r["term"]), file_bytes[o:o + n]))
    return out


@pytest.mark.parametrize("tail", [b"", bytes(64), bytes(5), b"\x40\x00"], ids=["eof", "zeros", "short_zeros", "partial"])
def test_scan_walks_the_file_like_recover_records(tail):
    rng = np.random.default_rng(2)
    f, specs, payloads = build_file(rng, tail=tail)
    scanned, consumed, end = engine.wal_scan(f)
    want, outcome = O.wal_recover_records(f)
    assert outcome == ("zeros" if tail == bytes(64) else "eof")
    assert end == (abi.WAL_END_ZEROS if tail == bytes(64) else abi.WAL_END_DATA)
    assert consumed == len(f) - len(tail)
    assert scanned_as_tuples(f, scanned) == want
    assert len(want) == len(specs) and [w[4] for w in want] == payloads
    assert all(int(r["checksum"]) == zlib.adler32(struct.pack(">QQ", int(r["index"]), int(r["term"])) + p)
               for r, p in zip(scanned, payloads))


def test_scan_skips_records_of_writers_never_named_in_this_file():
    """A short header whose IdRef has no long header before it refers to a deleted UId (:968-971)."""
    rng = np.random.default_rng(3)
    specs = [(0, 0, b"w0", 1, 1, 10), (0, 7, None, 5, 1, 20), (1, 0, None, 2, 1, 30)]
    recs, data, payloads = make_batch(rng, specs)
    total = engine.wal_layout(recs, 0)
    f = abi.WAL_FILE_HEADER + O.wal_frame(recs, data, total).tobytes()
    scanned, _, _ = engine.wal_scan(f)
    assert [int(x) for x in scanned["flags"]] == [abi.WAL_REC_FIRST | abi.WAL_REC_VALIDATE, abi.WAL_REC_UNKNOWN,
                                                  abi.WAL_REC_VALIDATE]
    want, _ = O.wal_recover_records(f)
    assert scanned_as_tuples(f, scanned) == want == [(b"w0", 0, 1, 1, payloads[0]), (b"w0", 1, 2, 1, payloads[2])]


def test_scan_rejects_an_unknown_file_header_and_honours_cap():
    with pytest.raises(engine.RgbError):
        engine.wal_scan(b"RAWA\x02" + bytes(40))                    # exit({unknown_wal_file_format, ...})
    with pytest.raises(engine.RgbError):
        engine.wal_scan(b"RAW")
    rng = np.random.default_rng(4)
    f, specs, _ = build_file(rng, n=10)
    part, consumed, end = engine.wal_scan(f, cap=4)
    assert len(part) == 4 and end == abi.WAL_END_CAP and consumed == int(part["next_offset"][3])


def test_scan_survives_damaged_files_and_agrees_with_the_restatement():
    """recover_records/5 parses whatever is on disk: bit flips in headers and lengths, truncation at any
    byte, garbage tails.  The C walk must never read outside the buffer and must find exactly the records
    the clause-by-clause restatement finds."""
    rng = np.random.default_rng(77)
    clean, _, _ = build_file(rng, n=25, n_writers=3)
    cases = 0
    for trial in range(400):
        f = bytearray(clean)
        r = rng.random()
        if r < 0.35:
            for _ in range(int(rng.integers(1, 4))):
                f[int(rng.integers(5, len(f)))] ^= 1 << int(rng.integers(0, 8))
        elif r < 0.7:
            del f[int(rng.integers(5, len(f))):]
        elif r < 0.85:
            f += bytes(rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8))
        else:
            cut = int(rng.integers(5, len(f)))
            f = f[:cut] + bytes(int(rng.integers(0, 64))) + f[cut:]
        f = bytes(f)
        # a copy with nothing behind it: an out-of-bounds read would land in unmapped or foreign memory
        buf = np.frombuffer(f, dtype=np.uint8).copy()
        scanned, consumed, end = engine.wal_scan(buf)
        want, reason, want_consumed = O.wal_scan_records(f)
        assert len(scanned) == len(want), trial
        assert consumed == want_consumed and end == (abi.WAL_END_ZEROS if reason == "zeros" else abi.WAL_END_DATA), trial
        for rec, w in zip(scanned, want):
            first, trunc, id_ref, uoff, ulen, cs, idx, term, doff, dlen = w
            assert bool(int(rec["flags"]) & abi.WAL_REC_FIRST) == first and int(rec["trunc"]) == trunc, trial
            assert (int(rec["id_ref"]), int(rec["checksum"]), int(rec["index"]), int(rec["term"])) == (id_ref, cs, idx, term)
            assert (int(rec["data_offset"]), int(rec["data_len"]), int(rec["uid_len"])) == (doff, dlen, ulen), trial
            assert int(rec["data_offset"]) + int(rec["data_len"]) <= len(f)
            if first:
                assert int(rec["uid_offset"]) == uoff
        cases += len(want) != 25
    assert cases > 100


def test_scan_under_sanitizers(tmp_path):
    """The same walk compiled with AddressSanitizer + UBSan (host-only translation unit, plain g++): 300
    damaged files, each in an exactly-sized heap block; any out-of-bounds read or overflow aborts."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "wal_scan_harness"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(root, "include"), "-o", str(exe),
           os.path.join(root, "tests", "native", "wal_scan_harness.cpp"),
           os.path.join(root, "ra_amd", "csrc", "rgb_wal_host.cpp")]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert built.returncode == 0, built.stderr
    rng = np.random.default_rng(78)
    clean, _, _ = build_file(rng, n=30, n_writers=3)
    files, want = [], []
    for trial in range(300):
        f = bytearray(clean)
        r = rng.random()
        if r < 0.4:
            for _ in range(int(rng.integers(1, 6))):
                f[int(rng.integers(5, len(f)))] = int(rng.integers(0, 256))
        elif r < 0.8:
            del f[int(rng.integers(5, len(f))):]
        else:
            f = f[:int(rng.integers(5, len(f)))] + bytes(rng.integers(0, 256, size=int(rng.integers(0, 80)), dtype=np.uint8))
        path = tmp_path / f"w{trial}.wal"
        path.write_bytes(bytes(f))
        files.append(str(path))
        recs, reason, consumed = O.wal_scan_records(bytes(f))
        want.append((0, len(recs), consumed, abi.WAL_END_ZEROS if reason == "zeros" else abi.WAL_END_DATA))
    run = subprocess.run([str(exe)] + files, capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-2000:]
    got = [tuple(int(x) for x in line.split()) for line in run.stdout.splitlines()]
    assert got == want


# ------------------------------------------------------------------------------------------ GPU

def _open():
    import os
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    return engine.RaGpuBatch(1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("small", [0, 1, 2], ids=["wave_per_record", "four_per_wave", "eight_per_wave"])
@pytest.mark.parametrize("flags", [0, abi.WAL_NO_CHECKSUMS])
def test_gpu_frame_matches_oracle_bytes(small, flags):
    import torch
    rng = np.random.default_rng(20 + small)
    if small == 2:
        lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 255, 256, 257] + [int(x) for x in rng.integers(0, 600, size=1000)]
    elif small == 1:
        lens = [0, 1, 16, 255, 256, 257, 511, 512, 513, 1023] + [int(x) for x in rng.integers(300, 1000, size=1000)]
    else:
        lens = [0, 1, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536, 70001, 1 << 20] + \
               [int(x) for x in rng.integers(0, 20000, size=300)]
    specs = random_specs(rng, len(lens), lens, n_writers=9)
    recs, data, payloads = make_batch(rng, specs)
    base = int(rng.integers(0, 16))                                 # the batch continues a file at any offset
    total = engine.wal_layout(recs, base)
    assert (len(data) / len(lens) < 1024) == bool(small) and (len(data) / len(lens) <= 320) == (small == 2)
    want = O.wal_frame(recs, data, total, compute_checksums=not flags)
    assert want[base:].tobytes() == python_frame(specs, payloads, not flags)
    eng = _open()
    d_r = torch.from_numpy(recs.view(np.uint8)).cuda()
    d_d = torch.from_numpy(data).cuda()
    d_o = torch.full((total + 64,), 0xEE, dtype=torch.uint8, device="cuda")      # canary around the records
    d_c = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.wal_frame_device(d_r.data_ptr(), len(recs), d_d.data_ptr(), len(data), d_o.data_ptr(), total + 64,
                         d_c.data_ptr(), flags)
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    assert np.all(got[:base] == 0xEE) and np.all(got[total:] == 0xEE), "wrote outside the records"
    bad = np.flatnonzero(got[base:total] != want[base:])
    assert len(bad) == 0, f"first differing output byte {bad[0] + base} of {total}"
    sums = d_c.cpu().numpy().view(np.uint32)
    want_sums = np.array([0 if flags else zlib.adler32(struct.pack(">QQ", s[3], s[4]) + p)
                          for s, p in zip(specs, payloads)], dtype=np.uint32)
    assert np.array_equal(sums, want_sums)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("small", [False, True], ids=["wave_per_record", "four_per_wave"])
def test_gpu_frame_unaligned_device_pointers_and_long_header_data(small):
    """The kernel reads the payload in chunks aligned to the source ADDRESS and writes chunks aligned to the
    destination ADDRESS: device pointers at every phase mod 16, 200-byte uids (HeaderData longer than the lane
    group that copies it), payloads shorter than a chunk."""
    import torch
    rng = np.random.default_rng(90 + small)
    lens = ([0, 1, 2, 3, 5, 7, 11, 13, 15, 16, 17, 40, 100, 300] + [int(x) for x in rng.integers(0, 400, size=200)]) if small \
        else ([0, 1, 5, 15, 2000, 4096, 9000, 33000] + [int(x) for x in rng.integers(1000, 9000, size=60)])
    uids = [bytes(rng.integers(97, 123, size=200, dtype=np.uint8)) for _ in lens]
    specs = [(i & 1, i % 1000, uids[i] if i < 12 else None, 10 + i, 3, ln) for i, ln in enumerate(lens)]
    recs, data, payloads = make_batch(rng, specs)
    assert (len(data) / len(lens) < 1024) == small
    want = python_frame(specs, payloads)
    eng = _open()
    total = engine.wal_layout(recs, 0)                  # fills out_offset: before the descriptors go to the device
    d_r = torch.from_numpy(recs.view(np.uint8)).cuda()
    for skew_d, skew_o in ((0, 0), (1, 0), (0, 1), (7, 13), (15, 3), (9, 9)):
        d_d = torch.zeros(len(data) + 32, dtype=torch.uint8, device="cuda")
        d_d[skew_d:skew_d + len(data)] = torch.from_numpy(data).cuda()
        d_o = torch.full((total + 96,), 0xEE, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eng.wal_frame_device(d_r.data_ptr(), len(recs), d_d.data_ptr() + skew_d, len(data), d_o.data_ptr() + 32 + skew_o,
                             total, 0, 0)
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()
        lo = 32 + skew_o
        assert np.all(got[:lo] == 0xEE) and np.all(got[lo + total:] == 0xEE), "wrote outside the records"
        assert got[lo:lo + total].tobytes() == want, (skew_d, skew_o)
    eng.close()


@pytest.mark.gpu
def test_gpu_frame_host_form_then_recovery_round_trip():
    """frame -> file -> scan -> validate: clean; then the three corruptions of
    test/ra_log_wal_SUITE.erl:1439-1528 on 100 entries of 1006 bytes."""
    rng = np.random.default_rng(30)
    eng = _open()
    uid = b"recover_with_last_entry_corruption_pre_allocate"
    specs = [(int(i == 0), 0, uid if i == 0 else None, i + 1, 1, 1006) for i in range(100)]
    recs, data, payloads = make_batch(rng, specs)
    total = engine.wal_layout(recs, 0)
    body = eng.wal_frame(recs, data, total)
    assert body.tobytes() == python_frame(specs, payloads)
    clean = abi.WAL_FILE_HEADER + body.tobytes()
    assert len(clean) == 103354

    def recover(file_bytes):
        scanned, _, _ = engine.wal_scan(file_bytes)
        n_ok, status = eng.wal_validate(np.frombuffer(file_bytes, dtype=np.uint8), scanned)
        want, outcome = O.wal_recover_records(file_bytes)
        assert scanned_as_tuples(file_bytes, scanned[:n_ok]) == want
        return n_ok, status, outcome

    assert recover(clean) == (100, abi.WAL_CLEAN, "eof")
    # recover_with_last_entry_corruption: the last ten bytes of the file zeroed -> recovery resumes
    f = bytearray(clean); f[-10:] = bytes(10)
    assert recover(bytes(f)) == (99, abi.WAL_DROPPED_LAST, "dropped_last")
    # ..._pre_allocate: zeros behind the data, ten bytes zeroed at offset 103331
    f = bytearray(clean + bytes(4096)); f[103331:103341] = bytes(10)
    assert recover(bytes(f)) == (99, abi.WAL_DROPPED_LAST, "dropped_last")
    # checksum_failure_in_middle_of_file_should_fail: ten bytes zeroed at offset 1000
    f = bytearray(clean); f[1000:1010] = bytes(10)
    assert recover(bytes(f)) == (0, abi.WAL_CORRUPT, "corrupt")
    f = bytearray(clean); f[50000:50010] = bytes(10)
    n_ok, status, outcome = recover(bytes(f))
    assert (status, outcome) == (abi.WAL_CORRUPT, "corrupt") and n_ok == (50000 - 5 - 1082) // 1033 + 1
    # a stored checksum of 0 means "not used" (:1022-1024): frames written without checksums validate
    body0 = eng.wal_frame(recs, data, total, abi.WAL_NO_CHECKSUMS)
    f = bytearray(abi.WAL_FILE_HEADER + body0.tobytes()); f[1000:1010] = bytes(10)
    assert recover(bytes(f)) == (100, abi.WAL_CLEAN, "eof")
    # records of an unregistered writer are not validated (:902, :929): the caller clears the flag
    f = bytearray(clean); f[1000:1010] = bytes(10)
    scanned, _, _ = engine.wal_scan(bytes(f))
    scanned["flags"] &= ~np.uint8(abi.WAL_REC_VALIDATE)
    assert eng.wal_validate(np.frombuffer(bytes(f), dtype=np.uint8), scanned) == (100, abi.WAL_CLEAN)
    with pytest.raises(engine.RgbError):
        bad = recs.copy(); bad["out_offset"][5] = bad["out_offset"][4]            # overlapping records
        eng.wal_frame(bad, data, total)
    eng.close()


@pytest.mark.gpu
def test_gpu_frame_full_size_batch_round_trips_through_the_recovery_path():
    """A 64 MiB batch (16 384 entries x 4 KiB) cannot be framed by the Python checker in seconds:
    frame it on the device, then read it back the way recovery does -- every stored checksum must
    validate and the walk must find every record with its payload offset where the layout put it."""
    import torch
    rng = np.random.default_rng(40)
    n, ln = 16384, 4096
    recs = np.zeros(n, dtype=abi.WAL_RECORD_DTYPE)
    recs["index"] = np.arange(1, n + 1); recs["term"] = 3
    hdr0 = header_bytes(1, 5, b"big_batch_writer")
    hdr = header_bytes(0, 5, None)
    payload = rng.integers(0, 256, size=n * ln, dtype=np.uint8)
    data = np.concatenate([np.frombuffer(hdr0 + hdr, dtype=np.uint8), payload, np.zeros(16, dtype=np.uint8)])
    recs["hdr_offset"] = len(hdr0); recs["hdr_len"] = 3
    recs["hdr_offset"][0] = 0; recs["hdr_len"][0] = len(hdr0)
    recs["data_offset"] = len(hdr0) + 3 + np.arange(n, dtype=np.uint64) * ln
    recs["data_len"] = ln
    total = engine.wal_layout(recs, 5)
    eng = _open()
    d_r = torch.from_numpy(recs.view(np.uint8)).cuda()
    d_d = torch.from_numpy(data).cuda()
    d_o = torch.zeros(total, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.wal_frame_device(d_r.data_ptr(), n, d_d.data_ptr(), len(data), d_o.data_ptr(), total)
    torch.cuda.synchronize()
    f = d_o.cpu().numpy()
    f[:5] = np.frombuffer(abi.WAL_FILE_HEADER, dtype=np.uint8)
    scanned, consumed, end = engine.wal_scan(f)
    assert len(scanned) == n and consumed == total and end == abi.WAL_END_DATA
    assert np.array_equal(scanned["index"], recs["index"]) and np.all(scanned["term"] == 3)
    assert np.array_equal(scanned["data_offset"], recs["out_offset"] + recs["hdr_len"] + 24)
    assert eng.wal_validate(f, scanned) == (n, abi.WAL_CLEAN)
    sample = rng.choice(n, size=32, replace=False)
    for i in sample:
        o = int(scanned["data_offset"][i])
        assert np.array_equal(f[o:o + ln], payload[i * ln:(i + 1) * ln])
        assert int(scanned["checksum"][i]) == zlib.adler32(struct.pack(">QQ", i + 1, 3) + payload[i * ln:(i + 1) * ln].tobytes())
    f[total // 2] ^= 0x5A
    n_ok, status = eng.wal_validate(f, scanned)
    assert status == abi.WAL_CORRUPT and 0 < n_ok < n
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("small", [0, 1, 2], ids=["wave_per_record", "four_per_wave", "eight_per_wave"])
def test_gpu_frame_every_source_and_destination_phase(small):
    """Payload lengths around the 16-byte chunk and the lane-group boundaries at every source phase x destination
    phase: the chunk that spills over behind the last source chunk is written by the lane that holds that chunk."""
    rng = np.random.default_rng(95 + small)
    lens = (SWEEP_LARGE, SWEEP_MID, SWEEP_SMALL)[small]
    recs, data, out_bytes, want = phase_sweep_batch(rng, lens, range(16), range(16) if small else (0, 1, 5, 8, 11, 15))
    assert (len(data) / len(recs) < 1024) == bool(small) and (len(data) / len(recs) <= 320) == (small == 2)
    eng = _open()
    try:
        check_phase_sweep(eng.wal_frame(recs, data, out_bytes).tobytes(), want, out_bytes)
    finally:
        eng.close()
