#!/usr/bin/env python3
"""Throughput of the WAL record framing kernel (include/ra_gpu_wal.h: rgb_wal_frame_device) against
the HBM roofline.  Per record the kernel reads the payload and the 48-byte descriptor once and
writes the payload plus 27 prefix bytes once: algorithmic bytes = 2 * payload + 48 + 3 + 27 per
record (known-writer header).  The first records are checked against struct.pack + zlib."""
import json, os, struct, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine
HBM_PEAK = 8000.0
eng = engine.RaGpuBatch(1, 1)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
res = []
ALIGNED = os.environ.get("WAL_ALIGNED") == "1"     # experiment: 8-byte header, base 0 -> source and destination share their 16-byte phase
hdr = ((1 << 22) | 9).to_bytes(3, "big") + (b"\0" * 5 if ALIGNED else b"")
BASE = 0 if ALIGNED else 5
if os.environ.get("WAL_MEMCPY") == "1":
    a = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(json.dumps({"calibration": "torch copy_ of 1 GiB (read + write)", "us": us, "GBps_read_plus_write": 2 * (1 << 30) / (us * 1e-6) / 1e9}))
    del a, b
CASES = (("4 KiB payloads", 262144, 4096, 4096), ("1-16 KiB mixed", 131072, 1024, 16384),
         ("256 B payloads", 1 << 21, 256, 256), ("40-320 B mixed", 1 << 21, 40, 320),
         ("400-1000 B mixed", 1 << 20, 400, 1000))
ONLY = os.environ.get("WAL_CASES")                 # e.g. WAL_CASES="256 B,40-320": substrings of the labels to run
for label, n, lo, hi in CASES:
    if ONLY and not any(w.strip() in label for w in ONLY.split(",")):
        continue
    rng = np.random.default_rng(1)
    lens = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
    offs = 16 + np.concatenate([[0], np.cumsum(lens.astype(np.uint64))])[:-1]
    total = int(lens.astype(np.uint64).sum())
    recs = np.zeros(n, dtype=abi.WAL_RECORD_DTYPE)
    recs["index"] = np.arange(1, n + 1); recs["term"] = 3
    recs["data_offset"] = offs; recs["data_len"] = lens
    recs["hdr_offset"] = 0; recs["hdr_len"] = len(hdr)
    out_bytes = engine.wal_layout(recs, BASE)
    d_d = torch.randint(0, 256, (total + 32,), dtype=torch.uint8, device="cuda")
    d_d[:len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8)
    d_r = torch.from_numpy(recs.view(np.uint8)).cuda()
    d_o = torch.zeros(out_bytes, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.wal_frame_device(d_r.data_ptr(), n, d_d.data_ptr(), total + 32, d_o.data_ptr(), out_bytes, 0, 0, sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(stream)
        for _ in range(reps):
            eng.wal_frame_device(d_r.data_ptr(), n, d_d.data_ptr(), total + 32, d_o.data_ptr(), out_bytes, 0, 0, sp)
        e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    alg = 2 * total + n * (48 + len(hdr) + 24 + len(hdr))
    gbps = alg / (us * 1e-6) / 1e9
    k = 500
    end_k = int(recs["out_offset"][k])
    host_in = d_d[: int(offs[k])].cpu().numpy()
    want = b"".join(hdr + struct.pack(">II", zlib.adler32(struct.pack(">QQ", i + 1, 3) + host_in[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()), int(lens[i]))
                    + struct.pack(">QQ", i + 1, 3) + host_in[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes() for i in range(k))
    assert os.environ.get("WAL_NOCHECK") == "1" or d_o[BASE:end_k].cpu().numpy().tobytes() == want, label
    res.append({"kernel": "rgb_wal_frame_kernel", "workload": label, "records": n, "payload_bytes": total,
                "us_per_launch": us, "algorithmic_bytes": alg, "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK,
                "records_per_s": n / (us * 1e-6), "file_bytes_per_s": out_bytes / (us * 1e-6)})
    print(json.dumps(res[-1]))
    del d_d, d_r, d_o
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wal_frame_bench.json"), "w"), indent=1)
