#!/usr/bin/env python3
"""Throughput of the batched WAL entry checksum kernel (include/ra_gpu_wal.h) against the HBM
roofline: every payload byte is read exactly once, so algorithmic bytes = payload + 32-byte entry
record + 4-byte checksum per entry.  CPU beside it: the oracle's byte loop and zlib on one core."""
import json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine
HBM_PEAK = 8000.0
eng = engine.RaGpuBatch(1, 1)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
res = []
for label, n, lo, hi in (("4 KiB payloads", 262144, 4096, 4096), ("1-16 KiB mixed", 131072, 1024, 16384),
                         ("256 B payloads", 1 << 21, 256, 256), ("64 KiB payloads", 16384, 65536, 65536)):
    rng = np.random.default_rng(1)
    lens = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))])[:-1]
    total = int(lens.astype(np.uint64).sum())
    entries = np.zeros(n, dtype=abi.WAL_ENTRY_DTYPE)
    entries["index"] = np.arange(n); entries["term"] = 3
    entries["data_offset"] = offs; entries["data_len"] = lens
    d_d = torch.randint(0, 256, (total + 16,), dtype=torch.uint8, device="cuda")
    d_e = torch.from_numpy(entries.view(np.uint8)).cuda()
    d_o = torch.zeros(n, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.wal_adler32_device(d_e.data_ptr(), n, d_d.data_ptr(), total + 16, d_o.data_ptr(), sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(stream)
        for _ in range(reps):
            eng.wal_adler32_device(d_e.data_ptr(), n, d_d.data_ptr(), total + 16, d_o.data_ptr(), sp)
        e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    alg = total + n * 36
    gbps = alg / (us * 1e-6) / 1e9
    # spot check + CPU rate on a bounded sample
    k = min(n, 2000)
    host = d_d[: int(offs[k - 1] + lens[k - 1])].cpu().numpy()
    t0 = time.perf_counter()
    want = np.array([zlib.adler32(int(entries["index"][i]).to_bytes(8, "big") + int(3).to_bytes(8, "big") +
                                  host[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()) for i in range(k)], dtype=np.uint32)
    cpu_s = time.perf_counter() - t0
    assert np.array_equal(d_o[:k].cpu().numpy().view(np.uint32), want), label
    res.append({"workload": label, "entries": n, "payload_bytes": total, "us_per_launch": us,
                "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK,
                "entries_per_s": n / (us * 1e-6),
                "cpu_zlib_GBps_1core": float(lens[:k].sum()) / cpu_s / 1e9})
    print(json.dumps(res[-1]))
    del d_d, d_e, d_o
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wal_bench.json"), "w"), indent=1)
