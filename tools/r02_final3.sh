#!/bin/bash
# WAL framing kernel after the tail-fold + shift-chain stores: gpu tests, then the frame bench on the product and on the
# -DWAL_X_TAILFOLD=0 variant (same box)
set -u
TAG=r02f3; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 150 python -m pytest tests/test_wal_framing.py tests/test_wal_checksum.py -m gpu -x -q > $OUT/wal_tests.log 2>&1; tail -3 $OUT/wal_tests.log
WAL_MEMCPY=1 timeout 80 python tools/wal_frame_bench.py > $OUT/wal_frame_product.json 2> $OUT/wal_frame_product.err
RGB_LIB=$PWD/ra_amd/csrc/variants/notailfold.so timeout 80 python tools/wal_frame_bench.py > $OUT/wal_frame_notailfold.json 2> $OUT/wal_frame_notailfold.err
python - <<'PY'
import json
for v in ("product","notailfold"):
    for l in open(f"gpurun_out/r02f3/wal_frame_{v}.json"):
        d=json.loads(l)
        if "workload" in d: print(v, d["workload"], round(d["us_per_launch"],1), "us", round(d["frac_of_8TBps"],3))
        elif "calibration" in d: print(v, "memcpy", round(d["GBps_read_plus_write"]/8000,3))
PY
