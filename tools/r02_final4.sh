#!/bin/bash
# the shipped WAL framing kernel (shift-chain head/tail stores): gpu tests + the frame bench with the memcpy calibration
set -u
TAG=r02f4; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 120 python -m pytest tests/test_wal_framing.py tests/test_wal_checksum.py -m gpu -x -q > $OUT/wal_tests.log 2>&1; tail -2 $OUT/wal_tests.log
WAL_MEMCPY=1 timeout 80 python tools/wal_frame_bench.py > $OUT/wal_frame.json 2> $OUT/wal_frame.err; cat $OUT/wal_frame.json | cut -c1-260
