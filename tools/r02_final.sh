#!/bin/bash
# last GPU call of round 2: (1) state-round-trip slope probe -- product vs RGB_X_EXTRA_FETCH=1/2 variants (+64 / +128
# gathered bytes per message, thrown away) on the same box; (2) the WAL gpu tests + smoke after the host-form fixes
set -u
TAG=r02f; OUT=gpurun_out/$TAG; mkdir -p $OUT
STEPS=300 WARM=32 EXTRA="--literal-ticks 0" timeout 330 tools/knob_sweep.sh $TAG "" 2>&1 | tee $OUT/sweep.txt
timeout 200 python -m pytest tests/test_wal_framing.py tests/test_wal_checksum.py -m gpu -x -q > $OUT/wal_tests.log 2>&1; tail -3 $OUT/wal_tests.log
timeout 100 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
