#!/bin/bash
# round 4, the very last GPU seconds: the closed-loop stream with WAL outages replayed on the device.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 40 python -m pytest tests/test_cluster_safety.py -m gpu -x -q -p no:cacheprovider -k "wal_down" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
