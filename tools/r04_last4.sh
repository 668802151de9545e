#!/bin/bash
# round 4, the last GPU seconds: the wal_down tests (per tick, class-dispatch launch, inside a train) and the WAL
# kernels' tests on the final build.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04i; mkdir -p $OUT
timeout 45 python -m pytest tests/test_gpu_parity.py tests/test_wal_framing.py tests/test_wal_checksum.py -m gpu -x -q -p no:cacheprovider -k "wal" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
