#!/bin/bash
# the whole -m gpu suite on the box (the product library is built on the box if the .so is stale)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05suite; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
