#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05j; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
TL_HINT=2 TL_TICKS=32 TL_AGE=512 RGB_LIB=$V/timeline.so timeout 300 python tools/train_timeline.py > $OUT/timeline.txt 2> $OUT/timeline.err
grep -A34 "by MEANS" $OUT/timeline.txt; grep "train of" $OUT/timeline.txt
