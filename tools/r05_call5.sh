#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05e; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
for h in 1 2; do
  TL_HINT=$h TL_TICKS=32 TL_AGE=512 RGB_LIB=$V/timeline.so timeout 300 python tools/train_timeline.py > $OUT/timeline_hint$h.txt 2> $OUT/timeline_hint$h.err
  grep -A16 "steady ticks" $OUT/timeline_hint$h.txt; grep "train of" $OUT/timeline_hint$h.txt
done
