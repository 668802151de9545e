#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05h; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
for h in 2; do
  TL_HINT=$h TL_TICKS=32 TL_AGE=512 RGB_LIB=$V/timeline.so timeout 300 python tools/train_timeline.py > $OUT/timeline_hint$h.txt 2> $OUT/timeline_hint$h.err
  grep -B1 -A45 "steady ticks" $OUT/timeline_hint$h.txt | head -70; grep "train of" $OUT/timeline_hint$h.txt
done
