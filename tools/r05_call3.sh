#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05c; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
RGB_LIB=$V/hist.so timeout 300 python tools/train_decline_hist.py 2> $OUT/hist.err | tee $OUT/train_hist.txt
tail -5 $OUT/hist.err
