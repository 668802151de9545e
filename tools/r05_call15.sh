#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05o; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 0 --steps 20 --warmup 5 --device-plan"
RGB_LIB=$V/cur.so rocprofv3 --kernel-trace --stats -d $OUT/prof -o devplan -- python bench.py $Q > $OUT/devplan.json 2> $OUT/devplan.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200
