#!/bin/bash
# per-wavefront timelines of the literal configurations 3 (N = 5) and 5 (N = 7) as trains: where the chain-bound ticks' lives go
# (variants c3_timeline.so / c5_timeline.so = -DRGB_X_TRAIN_TIMELINE builds, ONLY_N = 5 / 7)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06tl}; mkdir -p $OUT
RGB_LITERAL_TIMELINE=1 RGB_LIB=$R/ra_amd/csrc/variants/c3_timeline.so timeout 400 python tools/cfg5_probe.py 3 32 > $OUT/config3_timeline.txt 2> $OUT/config3.err
RGB_LITERAL_TIMELINE=1 RGB_LIB=$R/ra_amd/csrc/variants/c5_timeline.so timeout 400 python tools/cfg5_probe.py 5 32 > $OUT/config5_timeline.txt 2> $OUT/config5.err
tail -5 $OUT/config3.err $OUT/config5.err
