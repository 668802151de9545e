#!/bin/bash
set -u
TAG=${1:-r02i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
for v in proffast profocc3; do
RGB_LIB=$R/ra_amd/csrc/variants/$v.so TL_DBG=16 TL_TICKS=300 timeout 300 python tools/wave_timeline.py > $OUT/tl_$v.txt 2>&1
echo "== $v"; grep -E "^class [0-3]:|^waves|lane 0|t= " $OUT/tl_$v.txt | cut -c1-260 | grep -v "^class [5-9]\|class 1[01]" | head -24
done
