#!/bin/bash
# wave timeline of a profiling variant (aged state): TL_TICKS ticks then the stamps of the last launch
set -u
TAG=${1:-r02s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in $VARIANTS; do
  RGB_LIB=$PWD/ra_amd/csrc/variants/$v.so TL_TICKS=${TL_TICKS:-300} timeout 300 python tools/wave_timeline.py > $OUT/tl_$v.txt 2>&1
  grep -A40 "end time per class" $OUT/tl_$v.txt | head -60
done
