#!/bin/bash
# First GPU call of round 3:  build the probe variants HERE first
#     tools/build_variants.sh xs16:"-DRGB_X_EXTRA_STORE=16" xs64:"-DRGB_X_EXTRA_STORE=64" xf1:"-DRGB_X_EXTRA_FETCH=1"
# then   gpurun --timeout 300 -- 'bash tools/r03_first_call.sh'
# Same box, same command: product, +16 / +64 dirty bytes per message in a line of their own (write side: per line or
# per byte?), +64 gathered bytes per message (read side, the round-2 reference point: +4 us), product again.
# Reading: if xs16 ~ xs64 the end-of-kernel write-back is priced per dirty LINE (then fewer dirty lines per decision --
# peers words next to the hot words -- is what shortens it); if xs64 costs ~4x xs16 it is priced per byte.
set -u
TAG=${1:-r03a}
STEPS=300 WARM=32 EXTRA="--literal-ticks 0" timeout 280 tools/knob_sweep.sh $TAG "" 2>&1 | tee gpurun_out/$TAG.sweep.txt
