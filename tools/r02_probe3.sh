#!/bin/bash
# fast paths A/B: product-like variants on the bench stream, and the profiling builds on benign traffic (knob 128:
# every lane takes the steady-state outcome) = the upper bound of what compaction of the rare lanes can give
set -u
TAG=${1:-r02g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
one() {  # label lib dbg
  RGB_LIB=$PWD/ra_amd/csrc/variants/$2.so RGB_DEBUG=$3 timeout 300 python bench.py --steps 300 --warmup 32 --age ${AGE:-300} --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks ${CHECK:-0} \
      > $OUT/$1.json 2> $OUT/$1.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3), d['config']['state_checksum'], int(d['config']['decisions_per_tick']))
except Exception as e: print('$1 failed', e, open('$OUT/$1.err').read()[-400:])"
}
CHECK=2 one fast fast ""
CHECK=2 one nofast nofast ""
one fast_again fast ""
one benign_fast proffast 128
one benign_nofast profnofast 128
one prof_fast proffast 0
