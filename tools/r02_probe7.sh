#!/bin/bash
set -u
TAG=${1:-r02q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
one() {
  RGB_LIB=$R/ra_amd/csrc/variants/$2.so timeout 300 python bench.py --steps 1000 --warmup 32 --age 512 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 0 > $OUT/$1.json 2> $OUT/$1.err
  python -c "
import json
d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3), d['config']['state_checksum'])"
}
for r in 1 2 3; do for v in $VARIANTS; do one ${v}_$r $v; done; done
