#!/bin/bash
set -u
TAG=${1:-r02k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
one() {  # label lib
  RGB_LIB=$R/ra_amd/csrc/variants/$2.so timeout 300 python bench.py --steps 300 --warmup 32 --age 300 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks ${CHECK:-0} \
      > $OUT/$1.json 2> $OUT/$1.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3), d['config']['state_checksum'], int(d['config']['decisions_per_tick']))
except Exception as e: print('$1 failed', e, open('$OUT/$1.err').read()[-400:])"
}
for v in ${VARIANTS:-fast reread}; do CHECK=1 one $v $v; done
for v in ${VARIANTS:-fast reread}; do one ${v}_again $v; done
for v in ${PROFV:-profreread}; do
RGB_LIB=$R/ra_amd/csrc/variants/$v.so TL_DBG=16 TL_TICKS=300 timeout 300 python tools/wave_timeline.py > $OUT/tl_$v.txt 2>&1
echo "== $v"; grep -E "^class [0-3]:|^class 11|^waves|lane 0" $OUT/tl_$v.txt | cut -c1-250 | head -12
done
