#!/bin/bash
# other sizes on the current kernels: groups per GPU at N = 5, and N = 3 / 7 at 65 536 groups (200 timed ticks, aged 300)
set -u
TAG=${1:-r02z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
one() {
  timeout 600 python bench.py --groups $1 --members $2 --steps 200 --warmup 32 --age 300 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 2 > $OUT/g$1_n$2.json 2> $OUT/g$1_n$2.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/g$1_n$2.json').read().strip().splitlines()[-1]); print('groups $1 members $2:', round(d['roofline']['avg_launch_us'],2), 'us/tick', int(d['config']['decisions_per_tick']), 'decisions/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3))
except Exception as e: print('groups $1 members $2 failed', e, open('$OUT/g$1_n$2.err').read()[-300:])"
}
one 32768 5; one 65536 5; one 131072 5; one 262144 5; one 65536 3; one 65536 7
