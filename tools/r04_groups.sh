#!/bin/bash
# does a train's state footprint per XCD (hot + peers + run-table lines touched per tick) fit the 4 MB L2?  time per
# decision against the number of groups
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04g; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0 --steps 192 --warmup 16"
for g in 8192 16384 24576 32768 40960 49152 57344 65536 98304; do
  python bench.py $Q --groups $g > $OUT/g$g.json 2> $OUT/g$g.err
  python -c "
import json; d=json.loads(open('$OUT/g$g.json').read().strip().splitlines()[-1]); r=d['roofline']; n=d['config']['decisions_per_tick']; print('groups $g', 'us/tick', round(r['avg_tick_us'],2), 'ps/decision', round(r['avg_tick_us']*1e6/n,1), 'frac', round(r['frac'],4))" || tail -3 $OUT/g$g.err
done
