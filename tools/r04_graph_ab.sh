#!/bin/bash
# the driver's 20-step form: ONE train launch per timed region -- hipGraph replay vs eager launches
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04f; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0"
for rep in 1 2 3; do for m in graph eager; do
  f=""; [ $m = eager ] && f="--no-graph"
  python bench.py --steps 20 --warmup 5 $Q $f > $OUT/${m}_$rep.json 2> $OUT/${m}_$rep.err
  python -c "
import json; d=json.loads(open('$OUT/${m}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$m $rep', 'events us/tick', round(r['avg_tick_us'],2), 'wall us/step', round(d['ms_per_step']*1e3,2), 'frac', round(r['frac'],4), 'G/s', round(d['value']/1e9,2))"
done; done
