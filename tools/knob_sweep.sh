#!/bin/bash
# usage: tools/knob_sweep.sh "0 1 2 3"   (RGB_DEBUG values; non-zero knobs break parity: timing only)
for dbg in ${1:-0}; do
  RGB_DEBUG=$dbg python bench.py --steps 200 --warmup 16 --no-cpu-baseline --check-ticks 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$dbg', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s')"
done
