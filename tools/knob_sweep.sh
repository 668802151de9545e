#!/bin/bash
for dbg in 0 64; do
  RGB_DEBUG=$dbg python bench.py --steps 200 --warmup 16 --no-cpu-baseline --check-ticks 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$dbg', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', d['config']['state_checksum'])"
done
