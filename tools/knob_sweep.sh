#!/bin/bash
# timing-only experiment: which part of the tick kernel costs what (results are WRONG with knobs on)
for lib in "" ra_amd/csrc/variants/lib_cw2.so ra_amd/csrc/variants/lib_cw3.so; do
for dbg in 0 15; do
  RGB_LIB=${lib:+$PWD/$lib} RGB_DEBUG=$dbg python bench.py --steps 100 --warmup 16 --no-cpu-baseline --check-ticks 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib dbg=$dbg', round(d['roofline']['avg_launch_us'],2), 'us/tick')"
done; done
