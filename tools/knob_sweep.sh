#!/bin/bash
for lib in "" ra_amd/csrc/variants/lib_O2.so ra_amd/csrc/variants/lib_Os.so ra_amd/csrc/variants/lib_Oz.so; do
  RGB_LIB=${lib:+$PWD/$lib} python bench.py --steps 200 --warmup 16 --no-cpu-baseline --check-ticks 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', d['config']['state_checksum'])"
done
