#!/bin/bash
# usage (GPU box): tools/knob_sweep.sh TAG "0 1 2 8 32 64"  -- RGB_DEBUG values on the profiling build
# (make -C ra_amd/csrc prof); non-zero knobs break parity: timing only.  Also times the product library and,
# when present, ra_amd/csrc/variants/*.so with the same command (same box, same settings).
TAG=${1:-ks}; KNOBS=${2:-0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
one() {  # label lib dbg
  RGB_LIB=$2 RGB_DEBUG=$3 timeout 300 python bench.py --steps ${STEPS:-300} --warmup ${WARM:-300} --no-cpu-baseline --no-host-path --check-ticks 0 ${EXTRA:-} \
      > $OUT/$1.json 2> $OUT/$1.err
  python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3), d['config']['state_checksum'])
except Exception as e: print('$1 failed', e)"
}
one product $PWD/ra_amd/csrc/libra_gpu_batch.so ""
for v in ra_amd/csrc/variants/*.so; do [ -f "$v" ] && one $(basename $v .so) $PWD/$v ""; done
[ -f ra_amd/csrc/libra_gpu_batch_prof.so ] && for dbg in $KNOBS; do one prof_dbg$dbg $PWD/ra_amd/csrc/libra_gpu_batch_prof.so $dbg; done
one product_again $PWD/ra_amd/csrc/libra_gpu_batch.so ""
