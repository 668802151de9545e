#!/bin/bash
# same-box A/B of the PK_OLDLT_SH shortcut (term gate of evaluate_quorum without the run-table walk): the variant
# without it first (absorbs the cold start), the product with the first 8 ticks oracle-checked, the variant again
set -u
TAG=r02f6; OUT=gpurun_out/$TAG; mkdir -p $OUT
one() {  # label lib extra
  RGB_LIB=$2 timeout 60 python bench.py --steps 300 --warmup 32 --no-cpu-baseline --no-host-path --literal-ticks 0 $3 > $OUT/$1.json 2> $OUT/$1.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', round(d['roofline']['avg_launch_us'],2), 'us/tick', round(d['value']/1e9,2),'G/s', 'frac', round(d['roofline']['frac'],3), d['config']['state_checksum'], 'checked', d['config'].get('oracle_checked_ticks'))
except Exception as e: print('$1 failed', e)"
}
one nooldlt_1 $PWD/ra_amd/csrc/variants/nooldlt.so "--check-ticks 0"
one product $PWD/ra_amd/csrc/libra_gpu_batch.so "--check-ticks 8"
one nooldlt_2 $PWD/ra_amd/csrc/variants/nooldlt.so "--check-ticks 0"
