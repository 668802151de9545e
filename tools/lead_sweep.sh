#!/bin/bash
# ordering probe on a literal configuration: the class leads of the train's row plan (rgb_train_lead, in ticks)
#   gpurun -- 'bash tools/lead_sweep.sh TAG CONFIG "1:0.0" "1:0.3" "1:0.5,0:0.1" ...'     (every run oracle-checked)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-leads}; CFG=${2:-3}; shift 2; mkdir -p $OUT
for rep in 1 2; do
  for L in "default" "$@"; do
    if [ "$L" = default ]; then unset RGB_TRAIN_LEAD; else export RGB_TRAIN_LEAD="$L"; fi
    timeout 300 python tools/cfg5_probe.py $CFG 32 2> $OUT/err.txt | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('%-24s %7.2f us/tick frac %.4f state_equal %s' % ('$L', d['us_per_tick'], d['frac'], d['final_state_equal']))" | tee -a $OUT/summary.txt
  done
done
