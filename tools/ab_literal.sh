#!/bin/bash
# one literal configuration of bench.py (tools/cfg5_probe.py: oracle-checked, then timed as per-tick launches and as ONE
# train) on every ra_amd/csrc/variants/PREFIX*.so, interleaved, three times:
#   gpurun -- 'bash tools/ab_literal.sh TAG PREFIX CONFIG'      (config 5 needs ONLY_N=7 builds, configs 2 / 3 N = 5)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06lit}; PRE=${2:-c5_}; CFG=${3:-5}; mkdir -p $OUT
for rep in 1 2 3; do
  for v in ra_amd/csrc/variants/${PRE}*.so; do
    n=$(basename $v .so)
    RGB_LIB=$R/$v timeout 300 python tools/cfg5_probe.py $CFG 32 2> $OUT/${n}_$rep.err | tail -1 > $OUT/${n}_$rep.json
    python - $OUT/${n}_$rep.json ${n}_$rep <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(f"{sys.argv[2]:20s} {d['us_per_tick']:7.2f} us/tick frac {d['frac']:.4f} state_equal {d['final_state_equal']} checked {d['oracle_checked_decisions']} train {d['train_launch']['us_per_tick']:.2f} per-tick {d['per_tick_launches']['us_per_tick']:.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
  done
done
