#!/bin/bash
# literal config 5 (65 536 x 7, repair) on every ra_amd/csrc/variants/c5_*.so (ONLY_N=7 builds), interleaved, oracle-checked
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06c5}; mkdir -p $OUT
for rep in 1 2 3; do
  for v in ra_amd/csrc/variants/c5_*.so; do
    n=$(basename $v .so)
    RGB_LIB=$R/$v timeout 300 python tools/cfg5_probe.py 5 32 2> $OUT/${n}_$rep.err | tail -1 > $OUT/${n}_$rep.json
    python - $OUT/${n}_$rep.json ${n}_$rep <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(f"{sys.argv[2]:20s} {d['us_per_tick']:7.2f} us/tick frac {d['frac']:.4f} state_equal {d['final_state_equal']} checked {d['oracle_checked_decisions']} train {d['train_launch']['us_per_tick']:.2f} per-tick {d['per_tick_launches']['us_per_tick']:.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
  done
done
