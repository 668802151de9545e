#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05m; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f} G/s {d['value']/1e9:6.2f} wall-events {d.get('wall_minus_events_us')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
L="--steps 192 --warmup 16"; D="--steps 20 --warmup 5"
for rep in 1 2 3; do
one head_drv_$rep head $D
one cur_drv_$rep cur $D
one cur_state_drv_$rep cur $D --hint state
one front_drv_$rep cur_front $D
done
for rep in 1 2; do
one head_long_$rep head $L
one cur_long_$rep cur $L
one cur_state_long_$rep cur $L --hint state
one front_long_$rep cur_front $L
done
