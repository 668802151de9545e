#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05l; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f} G/s {d['value']/1e9:6.2f} wall-events {d.get('wall_minus_events_us')} {d.get('wall_breakdown')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
L="--steps 192 --warmup 16"; D="--steps 20 --warmup 5"
for rep in 1 2 3; do
one head_drv_$rep head $D
one ep_drv_$rep ep $D
one epns_drv_$rep ep_nosplit $D
done
for rep in 1 2; do
one head_long_$rep head $L
one ep_long_$rep ep $L
one epns_long_$rep ep_nosplit $L
done
