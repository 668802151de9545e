#!/bin/bash
# round 5, call 27: the row plan built on the device INSIDE the timed region, launches in the DEALT form (grid = the rows
# bound of a tick, surplus blocks exit) against the persistent form and against the host-built plan
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05aa; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}, "
          f"G/s {d['value']/1e9:6.2f}, wall-events {d.get('wall_minus_events_us')} us, form {d['config']['train']['form']}, blocks/tick {d['config']['train']['blocks_per_tick']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
D="--steps 20 --warmup 5"
L="--steps 192 --warmup 16"
for i in 1 2; do
  one hostplan_drv_$i dpd5 $D
  one devplan_dealt_drv_$i dpd5 $D --device-plan
  one devplan_pers_drv_$i dpd5 $D --device-plan --train-form persistent
done
for i in 1 2; do
  one hostplan_long_$i dpd5 $L
  one devplan_dealt_long_$i dpd5 $L --device-plan
  one devplan_pers_long_$i dpd5 $L --device-plan --train-form persistent
done
