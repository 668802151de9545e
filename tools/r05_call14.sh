#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05n; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
timeout 600 python -m pytest tests/test_train.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_train.txt 2>&1
echo "pytest train: $(tail -1 $OUT/pytest_train.txt)" | tee -a $OUT/summary.txt
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f} G/s {d['value']/1e9:6.2f} wall-events {d.get('wall_minus_events_us')} plan_host_us/tick {d['config']['train']['plan_host_us_per_tick']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
L="--steps 192 --warmup 16"; D="--steps 20 --warmup 5"
for rep in 1 2; do
one cur_drv_$rep cur $D
one devplan_drv_$rep cur $D --device-plan
one persistent_drv_$rep cur $D --train-form persistent
done
one cur_long cur $L
one devplan_long cur $L --device-plan
one persistent_long cur $L --train-form persistent
