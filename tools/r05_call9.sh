#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05i; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
L="--steps 192 --warmup 16"
one new_1 new $L
one new2_1 new2 $L
one new_2 new $L
one new2_2 new2 $L
TL_HINT=2 TL_TICKS=32 TL_AGE=512 RGB_LIB=$V/timeline.so timeout 300 python tools/train_timeline.py > $OUT/timeline.txt 2> $OUT/timeline.err
grep -A14 "by MEANS" $OUT/timeline.txt | tee -a $OUT/summary.txt; grep "train of" $OUT/timeline.txt | tee -a $OUT/summary.txt
RGB_LIB=$V/hist.so timeout 300 python tools/train_decline_hist.py 2> $OUT/hist.err | grep snapshot_written | tee -a $OUT/summary.txt
