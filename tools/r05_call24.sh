#!/bin/bash
# round 5, call 24: (a) the dealt kernel's prologue with the tick header and the row entry requested together (pro5) against
# HEAD, N = 5 closed loop; (b) BASELINE configs[4]: leader-side slices of 32 with the 192-byte peers rows in LDS (wide7)
# against HEAD (head7), the bench's literal config 5 on its own (tools/cfg5_probe.py)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05x; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}, "
          f"G/s {d['value']/1e9:6.2f}, wall-events {d.get('wall_minus_events_us')} us")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
D="--steps 20 --warmup 5"
L="--steps 192 --warmup 16"
for i in 1 2 3; do
  one head_drv_$i head $D
  one pro5_drv_$i pro5 $D
done
for i in 1 2; do
  one head_long_$i head $L
  one pro5_long_$i pro5 $L
done
stamp n5
for i in 1 2; do
  for v in head7 wide7; do
    RGB_LIB=$V/$v.so timeout 300 python tools/cfg5_probe.py 5 32 > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    echo "$v $i: $(tail -1 $OUT/${v}_$i.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["us_per_tick"],2), "us/tick frac", round(d["frac"],4), "per-tick", round(d["per_tick_launches"]["us_per_tick"],2), "train", round(d["train_launch"]["us_per_tick"],2), "state", d["final_state_equal"], "checked", d["oracle_checked_decisions"])' 2>&1 | tail -1)" | tee -a $OUT/summary.txt
  done
done
stamp cfg5
