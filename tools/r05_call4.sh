#!/bin/bash
# round 5, call 4: the header-compare hint (rgb_synth_set_hint 2) and fast_aer_reply's walk of the runs behind the LDS line
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05d; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}, "
          f"G/s {d['value']/1e9:6.2f}, wall-events {d.get('wall_minus_events_us')} us, blocks/tick {d['config']['train']['blocks_per_tick']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
D="--steps 20 --warmup 5"; L="--steps 192 --warmup 16"
one head_drv_1 head $D
one new_state_drv_1 new $D --hint state
one new_header_drv_1 new $D --hint header
one head_long head $L
one new_state_long new $L --hint state
one new_header_long new $L --hint header
one head_drv_2 head $D
one new_state_drv_2 new $D --hint state
one new_header_drv_2 new $D --hint header
stamp bench
RGB_LIB=$V/hist.so timeout 300 python tools/train_decline_hist.py 2> $OUT/hist.err | tee $OUT/train_hist.txt | tail -22 | tee -a $OUT/summary.txt
stamp done
