#!/bin/bash
# round 5, first GPU call: where HEAD of round 4 stands on this round's box, and three probes that decide the round:
#  (1) RGB_X_TRAIN_NODEPS: the tick without dependency waits (the floor of any overlap tuning),
#  (2) RGB_TRAIN_RUNS_LDS=0: what the run-table line fetched by every leader-side wavefront costs / buys,
#  (3) graph vs eager for the driver's 20-step region on the WALL clock (what `value` reports), with the host's share
#      of the region broken down (bench.py wall_breakdown).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05a; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 0"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}, "
          f"G/s {d['value']/1e9:6.2f}, wall-events {d.get('wall_minus_events_us')} us {d.get('wall_breakdown')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
D="--steps 20 --warmup 5"
one head_drv_1 head $D
one head_drv_graph_1 head $D --graph
RGB_BENCH_NOCHECK=1 one nodeps_drv_1 nodeps $D
one noruns_drv_1 noruns $D
one head_drv_2 head $D
one head_drv_graph_2 head $D --graph
RGB_BENCH_NOCHECK=1 one nodeps_drv_2 nodeps $D
one noruns_drv_2 noruns $D
stamp driver-form
L="--steps 192 --warmup 16"
one head_long head $L
RGB_BENCH_NOCHECK=1 one nodeps_long nodeps $L
one noruns_long noruns $L
one head_drv_3 head $D
one head_drv_graph_3 head $D --graph
stamp done
