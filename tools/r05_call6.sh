#!/bin/bash
# round 5, call 6: with the header hint the bulk wavefronts are short -- what do the dependency waits cost now
# (NODEPS floor), and what do other class leads give (RGB_TRAIN_LEAD = "class:ticks,...")
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05f; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 0"
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
one new_h2_sk new $L --snapshot-kernel
RGB_BENCH_NOCHECK=1 one nodeps_h2_sk nodeps2 $L --snapshot-kernel
RGB_BENCH_NOCHECK=1 one nodeps_h1_sk nodeps2 $L --snapshot-kernel --hint state
one new_h2 new $L
# leads: index = class rank (0 aer 1 reply 2 written 3 append 4 pipeline 5 rv 6 vote_res 8 el_timeout 10 pre_vote_res 11 snap_written 12 hb_rpc 13 hb_reply 14 query)
RGB_TRAIN_LEAD="0:0,1:0.15,2:0,3:0.30,4:0.30,6:0.12,8:0.09,11:0.60,13:0.12,14:0.08" one lead_A new $L
RGB_TRAIN_LEAD="0:0,1:0.25,2:0,3:0.40,4:0.40,6:0.15,8:0.10,11:0.75,13:0.15,14:0.10,5:0.05,10:0.05" one lead_B new $L
RGB_TRAIN_LEAD="0:0,1:0.10,2:0,3:0.20,4:0.20,6:0.08,8:0.05,11:0.45,13:0.08,14:0.05" one lead_C new $L
RGB_TRAIN_LEAD="0:0,1:0,2:0,3:0,4:0,11:0" one lead_zero new $L
RGB_TRAIN_LEAD="0:0.2,1:0.35,2:0.2,3:0.50,4:0.50,6:0.3,8:0.3,11:0.80,13:0.3,14:0.3,5:0.2,10:0.2,12:0.2" one lead_D new $L
one new_h2_b new $L
stamp done
