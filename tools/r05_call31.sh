#!/bin/bash
# round 5, call 31: rpc records staged in LDS and written out by the wavefront (leader-side slices of trains) against the
# same tree with -DRGB_X_RPC_LDS=0: the N = 5 closed loop (parity checks on) and the literal config 5 (N = 7, oracle-checked)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05ae; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 2"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 100 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
L="--steps 192 --warmup 16"
D="--steps 20 --warmup 5"
for i in 1 2; do
  one norpc5_long_$i norpc5 $L
  one rpc5_long_$i rpc5 $L
done
one norpc5_drv norpc5 $D
one rpc5_drv rpc5 $D
for i in 1 2; do
  for v in norpc7 rpc7; do
    RGB_LIB=$V/$v.so timeout 200 python tools/cfg5_probe.py 5 32 > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    echo "$v $i: $(tail -1 $OUT/${v}_$i.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["us_per_tick"],2), "us/tick frac", round(d["frac"],4), "train", round(d["train_launch"]["us_per_tick"],2), "state", d["final_state_equal"], "checked", d["oracle_checked_decisions"])' 2>&1 | tail -1)" | tee -a $OUT/summary.txt
  done
done
