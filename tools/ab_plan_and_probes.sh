#!/bin/bash
# round 6: the producer-built device plan (fitted grid: rgb_train_plan_fit) against the host-built plan and against the
# device build inside the region, same box, both forms (variants/pro.so = the product's sources, N = 5 only)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06e}; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  local nc="" ck=""; case $lib in x_*) nc=1; ck="--check-ticks 0";; esac
  RGB_BENCH_NOCHECK=$nc RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" $ck > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}  blocks/tick {d['config']['train']['blocks_per_tick']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
L="--steps 192 --warmup 16"; D="--steps 20 --warmup 5"
for rep in 1 2 3; do
  one host_drv_$rep pro $D --plan host
  one producer_drv_$rep pro $D --plan producer
  one region_drv_$rep pro $D --plan region
done
for rep in 1 2; do
  one host_long_$rep pro $L --plan host
  one producer_long_$rep pro $L --plan producer
done
