#!/bin/bash
# round 6 A/B runner: every ra_amd/csrc/variants/*.so (tools/build_variants.sh, N = 5 only) through the closed-loop bench
# in the driver's form and in the long form, interleaved and repeated so that a box's drift shows; every decision of
# the timed replay is compared with the per-tick launches of the generation pass (bench.py), except for variants whose
# name starts with x_ (timing probes that break parity).
#   gpurun -- 'bash tools/ab_variants.sh TAG [reps] [PREFIX]'      (PREFIX: only variants/PREFIX*.so)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; TAG=${1:-r06ab}; REPS=${2:-2}; PRE=${3:-}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  local nc="" ck=""; case $lib in x_*) nc=1; ck="--check-ticks 0";; esac
  RGB_BENCH_NOCHECK=$nc RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" $ck > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}  {d['config'].get('state_checksum','')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
L="--steps 192 --warmup 16"
D="--steps 20 --warmup 5"
for rep in $(seq 1 $REPS); do
  for v in $V/${PRE}*.so; do n=$(basename $v .so); one ${n}_drv_$rep $n $D; done
  for v in $V/${PRE}*.so; do n=$(basename $v .so); one ${n}_long_$rep $n $L; done
done
stamp done
