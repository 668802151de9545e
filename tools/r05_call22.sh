#!/bin/bash
# round 5, call 22: the two poll probes again, with the leaderboard snapshots as a KERNEL between the launches (the rows of
# an in-launch snapshot wait for exact sequence bytes, which an unordered launch does not leave)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05v; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --snapshot-kernel"
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
L="--steps 192 --warmup 16"
for i in 1 2; do
  one aos_long_$i new2 $L --check-ticks 2
  RGB_BENCH_NOCHECK=1 one nopoll_long_$i nopoll $L --check-ticks 0
  RGB_BENCH_NOCHECK=1 one nowait_long_$i nowait $L --check-ticks 0
done
