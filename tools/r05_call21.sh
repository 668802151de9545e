#!/bin/bash
# round 5, call 21: planar streams with the half-wave-per-plane lane mapping; the two probes that bound a sequence tag
# inside the row: no poll at all (nopoll), one poll whose answer is ignored (nowait) -- both break parity (NOCHECK)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05u; mkdir -p $OUT
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
for i in 1 2; do
  one aos_drv_$i new2 $D --no-planes
  one planes_drv_$i new2 $D --planes
  RGB_BENCH_NOCHECK=1 one nopoll_drv_$i nopoll $D --check-ticks 0
  RGB_BENCH_NOCHECK=1 one nowait_drv_$i nowait $D --check-ticks 0
done
stamp driver-form
for i in 1 2; do
  one aos_long_$i new2 $L --no-planes
  one planes_long_$i new2 $L --planes
  RGB_BENCH_NOCHECK=1 one nopoll_long_$i nopoll $L --check-ticks 0
  RGB_BENCH_NOCHECK=1 one nowait_long_$i nowait $L --check-ticks 0
done
stamp done
