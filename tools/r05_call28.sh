#!/bin/bash
# round 5, call 28: probes (break parity): no poll; no poll and no sequence-byte store (all of the byte traffic gone); no
# decision stores.  192 timed ticks as twelve 16-tick launches with the snapshot kernel between them.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05ab; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --snapshot-kernel --steps 192 --warmup 16"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 100 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
for i in 1 2; do
  one cur_$i cur --check-ticks 2
  RGB_BENCH_NOCHECK=1 one nopoll_$i nopoll --check-ticks 0
  RGB_BENCH_NOCHECK=1 one nobytes_$i nobytes --check-ticks 0
  RGB_BENCH_NOCHECK=1 one nodec_$i nodec --check-ticks 0
done
