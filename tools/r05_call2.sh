#!/bin/bash
# round 5, second GPU call: (1) why lanes decline the fast paths (per class, per reason), (2) the tick without dependency
# waits (RGB_X_TRAIN_NODEPS; the snapshot rows need the bytes, so both sides run --snapshot-kernel), (3) the benign-traffic
# bound (profiling build, RGB_DEBUG=128: the generator emits no anomalies -- what removing the general path could give)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05b; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
RGB_LIB=$V/hist.so TICKS=32 timeout 200 python tools/decline_hist.py 2> $OUT/hist.err | tee -a $OUT/summary.txt
stamp hist
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --check-ticks 0"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:18s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}, "
          f"G/s {d['value']/1e9:6.2f}, dec/tick {d['config']['decisions_per_tick']:.0f}, alg MB/tick {r['algorithmic_bytes_per_tick']/1e6:.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
L="--steps 192 --warmup 16 --snapshot-kernel"
one head_long_sk head $L
RGB_BENCH_NOCHECK=1 one nodeps_long_sk nodeps $L
one prof_long prof --steps 192 --warmup 16
RGB_DEBUG=128 one prof_benign_long prof --steps 192 --warmup 16
one head_long_sk2 head $L
RGB_BENCH_NOCHECK=1 one nodeps_long_sk2 nodeps $L
stamp done
