#!/bin/bash
# round 5, call 19: half-row fetch (PK_FS) against HEAD on one box, with and without the run-table line, the NODEPS floor
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05s; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
timeout 400 python -m pytest tests/test_train.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest (product lib): $(tail -1 $OUT/pytest.txt)" | tee -a $OUT/summary.txt
stamp pytest
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
for i in 1 2; do
  one head_drv_$i head $D
  one new_drv_$i new $D
  one head_noruns_drv_$i head_noruns $D
  one new_noruns_drv_$i new_noruns $D
  one new_nohalf_drv_$i new_nohalf $D
  RGB_BENCH_NOCHECK=1 one new_nodeps_drv_$i new_nodeps $D
done
stamp driver-form
L="--steps 192 --warmup 16"
one head_long head $L
one new_long new $L
one new_noruns_long new_noruns $L
RGB_BENCH_NOCHECK=1 one new_nodeps_long new_nodeps $L
stamp done
