#!/bin/bash
# round 4, GPU call 1: GPU parity suite, then product vs the 4-waves-per-SIMD variants (tools/build_variants.sh) on the
# driver's form, the better one again on the long form, train timelines and the fast-path decline histogram.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=gpurun_out/r04a; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-path --literal-ticks 0"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest.txt
one() { # name lib args
  local lib=""; [ -n "$2" ] && lib="$R/ra_amd/csrc/variants/$2.so"
  RGB_LIB=$lib timeout 300 python bench.py $Q $3 ${2:+--members 5} > $OUT/$1.json 2> $OUT/$1.err
  python - $OUT/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:24s} {r.get('avg_tick_us', r['avg_launch_us']):7.2f} us/tick frac {r['frac']:.4f} ms/step {d['ms_per_step']:.5f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
for rep in 1 2; do
  one drv_product_$rep "" "--steps 20 --warmup 5"
  one drv_w4a_$rep w4a "--steps 20 --warmup 5"
  one drv_w4b_$rep w4b "--steps 20 --warmup 5"
done 2>&1 | tee $OUT/summary.txt
one long_product "" "--steps 192 --warmup 16" | tee -a $OUT/summary.txt
one long_w4a w4a "--steps 192 --warmup 16" | tee -a $OUT/summary.txt
one long_w4b w4b "--steps 192 --warmup 16" | tee -a $OUT/summary.txt
RGB_LIB=$R/ra_amd/csrc/variants_tools/timeline.so TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/timeline.txt 2>&1
RGB_LIB=$R/ra_amd/csrc/variants_tools/timeline4a.so TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/timeline4a.txt 2>&1
RGB_LIB=$R/ra_amd/csrc/variants_tools/hist.so timeout 200 python tools/decline_hist.py > $OUT/decline_hist.txt 2>&1
tail -25 $OUT/timeline.txt; tail -16 $OUT/timeline4a.txt | head -14; tail -30 $OUT/decline_hist.txt
