#!/bin/bash
# round 4, GPU call: product vs variants/*.so (tools/build_variants.sh) on the driver's form (twice) and the long form,
# then the train timeline of the product build.   usage: gpurun -- 'bash tools/r04_ab2.sh TAG'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=gpurun_out/${1:-r04c}; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-path --literal-ticks 0"
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
VARS=$(cd ra_amd/csrc/variants && ls *.so 2>/dev/null | sed 's/\.so$//')
for rep in 1 2; do
  one drv_product_$rep "" "--steps 20 --warmup 5"
  for v in $VARS; do one drv_${v}_$rep $v "--steps 20 --warmup 5"; done
done 2>&1 | tee $OUT/summary.txt
one long_product "" "--steps 192 --warmup 16" | tee -a $OUT/summary.txt
for v in $VARS; do one long_$v $v "--steps 192 --warmup 16" | tee -a $OUT/summary.txt; done
[ -f ra_amd/csrc/variants_tools/timeline.so ] && RGB_LIB=$R/ra_amd/csrc/variants_tools/timeline.so TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/timeline.txt 2>&1
grep -A14 "steady ticks" $OUT/timeline.txt; head -3 $OUT/timeline.txt | tail -2
