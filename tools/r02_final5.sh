#!/bin/bash
# same-box A/B of the WAL framing kernel's partial-chunk stores: shift chains (product) vs variable-offset extraction
set -u
TAG=r02f5; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for v in product oldstores; do
    lib=$PWD/ra_amd/csrc/variants/$v.so; [ $v = product ] && lib=$PWD/ra_amd/csrc/libra_gpu_batch.so
    RGB_LIB=$lib timeout 60 python tools/wal_frame_bench.py > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
  done
done
python - <<'PY'
import json
for rep in (1,2):
    for v in ("product","oldstores"):
        for l in open(f"gpurun_out/r02f5/{v}_{rep}.json"):
            d=json.loads(l)
            if "workload" in d: print(rep, v, d["workload"], round(d["us_per_launch"],1), "us", round(d["frac_of_8TBps"],3))
PY
