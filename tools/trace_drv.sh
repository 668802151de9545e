#!/bin/bash
# kernel trace of the driver's short form for the product library and (if present) variants: per-kernel calls / avg / total
set -u
TAG=${1:-trace}; R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0 --steps 20 --warmup 5 --members 5"
for lib in product $(ls $R/ra_amd/csrc/variants/*.so 2>/dev/null); do
  n=$(basename $lib .so)
  if [ "$lib" = product ]; then unset RGB_LIB; else export RGB_LIB=$lib; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o t -- python $R/bench.py $Q > $OUT/$n.log 2>&1
  echo "== $n"; python - "$OUT/$n" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) <= 64 or "train" in r["Name"]:
            print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.2f} total_us {float(r["TotalDurationNs"])/1e3:10.1f}')
PY
done
