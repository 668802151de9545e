#!/bin/bash
# bench.py's host-path legs (one thread, 4 + 3 threads, four rounds per batch, the small batch) for every
# ra_amd/csrc/variants/PREFIX*.so on one box, interleaved:   gpurun -- 'bash tools/ab_host_bench.sh TAG PREFIX [reps]'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-hb}; PRE=${2:-cs_}; REPS=${3:-2}; mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in ra_amd/csrc/variants/${PRE}*.so; do
    n=$(basename $v .so)
    RGB_LIB=$R/$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --literal-ticks 0 2> $OUT/${n}_$rep.err | tail -1 > $OUT/${n}_$rep.json
    python - $OUT/${n}_$rep.json ${n}_$rep <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    h = json.loads(open(sys.argv[1]).read())["host_path"]; s = h["rounds4_small"]["launch_per_round"]
    print(f"{sys.argv[2]:12s} one thread {h['value']/1e6:6.1f} M/s (view {h['collect_view']['value']/1e6:6.1f}) | 4+3 threads {h['threads4']['value']/1e6:6.1f} | rounds4 {h['rounds4']['launch_per_round']['value']/1e6:5.1f} / fused {h['rounds4']['fused_train']['value']/1e6:5.1f} | small {s['round_trip_us_p50']} us (view {s.get('collect_view_round_trip_us_p50')}) {s['breakdown_us_p50']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-1200:])
PY
  done
done
