#!/bin/bash
# the host path (rgb_submit -> kernels -> rgb_collect) of every ra_amd/csrc/variants/PREFIX*.so on one box, interleaved:
#   gpurun -- 'bash tools/ab_host_path.sh TAG [PREFIX] [reps]'
# one JSON line per run (tools/host_path_ab.py): the digest of everything handed out must be the same for all of them
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06hp}; PRE=${2:-hp_}; REPS=${3:-2}; mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in ra_amd/csrc/variants/${PRE}*.so; do
    n=$(basename $v .so)
    RGB_LIB=$R/$v timeout 600 python tools/host_path_ab.py 2> $OUT/${n}_$rep.err | tail -1 > $OUT/${n}_$rep.json
    python - $OUT/${n}_$rep.json ${n}_$rep <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); s = d["small_batch"]
    print(f"{sys.argv[2]:14s} digest {d['digest']} dec {d['decisions']} rpcs {d['rpcs']} state {d['state_checksum']} | one thread {d['one_thread_M_per_s']:6.1f} M/s (view {d.get('one_thread_view_M_per_s', 0):6.1f}) | ns/msg {d['ns_per_message']} | small {s['messages']}: "
          + " | ".join(f"{k} p50 {v['round_trip_us_p50']} us (view {v.get('view_round_trip_us_p50')}) {v['breakdown_us_p50']}" for k, v in s.items() if isinstance(v, dict)))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
  done
done
