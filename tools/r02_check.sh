#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/r02_check.sh TAG'  -- the -m gpu suite, then the default bench (what the driver runs)
set -u
TAG=${1:-r02d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
( time timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
tail -c 600 $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"]*1e3,2), "us/tick", round(d["value"]/1e9,2), "G/s frac", round(d["roofline"]["frac"],3))
print("cpu", {k: (round(v/1e6,2) if isinstance(v,float) else v) for k,v in d["cpu_baseline"].items() if k != "sample"})
for k, v in (d.get("literal_configs") or {}).items():
    print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("us_per_tick","value","frac","decisions_per_tick","oracle_checked_decisions","error","host_generation_s")})
print("host_path", d["host_path"] and round(d["host_path"]["value"]/1e6,1), "M/s")
print("wal", {k: v and round(v.get("frac",0),3) for k,v in d["aux_kernels"].items()})
PY
