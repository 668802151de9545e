#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/r02_check2.sh TAG'  -- GPU suite, WAL benches, steady-state tick timing
set -u
TAG=${1:-r02f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 4 $OUT/pytest_gpu.log
timeout 300 python tools/wal_frame_bench.py > $OUT/wal_frame.json 2> $OUT/wal_frame.err; tail -c 1500 $OUT/wal_frame.json; tail -c 300 $OUT/wal_frame.err
timeout 300 python tools/wal_bench.py > $OUT/wal_adler.json 2> $OUT/wal_adler.err; tail -c 800 $OUT/wal_adler.json
timeout 300 python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 2 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("tick", round(d["roofline"]["avg_launch_us"],2), "us", round(d["value"]/1e9,2), "G/s frac", round(d["roofline"]["frac"],3), d["config"]["state_checksum"])
PY
