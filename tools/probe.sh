#!/bin/bash
# One measurement round on the GPU box:  gpurun --timeout 900 -- 'bash tools/probe.sh TAG [pytest-args]'
# parity subset, steady-state bench (400 warm-up ticks + 400 timed), per-class wave timeline (profiling build).
set -u
TAG=${1:-p}
shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "${1:-}" != "nopytest" ]; then
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -n 3 $OUT/pytest_gpu.log
fi
timeout 300 python bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-host-path --check-ticks 2 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("$TAG", round(d["ms_per_step"] * 1e3, 2), "us/tick", round(d["value"] / 1e9, 2), "G decisions/s",
          "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print("$TAG bench failed:", e); print(open("$OUT/bench.err").read()[-2000:])
PY
if [ -f ra_amd/csrc/libra_gpu_batch_prof.so ]; then
  TL_TICKS=400 timeout 300 python tools/wave_timeline.py > $OUT/wave_timeline.txt 2>&1
  grep -E "^class|^waves" $OUT/wave_timeline.txt
fi
