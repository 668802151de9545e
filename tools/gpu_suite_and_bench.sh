#!/bin/bash
# the whole -m gpu suite, then the driver's bench command and the default one (their lines under gpurun_out/TAG/):
#   gpurun --timeout 1500 -- 'bash tools/gpu_suite_and_bench.sh TAG'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; TAG=${1:-suite}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/timing.txt; }
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/gpu_suite.txt; stamp suite
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/gpu_suite.txt; stamp smoke
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; stamp bench_driver
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; stamp bench
python - $OUT <<'PY'
import json, sys
for name in ("bench_driver_form", "bench"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{name}.json").read().strip().splitlines()[-1]); r = d["roofline"]
        print(name, round(d["ms_per_step"] * 1e3, 2), "us/step", round(d["value"] / 1e9, 3), "G/s frac", round(r["frac"], 4), "avg_tick_us", round(r["avg_tick_us"], 2),
              "wall-events us", d.get("wall_minus_events_us"))
        for k in ("host_path", "literal_configs"):
            print("   ", k, json.dumps(d.get(k))[:1800])
        print("    train", json.dumps(d["config"]["train"])[:1200])
    except Exception as e:
        print(name, "FAILED", e); print(open(f"{sys.argv[1]}/{name}.err").read()[-1500:])
PY
