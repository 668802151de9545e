#!/bin/bash
# the round's last tree on a GPU box: the whole -m gpu suite, smoke, the driver's exact command, the forced single-rank
# RCCL line (the multi-GPU path with one rank: communicator, all-gather between the launches), the GPU fuzz of fused rounds
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/${1:-r06final}; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/timing.txt; }
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/gpu_suite.txt; stamp suite
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/gpu_suite.txt; stamp smoke
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; stamp bench_driver
RGB_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --no-host-path --literal-ticks 0 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; stamp force_dist
timeout 600 python tools/gpu_fuzz_rounds.py 40 12 2>&1 | tail -6 | tee $OUT/gpu_fuzz_rounds.txt; stamp fuzz
python - $OUT <<'PY'
import json, sys
for name in ("bench_driver_form", "bench_force_dist"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{name}.json").read().strip().splitlines()[-1]); r = d["roofline"]
        print(name, round(d["ms_per_step"] * 1e3, 2), "us/step", round(d["value"] / 1e9, 3), "G/s frac", round(r["frac"], 4), "avg_tick_us", round(r["avg_tick_us"], 2),
              "n_gpus", d["n_gpus"], "literal", {k: round(v["frac"], 4) for k, v in (d.get("literal_configs") or {}).items() if isinstance(v, dict) and "frac" in v})
    except Exception as e:
        print(name, "FAILED", e); print(open(f"{sys.argv[1]}/{name}.err").read()[-1500:])
PY
