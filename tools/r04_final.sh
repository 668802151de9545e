#!/bin/bash
# final check of HEAD: the whole GPU suite, then the bench lines that go to profiles/ (default run, driver's form)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04z; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest_full.txt | tail -3
export RGB_TRAFFIC_JSON=$R/profiles/r04_traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python - <<PY
import json
for name in ("bench", "bench_driver_form"):
    d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(name, round(d["ms_per_step"]*1e3,2), "us/step wall", round(r["avg_tick_us"],2), "by events", round(d["value"]/1e9,3), "G/s frac", round(r["frac"],4), r["kernel"], "graph", d["config"]["hip_graph"])
    lc = d["literal_configs"]
    print("   ", {k: (round(v["us_per_tick"],2), round(v["frac"],4)) for k, v in lc.items()})
    hp = d["host_path"]; print("    host", round(hp["value"]/1e6,1), round(hp["threads4"]["value"]/1e6,1), round(hp["rounds4"]["fused_train"]["value"]/1e6,1), round(hp["rounds4"]["launch_per_round"]["value"]/1e6,1))
    print("    wal256", round(d["aux_kernels"]["wal_frame_256"]["frac"],3), "cpu", round(d["cpu_baseline"]["one_thread"]/1e6,2), round(d["cpu_baseline"]["all_cores"]/1e6,1))
PY
