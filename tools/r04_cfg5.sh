#!/bin/bash
# literal configs (config 5 = 65 536 x 7 repair) + the train GPU tests + the driver's form of the headline (eager)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 300 python -m pytest tests/test_train.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_train.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest_train.txt | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --check-ticks 3 --literal-ticks 32 > $OUT/lit.json 2> $OUT/lit.err
python - <<PY
import json
d=json.loads(open("$OUT/lit.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("headline driver form", round(r["avg_tick_us"],2), "us/tick frac", round(r["frac"],4), "graph", d["config"]["hip_graph"])
for k,v in d["literal_configs"].items():
    print(k, "us/tick", round(v["us_per_tick"],2), "frac", round(v["frac"],4), v["launch"], "per-tick", round(v["per_tick_launches"]["us_per_tick"],2), "train", round(v["train_launch"]["us_per_tick"],2), v["final_state_equal"])
PY
