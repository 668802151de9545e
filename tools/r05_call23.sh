#!/bin/bash
# round 5, call 23: BASELINE configs[4] (65 536 x 7 repair) -- the leader-side slices of 32 with the 192-byte peers rows in
# LDS (wide7) against HEAD (head7), same box, the bench's literal config 5 on its own (tools/cfg5_probe.py), twice each;
# + the N = 7 train tests on the GPU with the variant
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r05w; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
for i in 1 2; do
  for v in head7 wide7; do
    RGB_LIB=$V/$v.so timeout 300 python tools/cfg5_probe.py 5 32 > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    echo "$v $i: $(tail -1 $OUT/${v}_$i.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["us_per_tick"],2), "us/tick frac", round(d["frac"],4), "per-tick", round(d["per_tick_launches"]["us_per_tick"],2), "train", round(d["train_launch"]["us_per_tick"],2), "state", d["final_state_equal"], "checked", d["oracle_checked_decisions"])' 2>&1 | tail -1)" | tee -a $OUT/summary.txt
  done
done
RGB_LIB=$V/wide7.so timeout 300 python -m pytest tests/test_train.py -m gpu -x -q -p no:cacheprovider -k "1024-7 or repair" > $OUT/pytest.txt 2>&1
echo "pytest N=7 train tests (wide7): $(tail -1 $OUT/pytest.txt)" | tee -a $OUT/summary.txt
