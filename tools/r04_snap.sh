#!/bin/bash
# round 4: leaderboard snapshots inside the train -- GPU parity tests, then the driver's form and the long form with the
# snapshots as rows of the launch (default) and with the snapshot kernel between one-period launches (--snapshot-kernel)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout 400 python -m pytest tests/test_train.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_train.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest_train.txt | tail -3
Q="--no-cpu-baseline --no-host-path --literal-ticks 0"
one() { # name args
  timeout 300 python bench.py $Q $2 > $OUT/$1.json 2> $OUT/$1.err
  python - $OUT/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:24s} {r['avg_tick_us']:7.2f} us/tick frac {r['frac']:.4f} ms/step {d['ms_per_step']:.5f} tpl {r['ticks_per_launch']} snaps {d['config']['train'].get('leaderboard_snapshots_compared_with_snapshot_kernel')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
}
for rep in 1 2; do
  one drv_intrain_$rep "--steps 20 --warmup 5"
  one drv_kernel_$rep "--steps 20 --warmup 5 --snapshot-kernel"
done 2>&1 | tee $OUT/summary.txt
one long_intrain "--steps 192 --warmup 16" | tee -a $OUT/summary.txt
one long_kernel "--steps 192 --warmup 16 --snapshot-kernel" | tee -a $OUT/summary.txt
