#!/bin/bash
# round 4, last GPU call (a few minutes of box time): (1) the new parity tests on the device, (2) same-box A/B of the
# train kernel with and without the round's dispatcher change (variants/new.so, variants/head.so), (3) the small-record
# WAL framing kernel: funnel form (product) against the direct form (variants/wal_direct.so), parity of the latter.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04f; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
timeout 170 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "wal_down or reference_vector" > $OUT/pytest_new.txt 2>&1
stamp "pytest new: $(grep -E 'passed|failed|rror' $OUT/pytest_new.txt | tail -1)"
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5 --steps 20 --warmup 5"
one() { # name lib
  RGB_LIB=$V/$2.so timeout 100 python bench.py $Q > $OUT/$1.json 2> $OUT/$1.err
  python - $OUT/$1.json $1 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:16s} {r.get('avg_tick_us', r['avg_launch_us']):7.2f} us/tick by events, frac {r['frac']:.4f}, ms/step {d['ms_per_step']:.5f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
one drv_head_1 head; one drv_new_1 new
stamp bench-1
for lib in "" wal_direct; do
  RGB_LIB=${lib:+$V/$lib.so} WAL_CASES="256 B,40-320" timeout 60 python tools/wal_frame_bench.py 2> $OUT/wal_${lib:-product}.err | tee $OUT/wal_${lib:-product}.txt | \
    python -c "import sys, json; [print('wal ${lib:-product}', d['workload'], round(d['us_per_launch'], 1), 'us', round(d['frac_of_8TBps'], 4)) for d in map(json.loads, sys.stdin)]" | tee -a $OUT/summary.txt
done
stamp wal-ab
RGB_LIB=$V/wal_direct.so timeout 120 python -m pytest tests/test_wal_framing.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_wal_direct.txt 2>&1
stamp "pytest wal direct: $(grep -E 'passed|failed|rror' $OUT/pytest_wal_direct.txt | tail -1)"
one drv_new_2 new; one drv_head_2 head
stamp bench-2
for lib in wal_direct ""; do
  RGB_LIB=${lib:+$V/$lib.so} WAL_CASES="256 B" timeout 60 python tools/wal_frame_bench.py 2>> $OUT/wal_${lib:-product}.err | \
    python -c "import sys, json; [print('wal ${lib:-product} (2)', d['workload'], round(d['us_per_launch'], 1), 'us', round(d['frac_of_8TBps'], 4)) for d in map(json.loads, sys.stdin)]" | tee -a $OUT/summary.txt
done
stamp wal-ab-2
timeout 150 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
stamp "full driver-form bench rc=$?"
