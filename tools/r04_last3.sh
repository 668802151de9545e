#!/bin/bash
# round 4, last GPU seconds: the final build's WAL kernels and wal_down tests on the device, smoke(), the rocprofv3
# kernel trace of the WAL framing bench, the framing bench itself, and the driver's-form bench line.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04h; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
timeout 60 python -m pytest tests/test_wal_framing.py tests/test_wal_checksum.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "wal or W2" > $OUT/pytest.txt 2>&1
stamp "pytest wal + wal_down: $(grep -E 'passed|failed|rror' $OUT/pytest.txt | tail -1)"
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_wal -o wal -- python $R/tools/wal_frame_bench.py > $R/$OUT/prof_wal.log 2>&1 )
find $OUT/prof_wal -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/wal_kernel_stats.csv
find $OUT/prof_wal -type f ! -name "*stats.csv" -delete 2>/dev/null
stamp "rocprof wal: $(grep -c . $OUT/wal_kernel_stats.csv 2>/dev/null) lines"
timeout 40 python tools/wal_frame_bench.py 2> $OUT/wal.err | tee $OUT/wal.jsonl | python -c "import sys, json; [print('wal', d['workload'], round(d['us_per_launch'], 1), 'us', round(d['frac_of_8TBps'], 4)) for d in map(json.loads, sys.stdin)]" | tee -a $OUT/summary.txt
cp gpurun_out/wal_frame_bench.json $OUT/wal_frame.json 2>/dev/null
stamp wal-bench
RGB_TRAFFIC_JSON=$R/profiles/r04_traffic.json timeout 75 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
stamp "driver-form bench rc=$?"
