#!/bin/bash
# round 4, the closing GPU call: the whole GPU suite on the product build, then the WAL framing forms per lane-group size
# (variants/wal_d0 = funnel everywhere, product = direct for eight lanes, wal_d2 = + sixteen, wal_d3 = every size),
# GPU parity of wal_d3, and the rocprofv3 kernel trace of the small-record framing bench.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/r04g; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
stamp start
timeout 165 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_full.txt 2>&1
stamp "pytest full: $(grep -E 'passed|failed|rror' $OUT/pytest_full.txt | tail -1)"
wal() { # lib cases tag
  RGB_LIB=${1:+$V/$1.so} WAL_CASES="$2" timeout 60 python tools/wal_frame_bench.py 2>> $OUT/wal_${1:-product}.err | tee -a $OUT/wal_${1:-product}.txt | \
    python -c "import sys, json; [print('wal ${1:-product} $3', d['workload'], round(d['us_per_launch'], 1), 'us', round(d['frac_of_8TBps'], 4)) for d in map(json.loads, sys.stdin)]" | tee -a $OUT/summary.txt
}
ALL="4 KiB,1-16 KiB,256 B,40-320,400-1000"
wal wal_d0 "$ALL" 1; wal wal_d3 "$ALL" 1; wal wal_d2 "400-1000" 1; wal "" "256 B,40-320" 1
stamp wal-1
RGB_LIB=$V/wal_d3.so timeout 60 python -m pytest tests/test_wal_framing.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_wal_d3.txt 2>&1
stamp "pytest wal_d3: $(grep -E 'passed|failed|rror' $OUT/pytest_wal_d3.txt | tail -1)"
wal wal_d3 "4 KiB,1-16 KiB,400-1000" 2; wal wal_d0 "4 KiB,1-16 KiB,400-1000" 2
stamp wal-2
cd /tmp && export TMPDIR=/tmp
WAL_CASES="256 B,40-320" timeout 60 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_wal -- python $R/tools/wal_frame_bench.py > $R/$OUT/prof_wal.log 2>&1
cd $R; find $OUT/prof_wal -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/wal_kernel_stats.csv
find $OUT/prof_wal -type f ! -name "*stats.csv" -delete 2>/dev/null
stamp "rocprof wal: $(head -3 $OUT/wal_kernel_stats.csv 2>/dev/null | tail -2 | cut -c1-160)"
