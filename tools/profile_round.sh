#!/bin/bash
# One-shot evidence run for profiles/: default bench, rocprofv3 kernel stats, HBM PMC passes.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --check-ticks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $OUT/stats.log 2>&1
CMDS="python $REPO/bench.py --steps 64 --warmup 16 --no-cpu-baseline --check-ticks 0 --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMDS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMDS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o $TAG -- $CMDS > $OUT/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wal_stats -o ${TAG}_wal -- python $REPO/tools/wal_bench.py > $OUT/wal_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_wal_fetch -o ${TAG}_wal -- python $REPO/tools/wal_bench.py > $OUT/pmc_wal_fetch.log 2>&1
cat $OUT/bench.json
head -4 $OUT/wal_stats/${TAG}_wal_kernel_stats.csv
head -6 $OUT/stats/${TAG}_kernel_stats.csv
python $REPO/tools/pmc_summary.py $OUT | grep -E "classes|wal|==" 
