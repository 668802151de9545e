#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box; counters in separate runs, no tracing
# domains besides --kernel-trace).  Output: gpurun_out/pmc_<tag>/*counter_collection.csv
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 48 --warmup 16 --no-cpu-baseline --check-ticks 0 --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o p -- $CMD > $OUT/l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
ls -R $OUT | head -30
