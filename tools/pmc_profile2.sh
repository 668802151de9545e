#!/bin/bash
# instruction-side PMC passes (icache, ifetch, issue stalls) for the tick kernel
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 48 --warmup 16 --no-cpu-baseline --check-ticks 0 --no-graph"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL --output-format csv -d $OUT/sqc -o p -- $CMD > $OUT/sqc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
