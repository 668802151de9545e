#!/bin/bash
# round 5, call 25: where rgb_train_dealt_kernel<5> stalls -- six PMC passes (each its own run, --kernel-trace only) of the
# 240-tick closed-loop launches: L2 (requests, hits, tag stalls, fabric requests by size), the vector L1 (requests, their
# latency to the L2), address / translation units, the instruction cache, instruction mix and in-flight levels
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RGB_LIB=$R/ra_amd/csrc/variants/pro5.so
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/timing.txt; }
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0 --members 5"
C="python $R/bench.py --steps 240 --warmup 240 $Q --no-graph"
P="timeout 170 rocprofv3 --kernel-trace --output-format csv"
run() { # name counters...
  local name=$1; shift
  $P --pmc "$@" -d $OUT/pmc_$name -o r05 -- $C > $OUT/pmc_$name.log 2>&1; stamp $name
}
run tcc   TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
run tcc2  TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum
run tcp   TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
run ta    TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run sqc   SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
run sq    SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep "train_dealt_kernel<5>" $OUT/pmc_summary.txt
