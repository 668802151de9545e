#!/bin/bash
set -u
TAG=${1:-r02h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
# per-class wave timeline: bench stream (16) and benign traffic + fast paths (16|128)
RGB_LIB=$R/ra_amd/csrc/variants/proffast.so TL_DBG=16 TL_TICKS=300 timeout 300 python tools/wave_timeline.py > $OUT/tl_fast.txt 2>&1
RGB_LIB=$R/ra_amd/csrc/variants/proffast.so TL_DBG=144 TL_TICKS=300 timeout 300 python tools/wave_timeline.py > $OUT/tl_benign_fast.txt 2>&1
grep -E "^class|^waves|t= " $OUT/tl_fast.txt | cut -c1-230
echo ---- benign + fast
grep -E "^class|^waves|t= " $OUT/tl_benign_fast.txt | cut -c1-230
cd /tmp && export TMPDIR=/tmp
CMDS="python $R/bench.py --steps 48 --warmup 32 --age 300 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 0 --no-graph"
RGB_LIB=$R/ra_amd/csrc/variants/fast.so timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/$OUT/pmc_icache -o p -- $CMDS > $R/$OUT/pmc_icache.log 2>&1
RGB_LIB=$R/ra_amd/csrc/variants/fast.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $R/$OUT/pmc_if -o p -- $CMDS > $R/$OUT/pmc_if.log 2>&1
cd $R
python tools/pmc_summary.py $OUT 2>&1 | grep -E "==|classes"
tail -3 $OUT/pmc_icache.log
