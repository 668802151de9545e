#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r02_probe2.sh TAG "variant1 variant2"'   -- sweep + HBM write/fetch PMC per variant
set -u
TAG=${1:-r02b}; PMCV=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
STEPS=${STEPS:-300} WARM=${WARM:-300} bash tools/knob_sweep.sh $TAG "${KNOBS:-}" 2>&1 | tee $OUT/sweep.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in $PMCV; do
  lib=$R/ra_amd/csrc/variants/$v.so; [ "$v" = product ] && lib=$R/ra_amd/csrc/libra_gpu_batch.so
  CMDS="python $R/bench.py --steps 48 --warmup 400 --no-cpu-baseline --no-host-path --check-ticks 0 --no-graph"
  RGB_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write_$v -o p -- $CMDS > $R/$OUT/pmc_write_$v.log 2>&1
  RGB_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch_$v -o p -- $CMDS > $R/$OUT/pmc_fetch_$v.log 2>&1
done
cd $R
python tools/pmc_summary.py $OUT 2>&1 | grep -E "==|classes"
