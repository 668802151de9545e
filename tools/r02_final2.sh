#!/bin/bash
# FETCH_SIZE and L2 request counters of the product vs the RGB_X_EXTRA_FETCH variants (is a 64-byte gather half the
# fabric bytes of a 128-byte one, or does the L2 fill whole 128-byte lines?)
set -u
TAG=r02f2; R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMDS="python $R/bench.py --steps 48 --warmup 16 --age 256 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 0 --no-graph"
for v in product xf1 xf2; do
  lib=$R/ra_amd/csrc/variants/$v.so; [ $v = product ] && lib=$R/ra_amd/csrc/libra_gpu_batch.so
  RGB_LIB=$lib timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -o p -- $CMDS > $OUT/fetch_$v.log 2>&1
  RGB_LIB=$lib timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$v -o p -- $CMDS > $OUT/tcc_$v.log 2>&1
done
cd $R; python tools/pmc_summary.py $OUT 2>&1 | grep -E "^==|classes" | tee $OUT/summary.txt
tail -3 $OUT/tcc_product.log
