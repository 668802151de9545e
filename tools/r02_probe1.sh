#!/bin/bash
# Round-2 measurement call:  gpurun --timeout 1500 -- 'bash tools/r02_probe1.sh TAG'
#  1. the whole -m gpu suite on the product library        2. product / variants / profiling knobs, same command
#  3. per-class wave timeline (profiling build)            4. rocprofv3 kernel trace + PMC passes for profiles/
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "${SKIP_PYTEST:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -n 3 $OUT/pytest_gpu.log
fi
STEPS=${STEPS:-300} WARM=${WARM:-300} bash tools/knob_sweep.sh $TAG "${KNOBS:-1 2 3 8}" 2>&1 | tee $OUT/sweep.txt
if [ -f ra_amd/csrc/libra_gpu_batch_prof.so ] && [ "${SKIP_TIMELINE:-0}" != "1" ]; then
  TL_TICKS=400 timeout 300 python tools/wave_timeline.py > $OUT/wave_timeline.txt 2>&1
  grep -E "^class|^waves|t= " $OUT/wave_timeline.txt
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  R=$PWD
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -o $TAG -- \
      python $R/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-host-path --check-ticks 0 > $R/$OUT/stats.log 2>&1
  CMDS="python $R/bench.py --steps 48 --warmup 400 --no-cpu-baseline --no-host-path --check-ticks 0 --no-graph"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$OUT/pmc_sq -o $TAG -- $CMDS > $R/$OUT/pmc_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -o $TAG -- $CMDS > $R/$OUT/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -o $TAG -- $CMDS > $R/$OUT/pmc_write.log 2>&1
  cd $R
  head -5 $OUT/stats/*kernel_stats.csv
  python tools/pmc_summary.py $OUT 2>&1 | tail -20
fi
