#!/bin/bash
# Evidence run for profiles/ (round 2):  gpurun --timeout 1500 -- 'bash tools/profile_round2.sh r02'
# default bench (what the driver runs), the driver's short form, rocprofv3 kernel trace + stats of the default
# bench, PMC passes (FETCH_SIZE, WRITE_SIZE, the SQ set) each in its own run, per-class wave timeline.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $OUT/stats.log 2>&1
CMDS="python $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-host-path --literal-ticks 0 --check-ticks 0 --no-graph"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMDS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMDS > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o $TAG -- $CMDS > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o $TAG -- $CMDS > $OUT/pmc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
TL_TICKS=512 timeout 300 python tools/wave_timeline.py > $OUT/wave_timeline.txt 2>&1
python - <<PY
import json, csv, glob
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["ms_per_step"]*1e3,2), "us/tick", round(d["value"]/1e9,3), "G/s frac", round(d["roofline"]["frac"],4))
d2 = json.loads(open("$OUT/bench_driver_form.json").read().strip().splitlines()[-1])
print("driver form (--steps 20 --warmup 5)", round(d2["ms_per_step"]*1e3,2), "us/tick frac", round(d2["roofline"]["frac"],4))
for f in glob.glob("$OUT/stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "tick_classes" in r["Name"]: print("rocprof avg ns", r["AverageNs"], "calls", r["Calls"])
PY
grep -E "classes" $OUT/pmc_summary.txt
grep -E "^class|^waves" $OUT/wave_timeline.txt | cut -c1-200
