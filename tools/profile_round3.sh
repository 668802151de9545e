#!/bin/bash
# Evidence run for profiles/ (round 3):  gpurun --timeout 1500 -- 'bash tools/profile_round3.sh r03'
# default bench, the driver's short form (train and per-tick launches), rocprofv3 kernel trace + stats, PMC passes
# (FETCH_SIZE, WRITE_SIZE, the SQ set: each in its own run, never with a trace domain besides --kernel-trace), the
# train timeline, and the same for the literal SURVEY 8(d) configurations 3 and 5 (rgb_tick_classes_kernel<5>/<7>).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --check-ticks 0"
# kernel trace + stats of the driver's command (trains of 11 + 9 ticks in the timed region, 5 warm-up ticks)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 $Q --literal-ticks 0 > $OUT/stats.log 2>&1
# a longer run for the per-launch average: 12 trains of 16 ticks, eager launches
CMDS="python $R/bench.py --steps 192 --warmup 16 $Q --literal-ticks 0 --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o $TAG -- $CMDS > $OUT/stats_long.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMDS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMDS > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o $TAG -- $CMDS > $OUT/pmc_sq.log 2>&1
# the traffic figure the bench lines carry comes from the PMC passes above (same box, same call)
python $R/tools/make_traffic_json.py $OUT 16 > $OUT/traffic.json 2> $OUT/traffic.err
export RGB_TRAFFIC_JSON=$OUT/traffic.json
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
python bench.py --steps 20 --warmup 5 --launch tick --no-cpu-baseline --no-host-path --literal-ticks 0 > $OUT/bench_driver_form_tick.json 2> $OUT/bench_driver_form_tick.err
cd /tmp
# the literal configurations (per-tick class kernels: <5> for configs 2/3, <7> for config 5), headline skipped quickly
LIT="python $R/bench.py --steps 4 --warmup 2 --age 0 $Q --literal-ticks 32 --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lit_stats -o $TAG -- $LIT > $OUT/lit_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/lit_pmc_fetch -o $TAG -- $LIT > $OUT/lit_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/lit_pmc_write -o $TAG -- $LIT > $OUT/lit_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/lit_pmc_sq -o $TAG -- $LIT > $OUT/lit_pmc_sq.log 2>&1
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
[ -f ra_amd/csrc/variants/timeline.so ] && RGB_LIB=$R/ra_amd/csrc/variants/timeline.so TL_AGE=512 TL_TICKS=32 timeout 300 python tools/train_timeline.py > $OUT/train_timeline.txt 2>&1
python - <<PY
import json, csv, glob
for name in ("bench", "bench_driver_form", "bench_driver_form_tick"):
    try:
        d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(name, round(d["ms_per_step"]*1e3,2), "us/step", round(d["value"]/1e9,3), "G/s frac", round(r["frac"],4), r["kernel"], "avg_launch_us", round(r["avg_launch_us"],1))
    except Exception as e:
        print(name, "FAILED", e)
for f in sorted(glob.glob("$OUT/*stats*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "tick" in r["Name"] or "train" in r["Name"]: print(f.split("/")[-3], r["Name"][:70], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
grep -E "train_kernel|tick_classes" $OUT/pmc_summary.txt | cut -c1-170
