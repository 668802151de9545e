#!/bin/bash
# Evidence run for profiles/ (round 4):  gpurun --timeout 900 -- 'bash tools/profile_round4.sh r04'
# kernel trace + stats of the driver's command, PMC passes of 240-tick launches (FETCH_SIZE, WRITE_SIZE, the SQ set: each
# in its own run, never with a trace domain besides --kernel-trace), the bench lines (default, driver's form, driver's
# form with one launch per tick), the train timeline.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 $Q > $OUT/stats.log 2>&1
# two launches of 240 ticks (15 leaderboard periods each, the snapshots as rows of the launch), eager
CMDS="python $R/bench.py --steps 240 --warmup 240 $Q --no-graph"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMDS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMDS > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o $TAG -- $CMDS > $OUT/pmc_sq.log 2>&1
python $R/tools/make_traffic_json.py $OUT 240 > $OUT/traffic.json 2> $OUT/traffic.err
export RGB_TRAFFIC_JSON=$OUT/traffic.json
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
python bench.py --steps 20 --warmup 5 --launch tick --no-cpu-baseline --no-host-path --literal-ticks 0 > $OUT/bench_driver_form_tick.json 2> $OUT/bench_driver_form_tick.err
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
[ -f ra_amd/csrc/variants_tools/timeline.so ] && RGB_LIB=$R/ra_amd/csrc/variants_tools/timeline.so TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/train_timeline.txt 2>&1
python - <<PY
import json, csv, glob
for name in ("bench", "bench_driver_form", "bench_driver_form_tick"):
    try:
        d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(name, round(d["ms_per_step"]*1e3,2), "us/step", round(d["value"]/1e9,3), "G/s frac", round(r["frac"],4), r["kernel"], "avg_launch_us", round(r["avg_launch_us"],1), "tpl", r["ticks_per_launch"], "traffic/tick MB", round((r["traffic"] or 0)/r["ticks_per_launch"]/1e6, 2))
        for k in ("host_path", "literal_configs", "aux_kernels", "cpu_baseline"):
            print("   ", k, json.dumps(d.get(k))[:600])
    except Exception as e:
        print(name, "FAILED", e)
for f in sorted(glob.glob("$OUT/*stats*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "tick" in r["Name"] or "train" in r["Name"] or "leaderboard" in r["Name"]: print(f.split("/")[-3], r["Name"][:70], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
grep -E "train_|tick_classes" $OUT/pmc_summary.txt | cut -c1-150
cat $OUT/traffic.json | head -20
