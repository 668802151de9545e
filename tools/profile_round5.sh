#!/bin/bash
# Evidence run for profiles/ (round 5), ONE gpurun call on ONE commit:
#   gpurun --timeout 1100 -- 'bash tools/profile_round5.sh r05 <commit>'
# kernel trace + stats of the driver's command; FETCH_SIZE / WRITE_SIZE / SQ passes of rgb_train_dealt_kernel<5> at the
# driver's launch length (20 ticks) AND at 240 ticks, and of <7> on the literal config 5 (each counter set in its own run,
# never with a trace domain besides --kernel-trace; every profiler run under its own timeout: rocprofv3 has hung at exit
# on this pool); then the bench lines (default, driver's form, device-built plan, persistent form, one launch per tick),
# the train timeline and the decline histogram.
set -u
TAG=${1:-r05}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
echo "$COMMIT" > $OUT/commit.txt
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/timing.txt; }
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0"
P="timeout 170 rocprofv3 --kernel-trace --output-format csv"
$P --stats -d $OUT/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 $Q > $OUT/stats.log 2>&1; stamp stats
C240="python $R/bench.py --steps 240 --warmup 240 $Q --no-graph"
C20="python $R/bench.py --steps 20 --warmup 20 $Q --no-graph"
$P --pmc FETCH_SIZE -d $OUT/pmc_fetch_240 -o $TAG -- $C240 > $OUT/pmc_fetch_240.log 2>&1
$P --pmc WRITE_SIZE -d $OUT/pmc_write_240 -o $TAG -- $C240 > $OUT/pmc_write_240.log 2>&1
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_sq_240 -o $TAG -- $C240 > $OUT/pmc_sq_240.log 2>&1
stamp pmc240
$P --pmc FETCH_SIZE -d $OUT/pmc_fetch_20 -o $TAG -- $C20 > $OUT/pmc_fetch_20.log 2>&1
$P --pmc WRITE_SIZE -d $OUT/pmc_write_20 -o $TAG -- $C20 > $OUT/pmc_write_20.log 2>&1
stamp pmc20
# literal config 5 (65 536 x 7, repair) as a train of rgb_train_dealt_kernel<7>: the headline shortened to nothing
LIT="python $R/bench.py --steps 4 --warmup 2 --age 0 --no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 32 --no-graph"
$P --stats -d $OUT/stats_lit -o $TAG -- $LIT > $OUT/stats_lit.log 2>&1
$P --pmc FETCH_SIZE -d $OUT/pmc_fetch_lit -o $TAG -- $LIT > $OUT/pmc_fetch_lit.log 2>&1
$P --pmc WRITE_SIZE -d $OUT/pmc_write_lit -o $TAG -- $LIT > $OUT/pmc_write_lit.log 2>&1
stamp pmc_lit
cd $R
python tools/make_traffic_json5.py $OUT $COMMIT > $OUT/traffic.json 2> $OUT/traffic.err
RGB_TRAFFIC_TICKS=20 python tools/make_traffic_json5.py $OUT $COMMIT > $OUT/traffic20.json 2>> $OUT/traffic.err
RGB_TRAFFIC_JSON=$OUT/traffic.json timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; stamp bench
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; stamp bench_driver
B="--steps 20 --warmup 5 --no-cpu-baseline --no-host-path --literal-ticks 0"
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B --device-plan > $OUT/bench_driver_form_device_plan.json 2> $OUT/bench_driver_form_device_plan.err
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B --train-form persistent > $OUT/bench_driver_form_persistent.json 2> $OUT/bench_driver_form_persistent.err
timeout 120 python bench.py $B --launch tick > $OUT/bench_driver_form_tick.json 2> $OUT/bench_driver_form_tick.err
RGB_TRAFFIC_JSON=$OUT/traffic.json timeout 120 python bench.py --steps 1000 --warmup 32 --no-cpu-baseline --no-host-path --literal-ticks 0 --device-plan > $OUT/bench_device_plan.json 2> $OUT/bench_device_plan.err
stamp bench_forms
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
[ -f ra_amd/csrc/variants/timeline.so ] && RGB_LIB=$R/ra_amd/csrc/variants/timeline.so TL_HINT=2 TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/train_timeline.txt 2>&1
[ -f ra_amd/csrc/variants/hist.so ] && RGB_LIB=$R/ra_amd/csrc/variants/hist.so timeout 200 python tools/train_decline_hist.py > $OUT/train_decline_hist.txt 2>&1
stamp tools
python - <<PY
import json, csv, glob
for name in ("bench", "bench_driver_form", "bench_driver_form_device_plan", "bench_driver_form_persistent", "bench_driver_form_tick", "bench_device_plan"):
    try:
        d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(name, round(d["ms_per_step"]*1e3,2), "us/step", round(d["value"]/1e9,3), "G/s frac", round(r["frac"],4), r["kernel"], "avg_tick_us", round(r["avg_tick_us"],2), "tpl", r["ticks_per_launch"], "traffic/tick MB", round((r["traffic"] or 0)/r["ticks_per_launch"]/1e6, 2), "wall-events us", d.get("wall_minus_events_us"))
        if name in ("bench", "bench_driver_form"):
            for k in ("host_path", "literal_configs", "aux_kernels", "cpu_baseline"):
                print("   ", k, json.dumps(d.get(k))[:700])
            print("    train", json.dumps(d["config"]["train"])[:900])
    except Exception as e:
        print(name, "FAILED", e)
for f in sorted(glob.glob("$OUT/stats*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "tick" in r["Name"] or "train" in r["Name"] or "leaderboard" in r["Name"]: print(f.split("/")[-3], r["Name"][:70], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
cat $OUT/traffic.json | head -60
