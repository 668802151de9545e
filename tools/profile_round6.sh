#!/bin/bash
# Evidence run for profiles/ (round 6), ONE gpurun call on ONE commit:
#   gpurun --timeout 2400 -- 'bash tools/profile_round6.sh r06 <commit>'
# the whole -m gpu suite + smoke; kernel trace + stats of the driver's command; FETCH_SIZE / WRITE_SIZE / fabric requests by
# size / L2 / L1 / SQ passes of rgb_train_dealt_kernel<5> at the driver's launch length (20 ticks) and at 240 ticks, and of
# <7> on the literal config 5 (each counter set in its own run -- at most four counters of a block per pass, only the train
# kernels counted -- never with a trace domain besides --kernel-trace, every profiler run under its own timeout); the
# calibration probes; then the bench lines (default, driver's form, host-built plan, plan inside the region, persistent
# form, one launch per tick), the train timeline and the decline histogram.
set -u
TAG=${1:-r06}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
echo "$COMMIT" > $OUT/commit.txt
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/timing.txt; }
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/gpu_suite.txt; stamp suite
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/gpu_suite.txt; stamp smoke
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0"
P="timeout 170 rocprofv3 --kernel-trace --output-format csv"
$P --stats -d $OUT/stats -o $TAG -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $Q > $OUT/stats.log 2>&1; stamp stats
PK="$P --kernel-include-regex rgb_train"
C240="python $R/bench.py --steps 240 --warmup 240 $Q --no-graph"
C20="python $R/bench.py --steps 20 --warmup 20 $Q --no-graph"
LIT="python $R/bench.py --steps 4 --warmup 2 --age 0 --no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 32 --no-graph"
pass() { # tag command counters...
  local tag=$1 cmd=$2; shift 2
  $PK --pmc "$@" -d $OUT/pmc_$tag -o $TAG -- $cmd > $OUT/pmc_$tag.log 2>&1
}
for L in 240 20 lit; do
  case $L in 240) C="$C240";; 20) C="$C20";; lit) C="$LIT";; esac
  pass fetch_$L "$C" FETCH_SIZE
  pass write_$L "$C" WRITE_SIZE
  pass ea_$L "$C" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum
  pass rdsz_$L "$C" TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
  stamp pmc_$L
done
pass tcc_20 "$C20" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass tcc2_20 "$C20" TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
pass tcp_20 "$C20" TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
pass tcp2_20 "$C20" TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
pass sq_240 "$C240" SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
stamp pmc_units
$P --stats -d $OUT/stats_lit -o $TAG -- $LIT > $OUT/stats_lit.log 2>&1; stamp stats_lit
# calibration probes: counters per request shape against the bytes moved
PC="timeout 60 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex _kernel"
for k in rd128 rd64 rd32 rd1 wr16 wr32 wr64nt wr32nt wr128; do
  $PC --pmc FETCH_SIZE -d $OUT/cal_fetch_$k -o $TAG -- $R/tools/probes/traffic_calib $k > $OUT/cal_fetch_$k.log 2>&1
  $PC --pmc WRITE_SIZE -d $OUT/cal_write_$k -o $TAG -- $R/tools/probes/traffic_calib $k > $OUT/cal_write_$k.log 2>&1
  $PC --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/cal_ea_$k -o $TAG -- $R/tools/probes/traffic_calib $k > $OUT/cal_ea_$k.log 2>&1
done
stamp calibration
cd $R
python - $OUT > $OUT/calibration.json <<'PY'
import csv, glob, json, sys
out = sys.argv[1]; res = {}
for k in "rd128 rd64 rd32 rd1 wr16 wr32 wr64nt wr32nt wr128".split():
    line = {}
    try: line["bytes_moved"] = json.loads(open(f"{out}/cal_fetch_{k}.log").read().strip().splitlines()[-1])["bytes"]
    except Exception: line["bytes_moved"] = None
    for what in ("fetch", "write", "ea"):
        for f in glob.glob(f"{out}/cal_{what}_{k}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "fill" in row["Kernel_Name"]: continue
                line[row["Counter_Name"]] = line.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    if line.get("bytes_moved"):
        if line.get("FETCH_SIZE"): line["fetch_size_factor"] = round(line["bytes_moved"] / (line["FETCH_SIZE"] * 1024), 3)
        if line.get("WRITE_SIZE"): line["write_size_factor"] = round(line["bytes_moved"] / (line["WRITE_SIZE"] * 1024), 3)
    res[k] = line
print(json.dumps({"what": "tools/probes/traffic_calib.hip: one request shape per kernel, 2 M units over a 512 MiB table; factor = bytes moved / counter bytes", "shapes": res}, indent=1))
PY
python tools/make_traffic_json6.py $OUT $COMMIT > $OUT/traffic.json 2> $OUT/traffic.err
RGB_TRAFFIC_TICKS=20 python tools/make_traffic_json6.py $OUT $COMMIT > $OUT/traffic20.json 2>> $OUT/traffic.err
RGB_TRAFFIC_JSON=$OUT/traffic.json timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; stamp bench
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; stamp bench_driver
B="--steps 20 --warmup 5 --no-cpu-baseline --no-host-path --literal-ticks 0"
for rep in 1 2; do
  RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B > $OUT/bench_driver_form_quick_$rep.json 2> $OUT/bench_driver_form_quick_$rep.err
  RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B --plan host > $OUT/bench_driver_form_host_plan_$rep.json 2> $OUT/bench_driver_form_host_plan_$rep.err
done
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B --plan region > $OUT/bench_driver_form_plan_in_region.json 2> $OUT/bench_driver_form_plan_in_region.err
RGB_TRAFFIC_JSON=$OUT/traffic20.json timeout 120 python bench.py $B --train-form persistent > $OUT/bench_driver_form_persistent.json 2> $OUT/bench_driver_form_persistent.err
timeout 120 python bench.py $B --launch tick > $OUT/bench_driver_form_tick.json 2> $OUT/bench_driver_form_tick.err
stamp bench_forms
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
[ -f ra_amd/csrc/variants/timeline.so ] && RGB_LIB=$R/ra_amd/csrc/variants/timeline.so TL_HINT=2 TL_AGE=512 TL_TICKS=32 timeout 200 python tools/train_timeline.py > $OUT/train_timeline.txt 2>&1
[ -f ra_amd/csrc/variants/hist.so ] && RGB_LIB=$R/ra_amd/csrc/variants/hist.so timeout 200 python tools/train_decline_hist.py > $OUT/train_decline_hist.txt 2>&1
stamp tools
python - <<PY
import json, csv, glob
for name in ("bench", "bench_driver_form", "bench_driver_form_quick_1", "bench_driver_form_quick_2", "bench_driver_form_host_plan_1", "bench_driver_form_host_plan_2", "bench_driver_form_plan_in_region", "bench_driver_form_persistent", "bench_driver_form_tick"):
    try:
        d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(name, round(d["ms_per_step"]*1e3,2), "us/step", round(d["value"]/1e9,3), "G/s frac", round(r["frac"],4), r["kernel"], "avg_tick_us", round(r["avg_tick_us"],2), "tpl", r["ticks_per_launch"], "traffic/tick MB", round((r["traffic"] or 0)/r["ticks_per_launch"]/1e6, 2), "wall-events us", d.get("wall_minus_events_us"))
        if name in ("bench", "bench_driver_form"):
            for k in ("host_path", "literal_configs", "aux_kernels", "cpu_baseline"):
                print("   ", k, json.dumps(d.get(k))[:900])
            print("    train", json.dumps(d["config"]["train"])[:900])
    except Exception as e:
        print(name, "FAILED", e)
for f in sorted(glob.glob("$OUT/stats*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "tick" in r["Name"] or "train" in r["Name"] or "leaderboard" in r["Name"]: print(f.split("/")[-3], r["Name"][:70], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
head -60 $OUT/traffic.json
