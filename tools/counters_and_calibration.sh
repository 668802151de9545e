#!/bin/bash
# round 6, call 1: (A) what the sequence protocol's pieces cost TODAY (no poll + no publish store, and with them no wait
# for the store acknowledgements: the bound of a tag-inside-the-row protocol), (B) the L2 / L1 / EA counter passes of the
# 20-tick launch, FOUR counters of a block per pass and only the train kernel counted (round 5 asked for eight TCC
# counters per pass over every dispatch and never finished), (C) FETCH_SIZE / WRITE_SIZE calibrated on this kernel's own
# request shapes (tools/probes/traffic_calib.hip), (D) the class kernel built with FORCED spills: does it miscompare?
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/r06a; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5"
one() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  RGB_LIB=$V/$lib.so timeout 120 python bench.py $Q "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:26s} {r['avg_tick_us']:7.2f} us/tick by events, frac {r['frac']:.4f}, wall us/step {d['ms_per_step']*1e3:7.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
# ---- (D) first: it is the one that may need a second look
for v in pro5 spill5 spill6; do
  [ -f $V/$v.so ] || continue
  echo "== parity_tick0 $v" | tee -a $OUT/summary.txt
  RGB_LIB=$V/$v.so T=3 timeout 200 python tools/parity_tick0.py 2>&1 | tail -12 | tee -a $OUT/summary.txt
done
stamp spill
# ---- (A)
L="--steps 192 --warmup 16 --snapshot-kernel"
D="--steps 20 --warmup 5 --snapshot-kernel"
one pro_long pro5 $L
RGB_BENCH_NOCHECK=1 one nopub_long x_nopub $L --check-ticks 0
RGB_BENCH_NOCHECK=1 one nopub_nowait_long x_nopub_nowait $L --check-ticks 0
RGB_BENCH_NOCHECK=1 one nowait_long x_nowait $L --check-ticks 0
one pro_drv pro5 $D
RGB_BENCH_NOCHECK=1 one nopub_drv x_nopub $D --check-ticks 0
RGB_BENCH_NOCHECK=1 one nopub_nowait_drv x_nopub_nowait $D --check-ticks 0
one pro_long2 pro5 $L
RGB_BENCH_NOCHECK=1 one nopub_nowait_long2 x_nopub_nowait $L --check-ticks 0
stamp probes
# ---- (B)
cd /tmp && export TMPDIR=/tmp
export RGB_LIB=$V/pro5.so
C="python $R/bench.py --steps 20 --warmup 20 $Q --check-ticks 0 --no-graph"
P="timeout 150 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex rgb_train_dealt"
run() { local name=$1; shift; $P --pmc "$@" -d $OUT/pmc_$name -o r06 -- $C > $OUT/pmc_$name.log 2>&1; stamp pmc_$name; }
run tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcc3 TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
run tcc4 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
run tcc5 TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum
run tcc6 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum
run tcp3 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_ACCESSES_sum
run ta1 TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
# ---- (C)
PC="timeout 60 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex _kernel"
for k in rd128 rd128nt rd64 rd64sc1 rd32 rd1 wr16 wr32 wr64nt wr32nt wr128 wr1; do
  $PC --pmc FETCH_SIZE -d $OUT/cal_fetch_$k -o r06 -- $R/tools/probes/traffic_calib $k > $OUT/cal_fetch_$k.log 2>&1
  $PC --pmc WRITE_SIZE -d $OUT/cal_write_$k -o r06 -- $R/tools/probes/traffic_calib $k > $OUT/cal_write_$k.log 2>&1
  $PC --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/cal_ea_$k -o r06 -- $R/tools/probes/traffic_calib $k > $OUT/cal_ea_$k.log 2>&1
done
stamp calib
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
python - $OUT <<'PY' | tee -a $OUT/summary.txt
import csv, glob, os, sys, json, collections
out = sys.argv[1]
print("== calibration: counter value per dispatch of the probe kernel against the bytes it moved")
for k in "rd128 rd128nt rd64 rd64sc1 rd32 rd1 wr16 wr32 wr64nt wr32nt wr128 wr1".split():
    line = {}
    try:
        line["bytes"] = json.loads(open(f"{out}/cal_fetch_{k}.log").read().strip().splitlines()[-1])["bytes"]
    except Exception as e:
        line["bytes"] = None
    for what in ("fetch", "write", "ea"):
        for f in glob.glob(f"{out}/cal_{what}_{k}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "fill" in row["Kernel_Name"]: continue
                line[row["Counter_Name"]] = line.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print(k, json.dumps(line))
PY
grep "train_dealt_kernel<5>" $OUT/pmc_summary.txt | tee -a $OUT/summary.txt
stamp done
