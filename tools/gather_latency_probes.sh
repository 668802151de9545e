#!/bin/bash
# round 6, call 3: (A) does the line poll cut the L1 -> L2 read requests it was built to cut (base against coop, same
# counters)?  (B) the fabric's read latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) under a PURE gather of 128-byte rows and
# under gather + write-back of 0 / 16 / 32 / 64 / 128 bytes of every row, unprofiled timing beside it -- what do the
# train's partial write-backs do to the read side?
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=$R/gpurun_out/r06c; mkdir -p $OUT
V=$R/ra_amd/csrc/variants
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/summary.txt; }
for k in rw0 rw16 rw32 rw64 rw128 rd128 rd64 wr32 wr64nt wr128; do
  for i in 1 2 3; do $R/tools/probes/traffic_calib $k 4194304; done | tail -1 | tee -a $OUT/summary.txt
done
stamp unprofiled
cd /tmp && export TMPDIR=/tmp
PC="timeout 60 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex _kernel"
for k in rw0 rw16 rw32 rw64 rw128 rd64; do
  $PC --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum -d $OUT/lat_$k -o r06 -- $R/tools/probes/traffic_calib $k 4194304 > $OUT/lat_$k.log 2>&1
  $PC --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $OUT/tcp_$k -o r06 -- $R/tools/probes/traffic_calib $k 4194304 > $OUT/tcp_$k.log 2>&1
done
stamp probes
Q="--no-cpu-baseline --no-host-path --literal-ticks 0 --members 5"
C="python $R/bench.py --steps 20 --warmup 20 $Q --check-ticks 0 --no-graph"
P="timeout 150 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex rgb_train_dealt"
for v in base coop; do
  export RGB_LIB=$V/$v.so
  $P --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_${v}_tcp -o r06 -- $C > $OUT/pmc_${v}_tcp.log 2>&1
  $P --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum -d $OUT/pmc_${v}_tcc -o r06 -- $C > $OUT/pmc_${v}_tcc.log 2>&1
done
stamp train
cd $R
python tools/pmc_summary.py $OUT 2>&1 | grep -v "^==" | sed 's/void (anonymous namespace):://' | tee -a $OUT/summary.txt
python - $OUT <<'PY' | tee -a $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for k in "rw0 rw16 rw32 rw64 rw128 rd64".split():
    acc = {}
    for f in glob.glob(f"{out}/lat_{k}/**/*counter_collection.csv", recursive=True) + glob.glob(f"{out}/tcp_{k}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fill" in row["Kernel_Name"]: continue
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    try:
        print(k, "EA read latency", round(acc["TCC_EA0_RDREQ_LEVEL_sum"] / acc["TCC_EA0_RDREQ_sum"]), "cycles over", int(acc["TCC_EA0_RDREQ_sum"]), "reads;",
              "EA write latency", round(acc["TCC_EA0_WRREQ_LEVEL_sum"] / max(acc["TCC_EA0_WRREQ_sum"], 1)), "over", int(acc["TCC_EA0_WRREQ_sum"]), "writes;",
              "L1->L2 read latency", round(acc["TCP_TCC_READ_REQ_LATENCY_sum"] / acc["TCP_TCC_READ_REQ_sum"]), "over", int(acc["TCP_TCC_READ_REQ_sum"]),
              "; L1 pending stall", round(acc["TCP_PENDING_STALL_CYCLES_sum"] / acc["TCP_GATE_EN1_sum"], 3))
    except Exception as e:
        print(k, "incomplete", e, acc)
PY
stamp done
