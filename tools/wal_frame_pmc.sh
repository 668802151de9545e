#!/bin/bash
# PMC passes over tools/wal_frame_bench.py (the framing kernel's three workloads): gpurun_out/$1/
#   tools/wal_frame_pmc.sh TAG [RGB_LIB]
set -u
TAG=${1:-walpmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
[ -n "${2:-}" ] && export RGB_LIB=$2
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/tools/wal_frame_bench.py"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o w -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o w -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o w -- $CMD > $OUT/pmc_sq2.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "frame_kernel" not in r["Kernel_Name"]: continue
            grp = "G8" if "kernel<8>" in r["Kernel_Name"] else "G16" if "kernel<16>" in r["Kernel_Name"] else "G64"
            acc[(grp, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in sorted(acc.items()):
            print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
