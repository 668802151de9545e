#!/bin/bash
# First gpurun of the next round (one call, ~6 GPU-minutes):  gpurun --timeout 900 -- 'bash tools/next_round_probe.sh'
# 1. parity on the GPU for everything that was only verified on the CPU emulation at the end of round 1
#    (ABI v4 backoff mask, checksum over it, class-kernel PART parameter, commit_index_sent without per-peer
#    copies -- expect a slightly shorter tick than profiles/r01_bench.json from the last one), 2. the single-launch tick against the experimental two-launch tick
#    (RGB_DEBUG 8192 = same stream, 16384 = forked side stream / parallel graph branches; parity-preserving knobs),
# 3. a fresh rocprofv3 kernel trace of the default bench for profiles/.
set -u
mkdir -p gpurun_out/r02a
python -m pytest tests -x -q -m gpu > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest_gpu.log
for dbg in 0 8192 16384; do
  RGB_DEBUG=$dbg python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-host-path --check-ticks 2 \
      > gpurun_out/r02a/bench_dbg$dbg.json 2> gpurun_out/r02a/bench_dbg$dbg.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r02a/bench_dbg$dbg.json").read().strip().splitlines()[-1])
    print("dbg=$dbg", round(d["ms_per_step"] * 1e3, 2), "us/tick", round(d["value"] / 1e9, 2), "G decisions/s",
          "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print("dbg=$dbg failed:", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02a/prof" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 400 --warmup 32 --no-cpu-baseline --no-host-path --check-ticks 0 \
    > "$GRAFT_REPO_ROOT/gpurun_out/r02a/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
# 4. per-class wave timeline at steady state (400 generator ticks) and the SQ counter set for profiles/
TL_TICKS=400 timeout 300 python tools/wave_timeline.py > gpurun_out/r02a/wave_timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMDS="python $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 400 --no-cpu-baseline --no-host-path --check-ticks 0 --no-graph"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02a/sq1" -o p -- $CMDS > "$GRAFT_REPO_ROOT/gpurun_out/r02a/sq1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_fetch" -o p -- $CMDS > "$GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_write" -o p -- $CMDS > "$GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_write.log" 2>&1
