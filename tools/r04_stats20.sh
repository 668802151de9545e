#!/bin/bash
# kernel trace of train launches that all have the driver form's size (20 ticks: warm-up launch + timed launch), and
# the same command without the profiler: the dominant kernel's average duration by rocprofv3 next to bench.py's events
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r04e; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-path --check-ticks 0 --literal-ticks 0"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r04 -- python $R/bench.py --steps 20 --warmup 20 $Q > $OUT/stats.log 2>&1
cd $R
python bench.py --steps 20 --warmup 20 $Q > $OUT/bench_20_20.json 2> $OUT/bench_20_20.err
grep -h "train_dealt\|leaderboard\|prolog" $OUT/stats/*kernel_stats.csv | cut -c1-60,150-400
python -c "
import json; d=json.loads(open('$OUT/bench_20_20.json').read().strip().splitlines()[-1]); r=d['roofline']; print(r['kernel'], 'avg_launch_us', r['avg_launch_us'], 'frac', r['frac'])"
