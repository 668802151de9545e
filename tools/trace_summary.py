#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a default bench.py run -> per-phase duration of rgb_tick_classes_kernel:
the untimed fast-forward (eager launches, interleaved with the generator), the warm-up and the timed hipGraph
replay.  usage: python tools/trace_summary.py <kernel_trace.csv> [age warmup]"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
age, warm = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 64)
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "tick_classes" in r["Kernel_Name"]], dtype=np.float64) / 1e3
print(f"rgb_tick_classes_kernel launches: {len(d)} (fast-forward {age}, warm-up {warm}, timed {len(d) - age - warm})")
for name, x in (("fast-forward (eager, generator in between)", d[:age]), ("warm-up", d[age:age + warm]), ("timed replay (hipGraph)", d[age + warm:])):
    if len(x):
        print(f"  {name:45s} mean {x.mean():6.2f} us  median {np.median(x):6.2f}  p5 {np.percentile(x, 5):6.2f}  p95 {np.percentile(x, 95):6.2f}  min {x.min():6.2f}  max {x.max():6.2f}")
