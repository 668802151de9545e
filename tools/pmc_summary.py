#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel, per counter, mean per dispatch."""
import csv, sys, glob, collections, os
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    print("==", os.path.relpath(f, root))
    for (kn, cn), (s, n) in sorted(acc.items()):
        if "tick" in kn or "train" in kn or "leaderboard" in kn or "wal" in kn:
            print(f"  {kn:60s} {cn:16s} mean={s/n:14.1f} n={n}")
