#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the PMC passes of one evidence run (tools/profile_round4.sh):
   tools/make_traffic_json.py OUTDIR [TICKS_PER_LAUNCH] > traffic.json
FETCH_SIZE / WRITE_SIZE (KB, separate passes) per launch of rgb_train_kernel<5> and of rgb_tick_classes_kernel<5> in the
same run; FETCH_SIZE x2 is the guide's gfx950 correction (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
tpl = int(sys.argv[2]) if len(sys.argv) > 2 else 16


def mean(dirname, counter, needle):
    vals = []
    for f in glob.glob(os.path.join(root, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(n in r["Kernel_Name"] for n in needle.split("|")):
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round4.sh), mean per launch; "
                 "the bench lines of the same gpurun call read this file (RGB_TRAFFIC_JSON)",
       "ticks_per_launch": tpl,
       "correction": "gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); "
                     "WRITE_SIZE uncorrected"}
f, nf = mean("pmc_fetch", "FETCH_SIZE", "rgb_train_kernel<5>|rgb_train_dealt_kernel<5>")
w, nw = mean("pmc_write", "WRITE_SIZE", "rgb_train_kernel<5>|rgb_train_dealt_kernel<5>")
if f is not None and w is not None:
    out.update(fetch_size_kb=f, write_size_kb=w, launches=min(nf, nw),
               traffic_bytes_per_launch=(2 * f + w) * 1024, traffic_bytes_per_tick=(2 * f + w) * 1024 / tpl)
f, nf = mean("pmc_fetch", "FETCH_SIZE", "rgb_tick_classes_kernel<5>")
w, nw = mean("pmc_write", "WRITE_SIZE", "rgb_tick_classes_kernel<5>")
if f is not None and w is not None:
    out["per_tick_kernel_same_run"] = {"kernel": "rgb_tick_classes_kernel<5> (the ageing and generation passes of the same run)",
                                       "launches": min(nf, nw), "fetch_size_kb": f, "write_size_kb": w,
                                       "traffic_bytes_per_launch": (2 * f + w) * 1024}
out["note"] = ("Infinity-Cache hits are counted by these memory-side counters (the 170 MB state is resident in the 256 MB "
               "Infinity Cache): fabric traffic, an upper bound of HBM traffic")
print(json.dumps(out, indent=1))
