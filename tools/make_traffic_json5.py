#!/usr/bin/env python3
"""profiles/r05_traffic.json from the PMC passes of ONE evidence run (tools/profile_round5.sh):
   tools/make_traffic_json5.py OUTDIR COMMIT > traffic.json
Per section: FETCH_SIZE / WRITE_SIZE (KB; separate rocprofv3 --pmc passes) per launch of the named kernel, mean over the
launches of that length; FETCH_SIZE x 2 is the guide's gfx950 correction (MI355X_MICROARCH.md, HBM section), WRITE_SIZE
uncorrected.  Infinity-Cache hits are counted by these memory-side counters: fabric traffic, an upper bound of HBM's."""
import csv, glob, json, os, sys
root, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def rows(dirname, counter, needle):
    vals = []
    for f in glob.glob(os.path.join(root, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                vals.append((float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    return vals


def section(tag, kernel, ticks, pick="all"):
    f, w = rows(f"pmc_fetch_{tag}", "FETCH_SIZE", kernel), rows(f"pmc_write_{tag}", "WRITE_SIZE", kernel)
    if not f or not w:
        return None
    if pick == "largest":                   # the timed launch among warm-up launches of other lengths: the largest grid
        g = max(x[1] for x in f); f = [x for x in f if x[1] == g]
        g = max(x[1] for x in w); w = [x for x in w if x[1] == g]
    fm, wm = sum(x[0] for x in f) / len(f), sum(x[0] for x in w) / len(w)
    return {"kernel": kernel, "ticks_per_launch": ticks, "launches": min(len(f), len(w)), "fetch_size_kb": fm, "write_size_kb": wm,
            "traffic_bytes_per_launch": (2 * fm + wm) * 1024, "traffic_bytes_per_tick": (2 * fm + wm) * 1024 / ticks}


out = {"commit": commit,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of one gpurun call "
                 "(tools/profile_round5.sh); the bench lines of the same call read this file (RGB_TRAFFIC_JSON)",
       "correction": "gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads -> x2; WRITE_SIZE uncorrected"}
s240 = section("240", "rgb_train_dealt_kernel<5>", 240)
s20 = section("20", "rgb_train_dealt_kernel<5>", 20)
s7 = section("lit", "rgb_train_dealt_kernel<7>", int(os.environ.get("LIT_TICKS", 32)))
out["closed_loop_240_tick_launches"] = s240
out["closed_loop_20_tick_launches_the_drivers_form"] = s20
out["literal_config5_train_7_members"] = s7
# what bench.py reads: the section whose launch length matches is chosen by RGB_TRAFFIC_TICKS (default 240)
want = os.environ.get("RGB_TRAFFIC_TICKS", "240")
pickd = s20 if want == "20" and s20 else s240
if pickd:
    out.update(ticks_per_launch=pickd["ticks_per_launch"], traffic_bytes_per_launch=pickd["traffic_bytes_per_launch"],
               traffic_bytes_per_tick=pickd["traffic_bytes_per_tick"])
print(json.dumps(out, indent=1))
