#!/usr/bin/env python3
"""profiles/r06_traffic.json from the PMC passes of ONE evidence run (tools/profile_round6.sh):
   tools/make_traffic_json6.py OUTDIR COMMIT > traffic.json
Per section and launch of the named kernel (mean over the launches of that length): FETCH_SIZE / WRITE_SIZE (KB) AND the
L2 <-> fabric request counters by size, each set in its own rocprofv3 --pmc pass.  The bytes a launch moves are taken
from the REQUESTS: TCC_EA0_RDREQ_128B x 128 + TCC_EA0_RDREQ_64B x 64 read, TCC_EA0_WRREQ_64B x 64 + (TCC_EA0_WRREQ -
TCC_EA0_WRREQ_64B) x 32 written -- the calibration of tools/probes/traffic_calib.hip (profiles/r06_calibration.json:
FETCH_SIZE tallies every read at 64 bytes, so it is x2 for 128-byte rows, x1 for 64-byte records; WRITE_SIZE is exact).
`traffic_fetch_x2` = 2 x FETCH_SIZE + WRITE_SIZE is the guide's formula, kept beside it.  Infinity-Cache hits are
counted by these memory-side counters: fabric traffic, an upper bound of HBM's."""
import csv, glob, json, os, sys
root, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def mean_of(dirname, counter, needle, pick):
    vals = []
    for f in glob.glob(os.path.join(root, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                vals.append((float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    if not vals:
        return None
    if pick == "largest":
        g = max(x[1] for x in vals); vals = [x for x in vals if x[1] == g]
    return sum(x[0] for x in vals) / len(vals)


def section(tag, kernel, ticks, pick="all"):
    g = lambda d, c: mean_of(f"pmc_{d}_{tag}", c, kernel, pick)
    f, w = g("fetch", "FETCH_SIZE"), g("write", "WRITE_SIZE")
    rd, r128, r64 = g("ea", "TCC_EA0_RDREQ_sum"), g("rdsz", "TCC_EA0_RDREQ_128B_sum"), g("rdsz", "TCC_EA0_RDREQ_64B_sum")
    wr, w64 = g("ea", "TCC_EA0_WRREQ_sum"), g("ea", "TCC_EA0_WRREQ_64B_sum")
    if f is None or w is None:
        return None
    s = {"kernel": kernel, "ticks_per_launch": ticks, "fetch_size_kb": f, "write_size_kb": w,
         "traffic_fetch_x2_bytes_per_tick": (2 * f + w) * 1024 / ticks}
    if None not in (rd, r128, r64, wr, w64):
        rb, wb = r128 * 128 + r64 * 64 + max(rd - r128 - r64, 0) * 64, w64 * 64 + (wr - w64) * 32
        s.update(fabric_reads=rd, fabric_reads_128B=r128, fabric_reads_64B=r64, fabric_writes=wr, fabric_writes_64B=w64,
                 read_bytes_per_tick=rb / ticks, write_bytes_per_tick=wb / ticks, traffic_bytes_per_launch=rb + wb,
                 traffic_bytes_per_tick=(rb + wb) / ticks, fabric_requests_per_tick=(rd + wr) / ticks)
    else:
        s.update(traffic_bytes_per_launch=(2 * f + w) * 1024, traffic_bytes_per_tick=(2 * f + w) * 1024 / ticks)
    return s


out = {"commit": commit,
       "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass>, separate passes of one gpurun call "
                 "(tools/profile_round6.sh); the bench lines of the same call read this file (RGB_TRAFFIC_JSON)",
       "calibration": "bytes from the fabric requests by size (tools/probes/traffic_calib.hip, profiles/r06_calibration.json)"}
s240 = section("240", "rgb_train_dealt_kernel<5>", 240)
s20 = section("20", "rgb_train_dealt_kernel<5>", 20)
s7 = section("lit", "rgb_train_dealt_kernel<7>", int(os.environ.get("LIT_TICKS", 32)))
out["closed_loop_240_tick_launches"] = s240
out["closed_loop_20_tick_launches_the_drivers_form"] = s20
out["literal_config5_train_7_members"] = s7
want = os.environ.get("RGB_TRAFFIC_TICKS", "240")
pickd = s20 if want == "20" and s20 else s240
if pickd:
    out.update(ticks_per_launch=pickd["ticks_per_launch"], traffic_bytes_per_launch=pickd["traffic_bytes_per_launch"],
               traffic_bytes_per_tick=pickd["traffic_bytes_per_tick"])
print(json.dumps(out, indent=1))
