#!/usr/bin/env python3
"""Static instruction mix PER CLASS PATH of rgb_tick_classes_kernel<N> (no GPU needed): compiles the kernels with
-DRGB_X_MARK (comment markers around every class's process_message) and counts the instructions between the
markers in layout order -- VALU / SALU / VMEM / LDS / scratch (spill traffic) / branches.  Basic blocks the
compiler moved out of line are attributed to the class whose marker precedes them, so treat the numbers as a
guide.   usage: python tools/class_isa.py [-DFLAG ...]   (N = 5)"""
import os, re, subprocess, sys, tempfile
from collections import Counter, OrderedDict
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [a for a in sys.argv[1:] if a.startswith("-")]
N = os.environ.get("ONLY_N", "5")
KERNEL = os.environ.get("KERNEL", "rgb_tick_classes_kernel")   # or rgb_train_kernel
names = ["aer", "aer_reply", "written", "append", "pipeline_rpcs", "request_vote", "vote_result", "await_timeout",
         "election_timeout", "pre_vote_rpc", "pre_vote_result", "snapshot_written", "heartbeat_rpc", "heartbeat_reply",
         "consistent_query"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-DRGB_X_MARK", f"-DRGB_X_ONLY_N={N}", *flags, "-o", out,
                           os.path.join(root, "ra_amd", "csrc", "rgb_kernels.hip")], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
inside, cur = False, "prologue"
per = OrderedDict()
meta = {}
for l in lines:
    if re.match(rf"^_ZN\S*{KERNEL}ILi{N}E\S*:", l):
        inside = True; cur = "prologue"; continue
    if not inside:
        continue
    t = l.strip()
    if t.startswith(".amdhsa_kernel"):
        inside = False; continue
    m = re.search(r"; RGB_MARK (begin|end) (\d+)", t)
    if m:
        cur = names[int(m.group(2))] if m.group(1) == "begin" else "after_" + names[int(m.group(2))]
        continue
    for key in ("ScratchSize", "NumVgprs", "Occupancy", "codeLenInByte"):
        mm = re.match(rf"; {key}: (\d+)", t)
        if mm: meta[key] = int(mm.group(1))
    if not l.startswith("\t") or not t or t.startswith((".", ";")):
        continue
    x = t.split()[0]
    key = ("vmem_load" if x.startswith(("global_load", "buffer_load", "flat_load")) else
           "vmem_store" if x.startswith(("global_store", "buffer_store", "flat_store")) else
           "scratch" if x.startswith("scratch_") else "lds" if x.startswith("ds_") else
           "waitcnt" if x.startswith("s_waitcnt") else "branch" if x.startswith(("s_cbranch", "s_branch")) else
           "valu" if x.startswith("v_") else "salu" if x.startswith("s_") else "other")
    per.setdefault(cur, Counter())[key] += 1
cols = ["valu", "salu", "branch", "vmem_load", "vmem_store", "lds", "scratch", "waitcnt"]
print("flags:", " ".join(flags) or "(product)")
print(f"{'segment':28s}" + "".join(f"{c:>11s}" for c in cols) + f"{'total':>9s}")
tot = Counter()
for seg, c in per.items():
    tot.update(c)
    print(f"{seg:28s}" + "".join(f"{c[k]:11d}" for k in cols) + f"{sum(c.values()):9d}")
print(f"{'ALL':28s}" + "".join(f"{tot[k]:11d}" for k in cols) + f"{sum(tot.values()):9d}")
