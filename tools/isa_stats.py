#!/usr/bin/env python3
"""Static instruction mix of the tick kernels (no GPU needed): compiles ra_amd/csrc/rgb_kernels.hip to
gfx950 assembly and counts, per kernel, VALU / SALU / branches / waitcnts / global loads and stores / LDS /
scratch (spill) instructions.  A wavefront runs one class path of rgb_tick_classes_kernel<N>, so the totals
bound the path lengths; the scratch count is the spill traffic to watch when the 128-VGPR budget binds.
usage: python tools/isa_stats.py [substring-of-kernel-name ...]   (default: classes kernels)"""
import os, re, subprocess, sys, tempfile
from collections import Counter
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1:] or ["rgb_tick_classes_kernel"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-o", out, os.path.join(root, "ra_amd", "csrc", "rgb_kernels.hip")], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
name, body, kernels = None, [], {}
for l in lines:
    m = re.match(r"^(_Z\w+):\s", l)
    if m:
        name, body = m.group(1), []
        continue
    if name and l.strip().startswith(".amdhsa_kernel"):
        kernels[name] = body
        name = None
        continue
    if name and l.startswith("\t"):
        t = l.strip()
        if t and not t.startswith((".", ";")):
            body.append(t.split()[0])
for k, ins in kernels.items():
    if not any(w in k for w in want):
        continue
    c = Counter()
    for x in ins:
        key = ("vmem_load" if x.startswith(("global_load", "buffer_load", "flat_load")) else
               "vmem_store" if x.startswith(("global_store", "buffer_store", "flat_store")) else
               "scratch" if x.startswith("scratch_") else "lds" if x.startswith("ds_") else
               "waitcnt" if x.startswith("s_waitcnt") else "branch" if x.startswith(("s_cbranch", "s_branch")) else
               "valu" if x.startswith("v_") else "salu" if x.startswith("s_") else "other")
        c[key] += 1
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)[:48]
    print(f"{short:50s} total {len(ins):6d}  " + "  ".join(f"{a} {c[a]}" for a in
          ("valu", "salu", "branch", "waitcnt", "vmem_load", "vmem_store", "lds", "scratch")))
