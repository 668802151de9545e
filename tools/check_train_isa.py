#!/usr/bin/env python3
"""Static check of the train kernel's publish step (no GPU needed): the sequence byte store must be preceded, on
every path, by an `s_waitcnt vmcnt(0)` that follows the last state store -- the compiler must neither drop nor move
the wait of rgb_tick_slice's step 4 -- and the decision stores behind it must NOT each wait for the previous one.
usage: python tools/check_train_isa.py [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [a for a in sys.argv[1:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-DRGB_X_ONLY_N=5", *flags, "-o", out, os.path.join(root, "ra_amd", "csrc", "rgb_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
body, inside = [], False
for l in lines:
    if re.match(r"^_ZN\S*rgb_train_kernelILi5E\S*:", l): inside = True; continue
    if inside:
        t = l.strip()
        if t.startswith(".amdhsa_kernel"): break
        if t and not t.startswith((";", ".")): body.append(t)
idx = [i for i, t in enumerate(body) if t.startswith("global_store_byte")]
assert len(idx) == 1, f"expected one sequence byte store, found {len(idx)}"
i = idx[0]
# walk back to the wait: only scalar / exec bookkeeping and the byte's own arithmetic may sit in between
back = body[max(0, i - 12):i]
w = [k for k, t in enumerate(back) if t.startswith("s_waitcnt") and "vmcnt(0)" in t]
assert w, "no s_waitcnt vmcnt(0) in front of the sequence byte store:\n" + "\n".join(back)
between = back[w[-1] + 1:]
assert not any(t.startswith(("global_", "buffer_", "flat_", "scratch_")) for t in between), "a memory operation between the wait and the byte store"
after = body[i + 1:]
nt = [k for k, t in enumerate(after) if t.startswith("global_store_dwordx4") and " nt" in t]
assert len(nt) >= 4, "decision stores not found"
waits = [t for t in after[:nt[3]] if t.startswith("s_waitcnt") and "vmcnt" in t]
print(f"publish: wait {len(between)} instructions before the byte store; vmcnt waits between the byte store and the 4th decision store: {len(waits)}")
assert len(waits) == 0, "the decision stores wait for each other (vmcnt) -- see rgb_tick_slice step 4"
print("ok")
