#!/usr/bin/env python3
"""Static check of the train kernels' publish steps (no GPU needed), for the dealt and the persistent form (N = 5):
every group of sequence byte stores -- the ONE store of a message slice (rgb_tick_slice step 4) and the N stores of a
leaderboard snapshot row (rgb_train_snap_slice) -- must be preceded, on every path, by an `s_waitcnt vmcnt(0)` with no
memory operation in between (the compiler must neither drop nor move the wait), and the decision stores behind the
slice's store must NOT each wait for the previous one.
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
MEM = ("global_", "buffer_", "flat_", "scratch_")
for kern in ("rgb_train_dealt_kernelILi5E", "rgb_train_kernelILi5E"):
    body, inside = [], False
    for l in lines:
        if re.match(rf"^_ZN\S*{kern}\S*:", l): inside = True; continue
        if inside:
            t = l.strip()
            if t.startswith(".amdhsa_kernel"): break
            if t and not t.startswith((";", ".")): body.append(t)
    assert body, f"{kern}: not found"
    idx = [i for i, t in enumerate(body) if t.startswith("global_store_byte")]
    groups = []                                    # runs of byte stores with only address arithmetic between them
    for i in idx:
        if groups and i - groups[-1][-1] <= 4 and not any(t.startswith(MEM) for t in body[groups[-1][-1] + 1:i]):
            groups[-1].append(i)
        else:
            groups.append([i])
    assert sorted(len(g) for g in groups) == [1, 5], f"{kern}: expected the slice's byte store and a snapshot row's five, found {[len(g) for g in groups]}"
    for g in groups:
        i = g[0]
        back = body[max(0, i - 32):i]
        w = [k for k, t in enumerate(back) if t.startswith("s_waitcnt") and "vmcnt(0)" in t]
        assert w, f"{kern}: no s_waitcnt vmcnt(0) in front of the sequence byte store(s) at {i}:\n" + "\n".join(back)
        between = back[w[-1] + 1:]
        assert not any(t.startswith(MEM) for t in between), f"{kern}: a memory operation between the wait and the byte store at {i}"
        if len(g) == 1:
            after = body[i + 1:]
            nt = [k for k, t in enumerate(after) if t.startswith("global_store_dwordx4") and " nt" in t]
            assert len(nt) >= 4, f"{kern}: decision stores not found"
            waits = [t for t in after[:nt[3]] if t.startswith("s_waitcnt") and "vmcnt" in t]
            print(f"{kern}: slice publish: wait {len(between)} instructions before the byte store; vmcnt waits between the byte store and the 4th decision store: {len(waits)}")
            assert len(waits) == 0, f"{kern}: the decision stores wait for each other (vmcnt) -- see rgb_tick_slice step 4"
        else:
            print(f"{kern}: snapshot row: wait {len(between)} instructions before its {len(g)} byte stores")
print("ok")
