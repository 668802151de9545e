#!/usr/bin/env python3
"""The analysis of tools/train_timeline.py as a function: the per-block stamps a -DRGB_X_TRAIN_TIMELINE build left in the
engine's debug buffer (rgb_debug_read) -> cadence per tick, per-class phase medians / means, late committers, residency.
Used by train_timeline.py (the closed loop) and tools/cfg5_probe.py (RGB_LITERAL_TIMELINE=1: a literal configuration)."""
import ctypes as C
import numpy as np
from ra_amd import engine


def report(eng, T, blocks_per_tick):
    class _P: pass
    plan = _P(); plan.blocks_per_tick = blocks_per_tick
    nblk = T * plan.blocks_per_tick
    buf = np.zeros(nblk * 8, dtype=np.uint64)
    L = engine.lib(); L.rgb_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    assert L.rgb_debug_read(eng._h, buf.ctypes.data, len(buf)) == 0
    b = buf.reshape(nblk, 8)
    tick_of = np.arange(nblk) // plan.blocks_per_tick
    ok = b[:, 0] > 0
    b, tick_of = b[ok], tick_of[ok]
    ts = b[:, :7].astype(np.int64); z = ts[:, 0].min()
    us = (ts - z) * 0.01                                    # wall_clock64 = 100 MHz
    cls = (b[:, 7] & np.uint64(0xFF)).astype(int); spins = ((b[:, 7] >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(int)
    names = {0: "aer", 1: "aer_reply", 2: "written", 3: "append", 4: "pipeline", 5: "req_vote", 6: "vote_res", 8: "el_timeout",
             10: "pre_vote_res", 11: "snap_written", 12: "hb_rpc", 13: "hb_reply", 14: "query"}
    print("span of the launch: %.1f us; wavefronts %d" % (us[:, 6].max(), len(b)))
    print("per tick: first start / median publish / last end (us), per-tick cadence of the median publish")
    prev = None
    for t in range(T):
        m = tick_of == t
        mp = np.median(us[m, 5])
        print(f"  tick {t:2d}: start {us[m,0].min():7.1f}  publish p50 {mp:7.1f} p99 {np.percentile(us[m,5],99):7.1f}  end {us[m,6].max():7.1f}"
              + (f"  cadence {mp-prev:5.1f}" if prev is not None else ""))
        prev = mp
    mid = (tick_of >= 4) & (tick_of < T - 2)
    print("steady ticks (4..T-3), per class: waves | msg load | dep wait (spins p50/p90) | row fetch | clause | publish | dec store | life  (medians, us)")
    for c in sorted(set(cls)):
        m = mid & (cls == c)
        if not m.any(): continue
        d = np.diff(us[m], axis=1)
        md = np.median(d, axis=0); p9 = np.percentile(d, 90, axis=0)
        print(f"  {names.get(c, c):>12}: {m.sum():6d} | {md[0]:5.2f} | {md[1]:5.2f} p90 {p9[1]:5.2f} ({np.median(spins[m]):.0f}/{np.percentile(spins[m],90):.0f}) | {md[2]:5.2f} | "
              f"{md[3]:5.2f} p90 {p9[3]:5.2f} | {md[4]:5.2f} | {md[5]:5.2f} | {np.median(us[m,6]-us[m,0]):5.2f}")
    # means and shares: the tick is (wavefronts x MEAN life) / resident wavefronts -- the tails count
    cntl = ((b[:, 7] >> np.uint64(8)) & np.uint64(0xFF)).astype(int)
    tot = (us[mid, 6] - us[mid, 0]).sum()
    print("the same by MEANS: waves | lanes per wave | msg load | dep wait | row fetch | clause | publish | dec store | life | share of all wave-time")
    for c in sorted(set(cls)):
        m = mid & (cls == c)
        if not m.any(): continue
        d = np.diff(us[m], axis=1).mean(axis=0)
        life = us[m, 6] - us[m, 0]
        print(f"  {names.get(c, c):>12}: {m.sum():6d} | {cntl[m].mean():5.1f} | {d[0]:5.2f} | {d[1]:5.2f} | {d[2]:5.2f} | {d[3]:5.2f} | {d[4]:5.2f} | {d[5]:5.2f} | "
              f"{life.mean():5.2f} | {100.0 * life.sum() / tot:5.1f} %")
    ticks_mid = len(set(tick_of[mid]))
    print(f"  all: {mid.sum() / ticks_mid:.0f} wavefronts per tick, mean life {tot / mid.sum():.2f} us, wave-time per tick {tot / ticks_mid:.0f} us "
          f"(/ 3072 slots = {tot / ticks_mid / 3072:.2f} us per tick if every slot were always busy)")
    # who commits late?  start -> publish above 12 / 14 / 16 us (what the next tick's wavefronts wait for), per class
    pub = us[:, 5] - us[:, 0]; nowait = pub - (us[:, 2] - us[:, 1])
    print("wavefronts per tick whose start->publish exceeds 12 / 14 / 16 us (of which: without their own dependency wait), per class")
    for c in sorted(set(cls)):
        m = mid & (cls == c)
        if not m.any(): continue
        print(f"  {names.get(c, c):>12}: " + "  ".join(f">{x}: {(pub[m] > x).sum() / ticks_mid:6.1f} ({(nowait[m] > x).sum() / ticks_mid:6.1f})" for x in (12, 14, 16)))
    # clause-time histogram of the three bulk classes (fast-path-only wavefronts against the ones that ran the general path)
    for c in (0, 1, 2):
        m = mid & (cls == c)
        if not m.any(): continue
        cl = us[m, 4] - us[m, 3]
        hist, edges = np.histogram(cl, bins=[0, 1, 2, 3, 4, 6, 8, 12, 100])
        print(f"  clause time of {names[c]} wavefronts (us bins 0-1-2-3-4-6-8-12+):", (hist / max(m.sum(), 1)).round(3).tolist())
    # resident wavefronts over time
    print("wavefronts in flight / waiting on dependencies, every 10 us:")
    for x in np.arange(0, us[:, 6].max(), 10.0):
        infl = ((us[:, 0] <= x) & (us[:, 6] > x)).sum(); wait = ((us[:, 1] <= x) & (us[:, 2] > x)).sum()
        print(f"  t={x:6.0f}  in flight {infl:5d}  waiting {wait:5d}")

