#!/usr/bin/env python3
"""One literal configuration of bench.py (default: config 5 = BASELINE configs[4], 65 536 x 7 repair) on its own, oracle
checked as in the bench -- for A/B runs of N = 7 builds (RGB_LIB=ra_amd/csrc/variants/<name>.so, ONLY_N=7).
usage: python tools/cfg5_probe.py [config-name [ticks]]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ra_amd import abi, engine, workload as W
name = sys.argv[1] if len(sys.argv) > 1 else "5"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=0))      # (the graphs of run_literal are captured on the current stream)
cb = None
if os.environ.get("RGB_LITERAL_TIMELINE"):        # RGB_LIB = a -DRGB_X_TRAIN_TIMELINE build: where the train's wavefronts spend their lives
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import timeline_stats
    cb = timeline_stats.report
if os.environ.get("RGB_LITERAL_HIST"):            # RGB_LIB = a -DRGB_X_DECLINE_HIST build: why lanes decline the three bulk fast paths
    def cb(eng, ticks_, bpt):                     # (counters of every launch of this engine so far: per-tick passes and trains)
        import ctypes as C
        import numpy as np
        L = engine.lib(); L.rgb_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        buf = np.zeros(128, dtype=np.uint64)
        assert L.rgb_debug_read(eng._h, buf.ctypes.data, 128) == 0
        for c, nm in ((0, "append_entries_rpc"), (1, "append_entries_reply"), (2, "written")):
            row = buf[c * 32:(c + 1) * 32].astype(np.int64); tot = int(row.sum())
            if tot:
                print(f"class {c} {nm}: {tot} lanes reached the fast path, taken {row[0] / tot:.3f}; declines by reason:",
                      {k: round(int(row[k]) / tot, 4) for k in range(1, 32) if row[k]})
if os.environ.get("RGB_TRAIN_LEAD"):              # ordering probe: "class:lead,..." in ticks over the defaults (rgb_train_lead)
    import ctypes as C
    import numpy as np
    lead = np.array([0.0, 0.15, 0.10, 0.08, 0.08, 0, 0, 0, 0, 0, 0, 0.25, 0, 0, 0], dtype=np.float32)
    for kv in os.environ["RGB_TRAIN_LEAD"].split(","):
        c, v = kv.split(":"); lead[int(c)] = float(v)
    engine.lib().rgb_train_set_lead(lead.ctypes.data_as(C.c_void_p))
r = bench.run_literal(name, ticks, torch, engine, W, abi, torch.device("cuda", 0), 0, on_train=cb)
print(json.dumps({k: r[k] for k in ("us_per_tick", "frac", "value", "launch", "final_state_equal", "oracle_checked_decisions",
                                    "per_tick_launches", "train_launch")}))
