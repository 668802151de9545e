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
r = bench.run_literal(name, ticks, torch, engine, W, abi, torch.device("cuda", 0), 0)
print(json.dumps({k: r[k] for k in ("us_per_tick", "frac", "value", "launch", "final_state_equal", "oracle_checked_decisions",
                                    "per_tick_launches", "train_launch")}))
