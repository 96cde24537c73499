#!/usr/bin/env python3
"""One leg of bench.py (config 3 VQ argmin, config 4 VICReg step, config 5 NT-Xent step) by itself, for per-leg rocprof summaries.
usage: python tools/legs_only.py config3|config4|config5"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
timer = bench.Timer(dev)
which = sys.argv[1]
if which == "config3":
    out = bench.leg_config3(timer, dev)
elif which == "config4":
    out = bench.leg_joint(timer, dev, "vicreg", 128, 5)
else:
    out = bench.leg_joint(timer, dev, "ntxent", 128, 5)
print(json.dumps(out))
