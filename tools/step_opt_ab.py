#!/usr/bin/env python3
"""The masked step (2048 lines; PERO_AB_LINES) with a library option off / on, interleaved in one process (boxes differ by a few %).
usage: python tools/step_opt_ab.py <option> [v0=0] [v1=1] [steps=8] [more values ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd._lib import call
opt = sys.argv[1].encode()
v0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
v1 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
vals = [v0, v1] + [int(a) for a in sys.argv[5:]]   # further values to interleave
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, opt_, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, int(os.environ.get("PERO_AB_LINES", "2048")), dev)
def step(i):
    sched.update_learning_rate(i)
    b = batches[i % 2]
    return trainer.train_step_prepared(b["images"], b["labels_dev"], b["mask_dev"])
for i in range(4): step(i)
torch.cuda.synchronize()
for rep in range(3):
    for v in vals:
        call("pero_set_option", opt, v)
        step(0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps): step(i)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}={v}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)
