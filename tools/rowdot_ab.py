#!/usr/bin/env python3
"""Same-process A/B: D = rowsum(dO * O) from the dX product's epilogue vs computed inside the dQ kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import functional as F
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, 256, dev)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3)
res = {True: [], False: []}
for r in range(4):
    for flag in (True, False):
        F.FUSE_ROWDOT = flag
        run(1); res[flag].append(run(4))
for flag in (True, False):
    v = sorted(res[flag]); print(f"FUSE_ROWDOT={flag!s:5s}: ms/step min {v[0]:.3f} median {v[len(v)//2]:.3f}")
