#!/usr/bin/env python3
"""Same-process A/B on the full step: input gradients as NN products on W (k-major) vs NT products on a transposed weight copy.
usage: python tools/dx_ab.py [batch] [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(3)
res = {}
for r in range(rounds):
    for side in (True, False):
        for wt, tile in ((False, 0), (True, 0), (True, 512)):
            F.SIDE_STREAM_DW, F.DX_ON_WT, F.DX_TILE_FLAGS = side, wt, tile
            run(1)
            res.setdefault((side, wt, tile), []).append(run(4))
for k, v in sorted(res.items()):
    v = sorted(v)
    print(f"side={k[0]!s:5s} dx_on_wt={k[1]!s:5s} tile_v={k[2] != 0!s:5s}: ms/step min {v[0]:.3f} median {v[len(v)//2]:.3f}")
