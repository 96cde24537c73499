#!/usr/bin/env python3
"""Same-process A/B of (side-stream dW on/off) x (GEMM tile policy) on the full training step.
usage: python tools/step_ab2.py [batch] [rounds] [policy ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib
from pero_pretraining_amd import functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pols = [int(a) for a in sys.argv[3:]] or [0, 8, 1, 2]
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(3)
FWD_NAMES = {0: "r256", 512: "v256", 513: "w256"}
res = {}
for r in range(rounds):
    for side in (True, False):
        for pol in pols:
            for fwdv in (0, 512, 513):
                F.SIDE_STREAM_DW = side
                F.FWD_TILE_FLAGS = 512 if fwdv else 0
                _lib.lib().pero_set_option(b"gemm_persistent", 1 if fwdv == 513 else 0)
                _lib.lib().pero_set_option(b"gemm_policy", pol)
                run(1)
                res.setdefault((side, pol, fwdv), []).append(run(4))
for (side, pol, fwdv), v in sorted(res.items()):
    v = sorted(v)
    print(f"side={side!s:5s} policy {pol} fwd={FWD_NAMES[fwdv]}: ms/step min {v[0]:.3f} median {v[len(v)//2]:.3f}  -> {B / v[len(v)//2] * 1e3:.0f} lines/s")
