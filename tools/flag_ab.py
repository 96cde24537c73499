#!/usr/bin/env python3
"""Same-process A/B of one boolean toggle of pero_pretraining_amd.functional on the full step.
usage: python tools/flag_ab.py FLAG [batch] [rounds]      e.g. RELU_GATE_BITS, DX_ON_WT, FUSE_ROWDOT"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import functional as F

flag = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(3)
res = {True: [], False: []}
for r in range(rounds):
    for v in (True, False):
        setattr(F, flag, v)
        run(1)
        res[v].append(run(5))
for v in (True, False):
    t = sorted(res[v])
    print(f"{flag}={v!s:5s}: ms/step min {t[0]:.3f} median {t[len(t)//2]:.3f}")
