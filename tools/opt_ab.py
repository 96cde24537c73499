#!/usr/bin/env python3
"""Same-process A/B of one integer library option (pero_set_option) on the full step.
usage: python tools/opt_ab.py NAME batch rounds value [value ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib

name = sys.argv[1].encode()
B, rounds = int(sys.argv[2]), int(sys.argv[3])
vals = [int(v) for v in sys.argv[4:]]
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(3)
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        _lib.lib().pero_set_option(name, v)
        run(1)
        res[v].append(run(4))
for v in vals:
    t = sorted(res[v])
    print(f"{sys.argv[1]}={v}: ms/step min {t[0]:.3f} median {t[len(t)//2]:.3f}")
