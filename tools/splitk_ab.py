#!/usr/bin/env python3
"""Same-process A/B of the split-K work-item target of the weight-gradient products on the whole step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vals = [int(a) for a in sys.argv[2:]] or [256, 384, 512, 768, 1024]
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
for r in range(3):
    for v in vals:
        for nearest in (0, 1):
            _lib.lib().pero_set_option(b"splitk_items", v)
            _lib.lib().pero_set_option(b"splitk_nearest", nearest)
            run(1); res.setdefault((v, nearest), []).append(run(4))
for (v, nearest), x in sorted(res.items()):
    x = sorted(x); print(f"splitk_items {v:5d} rounding {'nearest' if nearest else 'up     '}: ms/step min {x[0]:.3f} median {x[1]:.3f}")
