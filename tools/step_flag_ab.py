#!/usr/bin/env python3
"""The masked step with a Python-level switch of pero_pretraining_amd.functional at two values, interleaved in one process (boxes differ by a
few %: only a same-process A/B ranks two variants).
usage: python tools/step_flag_ab.py <FLAG> <v0> <v1> [lines=2048] [steps=8] [rounds=3]     (values are Python literals, e.g. True False 0 4096)"""
import ast, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import functional as F
flag = sys.argv[1]
v0, v1 = ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
lines = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 8
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
assert hasattr(F, flag), flag
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, opt_, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, lines, dev)
def step(i):
    sched.update_learning_rate(i)
    b = batches[i % 2]
    return trainer.train_step_prepared(b["images"], b["labels_dev"], b["mask_dev"])
for i in range(4): step(i)
torch.cuda.synchronize()
res = {repr(v0): [], repr(v1): []}
for rep in range(rounds):
    for v in (v0, v1):
        setattr(F, flag, v)
        step(0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps): loss = step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[repr(v)].append(ms)
        print(f"{flag}={v!r}: {ms:.2f} ms/step  (loss {float(loss.detach()):.5f}, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)
for k, v in res.items():
    print(f"{flag}={k}: median {sorted(v)[len(v) // 2]:.2f} ms/step over {len(v)} rounds")
