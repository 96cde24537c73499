#!/usr/bin/env python3
"""Same-process A/B of library options on the full training step (devices differ by up to ~12 % between gpurun
calls, so variants must be interleaved in ONE process).  usage: python tools/step_ab.py [batch] [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

from pero_pretraining_amd import functional as F
run(3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4):
    sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
t_host = (time.perf_counter() - t0) / 4 * 1e3
torch.cuda.synchronize()
print(f"host enqueue time per step: {t_host:.3f} ms (GPU step below)")
for flag in (True, False, True, False):
    F.SIDE_STREAM_DW = flag
    run(1)
    print(f"side-stream dW {flag}: {run(4):.3f} ms/step")
F.SIDE_STREAM_DW = True
for nst, items in ((1, 512), (2, 512), (2, 256), (4, 256), (4, 128), (1, 512), (2, 256), (4, 256)):
    F.SIDE_STREAMS = nst
    F._side_streams.clear()
    _lib.lib().pero_set_option(b"splitk_items", items)
    run(1)
    print(f"side streams {nst} split-K items {items}: {run(4):.3f} ms/step")
F.SIDE_STREAMS = 2
F._side_streams.clear()
_lib.lib().pero_set_option(b"splitk_items", 512)
names = {0: "default r256|s128 +o128at", 9: "p128 sw-pipelined 3 slots", 3: "s128", 4: "o128 all"}
res = {k: [] for k in names}
for r in range(rounds):
    for pol in names:
        _lib.lib().pero_set_option(b"gemm_policy", pol)
        run(1)
        res[pol].append(run(4))
for pol, name in names.items():
    v = sorted(res[pol])
    print(f"policy {pol} {name:18s}: ms/step min {v[0]:.3f} median {v[len(v)//2]:.3f}  -> {B / v[len(v)//2] * 1e3:.0f} lines/s")
