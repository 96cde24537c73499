#!/usr/bin/env python3
"""Same-process A/B of the data-parallel driver's cost with ONE rank (nothing to reduce: pure overhead of hooks, stream
joins and the collective calls).  usage: python tools/dp_ab.py [batch] [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
import bench
from pero_pretraining_amd.parallel import DataParallel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
variants = {"no DP": None, "DP overlap, 2 layers/bucket": dict(overlap=True, layers_per_bucket=2),
            "DP overlap, 12 layers/bucket": dict(overlap=True, layers_per_bucket=12), "DP no overlap": dict(overlap=False)}
res = {k: [] for k in variants}
run(3)
for r in range(rounds):
    for name, kw in variants.items():
        model.backbone._on_layer_grads_ready = None
        trainer.data_parallel = None if kw is None else DataParallel(model, opt, **kw)
        if kw is None: opt.grad_scale = 1.0
        run(1)
        res[name].append(run(4))
for name, v in res.items():
    v = sorted(v)
    print(f"{name:32s}: ms/step min {v[0]:.3f} median {v[len(v)//2]:.3f}")
dist.destroy_process_group()
