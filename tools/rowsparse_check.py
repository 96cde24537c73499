#!/usr/bin/env python3
"""functional.ROW_SPARSE_LAST_LAYER on / off: one training step's loss and every parameter gradient from the same state (bench model, 256 lines), then the step time at 2048 lines."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import functional as F
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, opt_, sched, trainer = bench.build(dev, True)
b = bench.synthetic(0, 256, dev)[0]
def grads(flag):
    F.ROW_SPARSE_LAST_LAYER = flag
    torch.manual_seed(5); torch.cuda.manual_seed_all(5)   # the forward draws its positional offsets
    model.zero_grad(set_to_none=True)
    out = model(b["images"], b["labels_dev"], b["mask_dev"])
    out["loss"].backward()
    torch.cuda.synchronize()
    return float(out["loss"]), {n: p.grad.detach().clone().float() for n, p in model.named_parameters() if p.grad is not None}
l0, g0 = grads(False)
l1, g1 = grads(True)
l2, g2 = grads(False)
print(f"loss {l0:.6f} / {l1:.6f}")
worst = []
for k in g0:
    ref = float(g0[k].abs().max()) + 1e-20
    e = float((g0[k] - g1[k]).abs().max()) / ref
    noise = float((g0[k] - g2[k]).abs().max()) / ref     # run-to-run (atomics)
    cos = float(torch.nn.functional.cosine_similarity(g0[k].reshape(1, -1), g1[k].reshape(1, -1)))
    worst.append((e, noise, cos, k))
worst.sort(reverse=True)
for e, nz, cos, k in worst[:12]:
    print(f"  {k:60s} max |diff| / max |g| = {e:.2e} (dense twice: {nz:.2e}) cosine {cos:.6f}")
print("all cosines >= %.6f" % min(w[2] for w in worst))
