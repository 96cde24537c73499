#!/usr/bin/env python3
"""Masked cross-entropy kernels at the step's shape (65536 x 4096 bf16 logits, 15 % masked rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
rows, V = 65536, 4096
lg = torch.randn(rows, V, device="cuda").bfloat16()
lab = torch.randint(0, V, (rows,), device="cuda")
msk = (torch.rand(rows, device="cuda") < 0.15).long()
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
loss, work = ops.masked_ce_fwd(lg, lab, msk, None)
tf = bench(lambda: ops.masked_ce_fwd(lg, lab, msk, None))
tb = bench(lambda: ops.masked_ce_bwd(lg, lab, msk, work, None))
act = int(msk.sum())
print(f"masked CE fwd (rows + final) {tf:.1f} us ({act * V * 2 / tf / 1e6:.2f} TB/s of active-row reads) | bwd {tb:.1f} us "
      f"({(rows * V * 2 + act * V * 2) / tb / 1e6:.2f} TB/s)")
