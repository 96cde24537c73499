#!/usr/bin/env python3
"""LayerNorm forward / backward kernel timing at the step's shape (65536 x 512 bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
rows, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 512
x = torch.randn(rows, d, device="cuda").bfloat16(); dy = torch.randn(rows, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
dg, db, dxs = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
tf = bench(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
tb = bench(lambda: ops.layernorm_bwd(dy, y, mean, rstd, g, dg, db, dxs))
print(f"layernorm fwd {tf:.1f} us ({rows * d * 4 / tf / 1e6:.2f} TB/s)   bwd (+reduce) {tb:.1f} us ({rows * d * 6 / tb / 1e6:.2f} TB/s)")
