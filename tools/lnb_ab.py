#!/usr/bin/env python3
"""The fused input-gradient + LayerNorm-backward launch at the step's two shapes (for same-box A/B of library builds through tools/lib_ab.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
M = 524288
torch.manual_seed(0)
gamma = torch.rand(512, device="cuda") + 0.5; beta = torch.randn(512, device="cuda") * 0.1
res = torch.randn(M, 512, device="cuda").bfloat16()
tt, mm, rr = ops.layernorm_fwd(res, gamma, beta, 1e-5)
dg, db, dxs = (torch.zeros(512, device="cuda") for _ in range(3))
out = []
for K in (1536, 2048):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(512, K, device="cuda") * 0.05).bfloat16()
    fn = lambda: ops.gemm_resid_layernorm_bwd(x, w, res, tt, rr, gamma, beta, dg, db, dxs)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    out.append(f"K={K}: {sorted(ts)[2]:.1f} us (min {min(ts):.1f})")
print(" | ".join(out))
