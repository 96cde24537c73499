#!/usr/bin/env python3
"""Split-K weight-gradient products (TT, f32 atomics) under different tile policies, interleaved in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pols = [int(a) for a in sys.argv[2:]] or [0, 13]
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = {p: 0.0 for p in pols}
for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2"), (4096, 512, "head"), (512, 1024, "patch")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); dy = (torch.randn(M, N, device="cuda") * 0.5).bfloat16()
    fl = 2.0 * M * N * K
    res, outs = {}, {}
    for rep in range(2):
        for p in pols:
            # p >= 100: automatic policy with the split-K workspace switched off (atomic epilogue)
            _lib.lib().pero_set_option(b"gemm_policy", p % 100)
            _lib.lib().pero_set_option(b"splitk_workspace", 0 if 100 <= p < 200 else 1)
            _lib.lib().pero_set_option(b"splitk_table", 0 if p >= 200 else 1)   # p >= 200: plain item order for unaligned slice counts
            dw = torch.zeros(N, K, device="cuda")
            ops.gemm(dy, x, out=dw, trans_a=True, trans_b=True, atomic=True, k_split=0)
            outs[p] = dw.clone()
            res[p] = min(res.get(p, 1e9), bench(lambda: ops.gemm(dy, x, out=dw, trans_a=True, trans_b=True, atomic=True, k_split=0)))
    for p in pols: tot[p] += res[p]
    ref = dy[:8192].float().t() @ x[:8192].float() if M <= 8192 else None
    err = max((outs[p] - outs[pols[0]]).abs().max().item() for p in pols)
    if ref is not None:
        err = max((outs[p] - ref).abs().max().item() for p in pols)
    print(f"{tag:5s} TT dW [{N}x{K}] over {M}: " + " | ".join(f"p{p}: {res[p]:6.1f} us {fl/res[p]/1e6:6.0f} TF" for p in pols) + f"  (max diff {err:.2e})")
print("sum: " + " | ".join(f"p{p}: {tot[p]:7.1f} us" for p in pols))
