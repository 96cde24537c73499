#!/usr/bin/env python3
"""pero_gemm tile policies side by side on the step's stored-output products (NT / NN), interleaved in one process.
usage: python tools/gemm_policy_bench.py [M] [policy ...]     policies: 7 r256 (default), 8 q256, 2 t256, 3 s128, 10 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pols = [int(a) for a in sys.argv[2:]] or [7, 8, 2, 3]
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = {p: 0.0 for p in pols}
for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    dy = (torch.randn(M, N, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    wt = w.t().contiguous()  # [K, N]: dX = dY W as an NT product on a transposed weight copy
    for name, fn in [("NT", lambda: ops.gemm(x, w, out=y)), ("NN", lambda: ops.gemm(dy, w, out=dx, trans_b=True)),
                     ("dX on W^T (NT)", lambda: ops.gemm(dy, wt, out=dx))]:
        res = {}
        for rep in range(2):
            for p in pols:
                _lib.lib().pero_set_option(b"gemm_policy", p)
                res[p] = min(res.get(p, 1e9), bench(fn))
        for p in pols: tot[p] += res[p]
        print(f"{tag:5s} {name} [{M}x{N}x{K}] " + " | ".join(f"p{p}: {res[p]:6.1f} us {fl/res[p]/1e6:6.0f} TF" for p in pols))
print("sum: " + " | ".join(f"p{p}: {tot[p]:7.1f} us" for p in pols))
