#!/usr/bin/env python3
"""Reference point only (not used by the product): what does the vendor library (torch.matmul -> hipBLASLt/rocBLAS)
reach on the step's GEMM shapes on this device, next to pero_gemm?  Interleaved in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2"), (4096, 512, "head")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    dy = (torch.randn(M, N, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dwf = torch.zeros(N, K, device="cuda"); dwb = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    rows = []
    for name, ours, theirs in [
        ("NT y=xW^T", lambda: ops.gemm(x, w, out=y), lambda: torch.matmul(x, w.t(), out=y)),
        ("NN dx=dyW", lambda: ops.gemm(dy, w, out=dx, trans_b=True), lambda: torch.matmul(dy, w, out=dx)),
        ("TT dW=dy^Tx", lambda: ops.gemm(dy, x, out=dwf, trans_a=True, trans_b=True, atomic=True, k_split=0), lambda: torch.matmul(dy.t(), x, out=dwb)),
    ]:
        a = bench(ours); b = bench(theirs); a2 = bench(ours); b2 = bench(theirs)
        print(f"{tag:5s} {name:12s} [{M}x{N}x{K}]  pero {min(a,a2):7.1f} us {fl/min(a,a2)/1e6:7.1f} TF | vendor {min(b,b2):7.1f} us {fl/min(b,b2)/1e6:7.1f} TF")
