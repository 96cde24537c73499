#!/usr/bin/env python3
"""Does a power-of-two leading dimension hurt (L2 channel conflicts)?  Same GEMM with padded row pitches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
M = 32768
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K) in [(2048, 512), (512, 2048), (1536, 512), (512, 512)]:
    for pa, pb, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (0, 0, 64), (64, 64, 64), (8, 8, 8), (32, 32, 32)]:
        xa = (torch.randn(M, K + pa, device="cuda") * 0.5).bfloat16(); wa = (torch.randn(N, K + pb, device="cuda") * 0.5).bfloat16()
        ya = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16)
        x, w, y = xa[:, :K], wa[:, :K], ya[:, :N]
        us = bench(lambda: ops.gemm(x, w, out=y))
        print(f"NT [{M}x{N}x{K}] pad A {pa:2d} B {pb:2d} C {pc:2d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF")
