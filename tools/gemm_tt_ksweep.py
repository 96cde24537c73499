#!/usr/bin/env python3
"""Weight-gradient product (TT, f32 atomics, 128x128x64 kernel) against the number of k-slices."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
M = 65536
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); dy = (torch.randn(M, N, device="cuda") * 0.5).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    tiles = (N // 128) * (K // 128)
    out = []
    for ks in (1, 2, 4, 8, 16, 32, 64):
        t = bench(lambda: ops.gemm(dy, x, out=dw, trans_a=True, trans_b=True, atomic=True, k_split=ks))
        out.append(f"ks {ks:2d} ({tiles * ks:4d} wg): {t:6.1f} us {fl / t / 1e6:4.0f} TF")
    print(f"{tag:5s} [{N}x{K}] " + " | ".join(out))
