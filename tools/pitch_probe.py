#!/usr/bin/env python3
"""Does the row pitch of a GEMM's output / input matter (power-of-two pitches against DRAM channel interleaving)?  The step's K = 512 products with C (and A) as
column-slice views of wider buffers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = 524288
torch.manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[2]
for N, K in ((2048, 512), (1536, 512), (512, 2048)):
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    line = []
    for rep in range(2):
        for pa, pc in ((0, 0), (0, 64), (0, 128), (0, 192), (64, 0), (64, 64)):
            xa = (torch.randn(M, K + pa, device="cuda") * 0.5).bfloat16()
            ca = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16)
            x, out = xa[:, :K], ca[:, :N]
            t = bench(lambda: ops.gemm(x, w, out))
            line.append(f"A+{pa} C+{pc}: {t:.0f}")
            del xa, ca
    print(f"[{N} x {K}] (pad in columns): " + " | ".join(line), flush=True)
