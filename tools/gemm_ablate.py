#!/usr/bin/env python3
"""Ablation of the bf16 tile GEMM (library built with -DPERO_GEMM_ABLATE): where does a k-step's time go?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K) in [(2048, 512), (512, 2048), (512, 512)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for name, fl_ in [("full", 0), ("no-mfma (loads+lds reads)", 256), ("no-glds (lds reads+mfma)", 512), ("loads only", 1024), ("no loads, no compute (epilogue only)", 512 | 1024)]:
        us = bench(lambda: ops.gemm_raw(x, w, y, M, N, K, K, K, N, flags=fl_))
        print(f"[{M}x{N}x{K}] {name:40s} {us:8.1f} us  ({fl / us / 1e6:7.1f} TF-equivalent)")
