#!/usr/bin/env python3
"""Ablation of the 256x256 GEMM (library built with EXTRA=-DPERO_GEMM_ABLATE): where does a k-step's time go?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pol = int(sys.argv[2]) if len(sys.argv) > 2 else 8
_lib.lib().pero_set_option(b"gemm_policy", pol)
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
MF, GL, LD = 1 << 12, 1 << 13, 1 << 14
for (N, K) in [(512, 2048), (2048, 512)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for name, fl_ in [("full", 0), ("no mfma (dma + lds reads)", MF), ("no dma (lds reads + mfma)", GL), ("no lds reads (dma + mfma)", LD),
                      ("dma only", MF | LD), ("mfma only", GL | LD), ("lds reads only", MF | GL), ("nothing (barriers + epilogue)", MF | GL | LD)]:
        us = bench(lambda: ops.gemm_raw(x, w, y, M, N, K, K, K, N, flags=fl_))
        print(f"[{M}x{N}x{K}] {name:32s} {us:8.1f} us  ({fl / us / 1e6:7.1f} TF-equivalent)")
