#!/usr/bin/env python3
"""Micro-benchmark of pero_gemm on the shapes of the masked-ViT step (config 2).  GPU box only.
usage: python tools/gemm_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 256
dev = "cuda"


def bench(name, fn, flops, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).bfloat16()


for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2"), (4096, 512, "head")]:
    x, w, dy = rnd(M, K), rnd(N, K), rnd(M, N)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device=dev, dtype=torch.float32)
    fl = 2.0 * M * N * K
    for tname, tf in (("t128", 64), ("t256", 128), ("s128", 256)):
        bench(f"{tag} NT y=x W^T  [{M}x{N}x{K}] {tname}", lambda: ops.gemm(x, w, out=y, extra_flags=tf), fl)
        bench(f"{tag} NN dx=dy W  [{M}x{K}x{N}] {tname}", lambda: ops.gemm(dy, w, out=dx, trans_b=True, extra_flags=tf), fl)
    bench(f"{tag} TT dW auto", lambda: ops.gemm(dy, x, out=dw, trans_a=True, trans_b=True, atomic=True, k_split=0), fl)
