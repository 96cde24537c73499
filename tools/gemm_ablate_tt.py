#!/usr/bin/env python3
"""Ablation of the 256x256x32 kernel on the weight-gradient layout (TT, both operands K-major), 256 workgroups via the
batch dimension (one k-slice per batch entry, separate outputs: no atomics).  Library built with EXTRA=-DPERO_GEMM_ABLATE."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
from pero_pretraining_amd._lib import GEMM_TRANS_A, GEMM_TRANS_B
pol = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
R = 65536
for (N, K) in [(2048, 512), (512, 512)]:
    dy = (torch.randn(R, N, device="cuda") * 0.5).bfloat16(); x = (torch.randn(R, K, device="cuda") * 0.5).bfloat16()
    tiles = (N // 256) * (K // 256)
    ks = 256 // tiles
    kchunk = R // ks
    out = torch.empty(ks, N, K, device="cuda", dtype=torch.float32)
    fl = 2.0 * R * N * K
    for name, fl_ in [("full", 0), ("no mfma (dma + lds reads)", MF), ("no dma (lds reads + mfma)", GL), ("dma only", MF | LD), ("lds reads only", MF | GL)]:
        us = bench(lambda: ops.gemm_raw(dy, x, out, N, K, kchunk, N, K, K, batch=ks, batch_inner=1, sA=(kchunk * N, 0), sB=(kchunk * K, 0),
                                        sC=(N * K, 0), flags=GEMM_TRANS_A | GEMM_TRANS_B | fl_, out_dtype=ops.PERO_F32))
        print(f"TT [{N}x{K}] over {R} in {ks} slices, policy {pol}: {name:28s} {us:8.1f} us  ({fl / us / 1e6:7.1f} TF-equivalent)")
