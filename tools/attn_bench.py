#!/usr/bin/env python3
"""Fused attention kernels: TFLOP/s vs sequence length (prologue / epilogue share)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
h, hd = 4, 128
for n, s in [(256, 256), (128, 512), (64, 1024), (32, 2048)]:
    d = h * hd
    qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
    dout = torch.randn(n * s, d, device="cuda").bfloat16()
    out, lse = ops.attention_fwd_fused(qkv, n, s, h)
    tf = bench(lambda: ops.attention_fwd_fused(qkv, n, s, h))
    tb = bench(lambda: ops.attention_bwd_fused(qkv, out, dout, lse, n, s, h))
    db = torch.zeros(3 * h * hd, device="cuda")
    tbb = bench(lambda: ops.attention_bwd_fused(qkv, out, dout, lse, n, s, h, dbias=db))
    fl = 4.0 * s * s * hd * n * h
    print(f"N={n:4d} S={s:5d}: fwd {tf:7.1f} us {fl/tf/1e6:6.1f} TF | bwd {tb:7.1f} us {2.5*fl/tb/1e6:6.1f} TF useful ({3.0*fl/tb/1e6:6.1f} executed) | with in_proj bias gradient {tbb:7.1f} us")
