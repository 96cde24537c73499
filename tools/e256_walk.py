#!/usr/bin/env python3
"""gemm_bf16_e256, stored products: N-tiles of one row panel per workgroup walked one after the other (gemm_e_var bits 16-19 = seq; 0: all ntn side by side).
usage: python tools/e256_walk.py [rows=524288]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
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
for N, K in ((2048, 512), (1536, 512), (2048, 2048)):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    ref = None
    for rep in range(2):
        for seq in ((0, 2, 4, 8) if N == 2048 else (0, 2, 3, 6)):
            _lib.call("pero_set_option", b"gemm_e_var", seq << 16)
            t = bench(lambda: ops.gemm(x, w, out, bias=bias))
            if ref is None: ref = out.clone()
            print(f"N={N} K={K} seq {seq}: {t:.0f} us ({fl / t / 1e6:.0f} TF/s) identical {torch.equal(out, ref)}", flush=True)
_lib.call("pero_set_option", b"gemm_e_var", 0)
