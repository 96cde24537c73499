#!/usr/bin/env python3
"""gemm_bf16_e256 plain epilogue by diagnostic variant (gemm_e_var): 0 default, 2 de-phased workgroups, 4 no stores, 16 staggered epilogues, 32 non-temporal stores.
usage: python tools/e256_var.py [rows=524288] [N=2048] [K=512]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
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
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * M * N * K
for rep in range(2):
    for var in (0, 1, 2, 4):
        _lib.call("pero_set_option", b"gemm_e_var", var)
        t = bench(lambda: ops.gemm(x, w, out))
        print(f"M={M} N={N} K={K} var {var}: {t:.0f} us ({fl / t / 1e6:.0f} TF/s)", flush=True)
_lib.call("pero_set_option", b"gemm_e_var", 0)
