#!/usr/bin/env python3
"""gemm_bf16_d128 against gemm_bf16_e256 by K (plain epilogue): the main loops side by side.  usage: python tools/d128_k.py [rows=262144]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[1]
for N, K in ((2048, 192), (2048, 512), (2048, 1024), (2048, 2048), (2048, 4096), (512, 2048), (512, 4096)):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    r = []
    for opt, var in ((0, 0), (64, 0)):
        _lib.call("pero_set_option", b"gemm_d128", opt)
        _lib.call("pero_set_option", b"gemm_e_var", var)
        r.append(bench(lambda: ops.gemm(x, w, out)))
    _lib.call("pero_set_option", b"gemm_d128", 0)
    _lib.call("pero_set_option", b"gemm_e_var", 0)
    print(f"N={N} K={K}: e256 {r[0]:.0f} us ({fl / r[0] / 1e6:.0f} TF/s) | d128 {r[1]:.0f} us ({fl / r[1] / 1e6:.0f} TF/s)", flush=True)
