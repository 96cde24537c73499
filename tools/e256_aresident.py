#!/usr/bin/env python3
"""Is the eight-phase GEMM's main loop waiting for HBM?  The same product with A streamed from HBM and with every tile reading its A rows from the first 4096 rows
(gemm_e_var bit 21: always in L2; results wrong by design), with the C stores (var 0), with L2-resident stores (1) and without stores (4); automatic walk off / on."""
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
for N, K in ((2048, 512), (2048, 2048), (512, 2048)):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for walk in (0, 1):
        _lib.call("pero_set_option", b"gemm_e_walk", walk)
        for var in (0, 1, 4):
            r = []
            for ares in (0, 1):
                _lib.call("pero_set_option", b"gemm_e_var", var | (ares << 21))
                r.append(bench(lambda: ops.gemm(x, w, out)))
            print(f"N={N} K={K} walk {walk} var {var}: A from HBM {r[0]:.0f} us ({fl / r[0] / 1e6:.0f} TF/s) | A from L2 {r[1]:.0f} us ({fl / r[1] / 1e6:.0f} TF/s)", flush=True)
_lib.call("pero_set_option", b"gemm_e_var", 0); _lib.call("pero_set_option", b"gemm_e_walk", 1)
