#!/usr/bin/env python3
"""Is the eight-phase GEMM's main loop waiting for HBM?  The same product with A as M distinct rows (streamed from HBM) and with A as ONE 256-row panel
repeated (row stride trick: every tile reads the same 256 x K panel - always in L2), with and without the C stores (gemm_e_var 4)."""
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
    xp = x[:256].unsqueeze(0).expand(M // 256, 256, K).reshape(M, K) if False else None
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    # a [M, K] view whose rows repeat with period 256: as_strided over the first 256 rows is not expressible with one row stride, so use ONE row (stride 0)
    x0 = x[:1].expand(M, K)
    for var in (0, 4):
        _lib.call("pero_set_option", b"gemm_e_var", var)
        t = bench(lambda: ops.gemm_raw(x, w, out, M, N, K, K, K, N, flags=0, in_dtype=_lib.PERO_BF16, out_dtype=_lib.PERO_BF16))
        t0 = bench(lambda: ops.gemm_raw(x, w, out, M, N, K, 0, K, N, flags=0, in_dtype=_lib.PERO_BF16, out_dtype=_lib.PERO_BF16))
        print(f"N={N} K={K} var {var}: A from HBM {t:.0f} us ({fl / t / 1e6:.0f} TF/s) | A = one row repeated (L2) {t0:.0f} us ({fl / t0 / 1e6:.0f} TF/s)", flush=True)
_lib.call("pero_set_option", b"gemm_e_var", 0)
