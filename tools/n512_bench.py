#!/usr/bin/env python3
"""The N = 512 products of a 2048-line step (524 288 rows) on the two tile kernels and with the LayerNorm epilogue: what an epilogue costs.
usage: python tools/n512_bench.py [rows=524288]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
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
gamma = torch.rand(512, device="cuda") + 0.5; beta = torch.randn(512, device="cuda") * 0.1
bias = torch.randn(512, device="cuda")
res = torch.randn(M, 512, device="cuda").bfloat16()
dy = torch.randn(M, 512, device="cuda").bfloat16()
dg, db, dxs = (torch.zeros(512, device="cuda") for _ in range(3))
for K in (512, 1536, 2048):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    x = torch.relu(x) if K == 2048 else x     # linear2's input is half zeros
    w = (torch.randn(512, K, device="cuda") * 0.05).bfloat16()
    fl = 2.0 * M * 512 * K
    out = {}
    out["e256 plain"] = bench(lambda: ops.gemm(x, w, bias=bias))
    out["e256 resid"] = bench(lambda: ops.gemm(x, w, bias=bias, residual=res))
    L.pero_set_option(b"gemm_nw", 1)
    out["n512 plain"] = bench(lambda: ops.gemm(x, w, bias=bias))
    out["n512 resid"] = bench(lambda: ops.gemm(x, w, bias=bias, residual=res))
    L.pero_set_option(b"gemm_nw", 0)
    out["n512 resid+LN (y, t)"] = bench(lambda: ops.gemm_resid_layernorm(x, w, bias, res, gamma, beta, 1e-5))
    out["n512 resid+LN (t)"] = bench(lambda: ops.gemm_resid_layernorm(x, w, bias, res, gamma, beta, 1e-5, store_y=False))
    tt, mm, rr = ops.layernorm_fwd(res, gamma, beta, 1e-5)
    out["n512 resid+LN bwd"] = bench(lambda: ops.gemm_resid_layernorm_bwd(x, w, res, tt, rr, gamma, beta, dg, db, dxs))
    print(f"K = {K}: " + " | ".join(f"{k} {v:.0f} us ({fl / v / 1e6:.0f} TF/s)" for k, v in out.items()), flush=True)
y, t, mean, rstd = ops.gemm_resid_layernorm(x, w, bias, res, gamma, beta, 1e-5)
print(f"layernorm fwd {bench(lambda: ops.layernorm_fwd(y, gamma, beta, 1e-5)):.0f} us | bwd {bench(lambda: ops.layernorm_bwd(dy, y, mean, rstd, gamma, dg, db, dxs)):.0f} us | "
      f"bwd from t {bench(lambda: ops.layernorm_bwd_out(dy, t, rstd, gamma, beta, dg, db, dxs)):.0f} us")
