#!/usr/bin/env python3
"""Correctness of a pero_gemm tile policy against the default one (all four operand layouts, bias / residual / relu / gate
epilogues, f32 and bf16 outputs).  usage: python tools/gemm_policy_check.py POLICY"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
pol = int(sys.argv[1])
torch.manual_seed(0)
M, N, K = 512, 768, 320 if pol not in (10, 15) else 384
bad = 0
for ta in (False, True):
    for tb in (False, True):
        a = (torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5).bfloat16()
        b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        res = (torch.randn(M, N, device="cuda")).bfloat16()
        gate = (torch.randn(M, N, device="cuda")).bfloat16()
        for kw in (dict(), dict(bias=bias, relu=True), dict(bias=bias, residual=res), dict(gate=gate), dict(out_dtype=torch.float32, bias=bias)):
            outs = []
            for p in (0, pol):
                _lib.lib().pero_set_option(b"gemm_policy", p)
                outs.append(ops.gemm(a, b, trans_a=ta, trans_b=tb, **kw).float())
            ref = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t())
            err = (outs[0] - outs[1]).abs().max().item()
            print(f"ta={ta} tb={tb} {sorted(kw)}: max |policy {pol} - default| = {err:.3e}  (|default - f32 matmul| = {(outs[0] - (ref if not kw else outs[0])).abs().max().item():.1e})")
            bad += err > 1e-2
# many tiles per workgroup (persistent kernels walk several tiles): 44 x 8 = 352 tiles of 256 x 256 on 256 CUs
M2, N2, K2 = 256 * 44, 2048, 192
a = (torch.randn(M2, K2, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N2, K2, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N2, device="cuda"); res = torch.randn(M2, N2, device="cuda").bfloat16()
outs = []
for p in (0, pol):
    _lib.lib().pero_set_option(b"gemm_policy", p)
    outs.append(ops.gemm(a, b, bias=bias, residual=res, relu=True).float())
err = (outs[0] - outs[1]).abs().max().item()
print(f"{M2}x{N2}x{K2} (352 tiles) bias+residual+relu: max |policy {pol} - default| = {err:.3e}")
bad += err > 1e-2
bt = (torch.randn(K2, N2, device="cuda") * 0.5).bfloat16()
outs = []
for p in (0, pol):
    _lib.lib().pero_set_option(b"gemm_policy", p)
    outs.append(ops.gemm(a, bt, trans_b=True).float())
err = (outs[0] - outs[1]).abs().max().item()
print(f"{M2}x{N2}x{K2} (352 tiles) NN: max |policy {pol} - default| = {err:.3e}")
bad += err > 1e-2
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
