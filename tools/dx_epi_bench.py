#!/usr/bin/env python3
"""The two input-gradient products with fused reductions (linear2's dX: ReLU gate + column sums; out_proj's dX: row dots),
persistent w256 epilogues against the v256 / r256 ones.  usage: python tools/dx_epi_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = 65536
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dy = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16()
w2t = (torch.randn(2048, 512, device="cuda") * 0.05).bfloat16()   # (in = 2048, out = 512) transposed copy of linear2.weight
h = torch.relu(torch.randn(M, 2048, device="cuda")).bfloat16()
cs = torch.zeros(2048, device="cuda")
dx = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16)
wot = (torch.randn(512, 512, device="cuda") * 0.05).bfloat16()
o = torch.randn(M, 512, device="cuda").bfloat16()
dots = torch.empty(M, 4, device="cuda")
da = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
bits = torch.zeros(M, 256, device="cuda", dtype=torch.uint8)
ops.gemm((torch.randn(M, 512, device="cuda") * 0.5).bfloat16(), (torch.randn(2048, 512, device="cuda") * 0.05).bfloat16(), relu=True,
         relu_bits=bits, extra_flags=_lib.GEMM_TILE_V)
res512 = torch.randn(M, 512, device="cuda").bfloat16()
w2 = (torch.randn(512, 2048, device="cuda") * 0.05).bfloat16()
y512 = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
x512 = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16()
w1 = (torch.randn(2048, 512, device="cuda") * 0.05).bfloat16()
b1 = torch.randn(2048, device="cuda")
cases = {
    "linear1 fwd relu [65536x2048x512]": lambda: ops.gemm(x512, w1, out=dx, bias=b1, relu=True, extra_flags=_lib.GEMM_TILE_V),
    "linear1 fwd plain": lambda: ops.gemm(x512, w1, out=dx, extra_flags=_lib.GEMM_TILE_V),
    "linear1 fwd bias only": lambda: ops.gemm(x512, w1, out=dx, bias=b1, extra_flags=_lib.GEMM_TILE_V),
    "linear1 fwd relu only": lambda: ops.gemm(x512, w1, out=dx, relu=True, extra_flags=_lib.GEMM_TILE_V),
    "linear1 fwd relu + bit mask out": lambda: ops.gemm(x512, w1, out=dx, bias=b1, relu=True, relu_bits=bits, extra_flags=_lib.GEMM_TILE_V),
    "linear2 fwd + residual [65536x512x2048]": lambda: ops.gemm(h, w2, out=y512, residual=res512, extra_flags=_lib.GEMM_TILE_V),
    "linear2 fwd plain": lambda: ops.gemm(h, w2, out=y512, extra_flags=_lib.GEMM_TILE_V),
    "out_proj fwd + residual [65536x512x512]": lambda: ops.gemm(dy, wot, out=da, residual=res512, extra_flags=_lib.GEMM_TILE_V),
    "linear2 dX plain [65536x2048x512]": lambda: ops.gemm(dy, w2t, out=dx, extra_flags=_lib.GEMM_TILE_V),
    "linear2 dX BIT gate+colsum": lambda: ops.gemm(dy, w2t, out=dx, relu_bits=bits, colsum_into=cs, extra_flags=_lib.GEMM_TILE_V),
    "linear2 dX BIT gate only": lambda: ops.gemm(dy, w2t, out=dx, relu_bits=bits, extra_flags=_lib.GEMM_TILE_V),
    "linear2 dX gate+colsum [65536x2048x512]": lambda: ops.gemm(dy, w2t, out=dx, gate=h, colsum_into=cs, extra_flags=_lib.GEMM_TILE_V),
    "linear2 dX gate only": lambda: ops.gemm(dy, w2t, out=dx, gate=h, extra_flags=_lib.GEMM_TILE_V),
    "out_proj dX rowdot [65536x512x512]": lambda: ops.gemm(dy, wot, out=da, rowdot=(o, dots), extra_flags=_lib.GEMM_TILE_V),
    "out_proj dX plain": lambda: ops.gemm(dy, wot, out=da, extra_flags=_lib.GEMM_TILE_V),
}
for name, fn in cases.items():
    res = []
    for pers in (1, 0, 1, 0):
        _lib.lib().pero_set_option(b"gemm_persistent", pers)
        res.append(bench(fn))
    print(f"{name:42s} persistent w256 {min(res[0], res[2]):7.1f} us | v256 / r256 {min(res[1], res[3]):7.1f} us")
