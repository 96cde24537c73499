#!/usr/bin/env python3
"""Where does the eight-phase kernel differ from an f32 product?  Prints the pattern of bad outputs of one 256x256 tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
torch.manual_seed(0)
M, N, K = 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N, device="cuda") if "--bias" in sys.argv else None
_lib.lib().pero_set_option(b"gemm_policy", 20)
y = ops.gemm(x, w, bias=bias).float()
ref = x.float() @ w.float().t() + (bias if bias is not None else 0)
bad = ~((y - ref).abs() <= 0.05 * ref.abs().max())
print("bad outputs:", int(bad.sum()), "of", bad.numel(), " nan:", int(torch.isnan(y).sum()))
print("bad by row group of 16:", bad.view(16, 16, N).sum((1, 2)).tolist())
print("bad by column group of 8:", bad.view(M, 32, 8).sum((0, 2)).tolist())
print("bad by column mod 8:", bad.view(M, 32, 8).sum((0, 1)).tolist())
print("bad by row mod 16:", bad.view(16, 16, N).sum((0, 2)).tolist())
if bad.any():
    i, j = [int(v) for v in bad.nonzero()[0]]
    print("first bad", i, j, float(y[i, j]), float(ref[i, j]), "diff/bias", float(y[i, j] - ref[i, j]), None if bias is None else float(bias[j]))
