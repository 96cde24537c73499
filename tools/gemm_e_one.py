#!/usr/bin/env python3
"""One product on the eight-phase kernel, repeated (for rocprofv3 --pmc passes): python tools/gemm_e_one.py M N K [reps] [tt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
tt = len(sys.argv) > 5 and sys.argv[5] == "tt"
torch.manual_seed(0)
_lib.lib().pero_set_option(b"gemm_policy", 20)
if tt:
    dy = (torch.randn(M, N, device="cuda") * 0.5).bfloat16(); x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    for _ in range(reps):
        ops.gemm(dy, x, out=dw, trans_a=True, trans_b=True, atomic=True, k_split=0)
else:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(reps):
        ops.gemm(x, w, out=y, bias=bias)
torch.cuda.synchronize()
