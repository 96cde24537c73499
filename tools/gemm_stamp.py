#!/usr/bin/env python3
"""Phase shares of gemm_bf16_s128's k-step (library built with EXTRA=-DPERO_GEMM_STAMP)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
h = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
M = 32768
for (N, K, tb) in [(2048, 512, False), (512, 2048, False), (512, 2048, True)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16() if not tb else (torch.randn(K, N, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(x, w, out=y, trans_b=tb)
    torch.cuda.synchronize(); h.pero_debug_read_stamps(buf, 1)
    for _ in range(5): ops.gemm(x, w, out=y, trans_b=tb)
    torch.cuda.synchronize(); h.pero_debug_read_stamps(buf, 1)
    steps = buf[5] or 1
    names = ["vmcnt wait", "barrier", "glds issue", "ds_read+land", "16 MFMA"]
    tot = sum(buf[i] for i in range(5))
    print(f"[{M}x{N}x{K} tb={tb}] cycles per wave-k-step: " + ", ".join(f"{n} {buf[i]/steps:.0f}" for i, n in enumerate(names)) + f"  | total {tot/steps:.0f} (MFMA-only floor 256)")
