#!/usr/bin/env python3
"""Diagnostic build of the eight-phase kernel (gemm_policy 28): s_memtime stamps per tile of waves 0 and 4.
Prints, per shape, the median cycles of main loop / epilogue / tile-to-tile and the in-kernel clock."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print(f"variant bits {VAR}")
torch.manual_seed(0)
for (N, K) in [(2048, 256), (2048, 512), (2048, 2048), (512, 512), (1536, 512)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    G = min(256, (M // 256) * (N // 256))
    st = torch.zeros((G, 2, 64, 4), device="cuda", dtype=torch.int64)
    _lib.lib().pero_set_option(b"gemm_policy", 20)
    _lib.lib().pero_set_option(b"gemm_e_var", VAR | 8)
    for _ in range(30):   # warm: the clock settles under load
        ops.gemm(x, w, out=y, bias=bias, gate=st.view(torch.bfloat16).view(-1, 8))
    torch.cuda.synchronize()
    s = st.cpu()
    nt = min(64, (M // 256) * (N // 256) // G)
    t0, t1, t2, rt = s[:, :, :nt, 0], s[:, :, :nt, 1], s[:, :, :nt, 2], s[:, :, :nt, 3]
    loop = (t1 - t0).float()
    epi = (t2 - t1).float()
    per = (t0[:, :, 1:] - t0[:, :, :-1]).float() if nt > 1 else loop
    clk = ((t0[:, :, -1] - t0[:, :, 0]).float() / ((rt[:, :, -1] - rt[:, :, 0]).float().clamp(min=1)) * 100.0) if nt > 1 else torch.zeros(1)
    nk = K // 64
    print(f"[{M}x{N}x{K}] tiles/WG {nt}: loop {loop.median():8.0f} cyc ({loop.median()/nk:6.0f}/K-tile; first tile {loop[:, :, 0].median():8.0f})  epilogue {epi.median():7.0f}"
          f"  tile-to-tile {per.median():8.0f}  clock {clk.median():6.0f} MHz   wave0 vs wave4 loop {loop[:, 0].median():.0f}/{loop[:, 1].median():.0f}", flush=True)
