#!/usr/bin/env python3
"""Diagnostic builds of the eight-phase kernel: s_memtime at 17 points of one steady-state K-tile (waves 0 and 4).
Per phase: load part (reads + LDS-DMA issue), wait at the first barrier, MFMA cluster, wait at the second barrier."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
torch.manual_seed(0)
L = _lib.lib()


def show(tag, ph):   # ph: [G][2][32] int64
    ph = ph.double()
    d = ph[:, :, 1:17] - ph[:, :, 0:16]
    med = d.median(dim=0).values   # [2][16]
    for w in range(2):
        parts = []
        for p_ in range(4):
            ld, w1, mm, w2 = [float(med[w, 4 * p_ + k]) for k in range(4)]
            parts.append(f"P{p_ + 1}: load {ld:4.0f} bar {w1:4.0f} mfma {mm:4.0f} bar {w2:4.0f}")
        tot = float((ph[:, w, 16] - ph[:, w, 0]).median())
        print(f"{tag} wave {4 * w}: " + " | ".join(parts) + f" | K-tile {tot:.0f} cyc", flush=True)


M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for (N, K) in [(2048, 512), (2048, 2048)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    G = 256
    st = torch.zeros(G * 2 * 64 * 4 + G * 2 * 32, device="cuda", dtype=torch.int64)
    L.pero_set_option(b"gemm_policy", 20); L.pero_set_option(b"gemm_e_var", 72)
    for _ in range(20):
        ops.gemm(x, w, out=y, gate=st.view(torch.bfloat16).view(-1, 8))
    torch.cuda.synchronize()
    show(f"NT [{M}x{N}x{K}]", st[G * 2 * 64 * 4:].view(G, 2, 32).cpu())
L.pero_set_option(b"gemm_e_var", 64)
Mt = 262144
for (N, K) in [(2048, 512), (1536, 512)]:
    x = (torch.randn(Mt, K, device="cuda") * 0.5).bfloat16()
    dy = (torch.randn(Mt, N, device="cuda") * 0.5).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    st = torch.zeros(256 * 2 * 32, device="cuda", dtype=torch.int64)
    for _ in range(10):
        ops.gemm_raw(dy, x, dw, N, K, Mt, N, K, K, gate=st, ldg=8, flags=_lib.GEMM_TRANS_A | _lib.GEMM_TRANS_B | _lib.GEMM_ATOMIC, k_split=0,
                     in_dtype=_lib.PERO_BF16, out_dtype=_lib.PERO_F32)
    torch.cuda.synchronize()
    g = (N // 256) * (K // 256)
    nwg = int((st.view(256, 2, 32)[:, 0, 0] != 0).sum())
    show(f"TT dW [{N}x{K}] over {Mt} ({nwg} workgroups)", st.view(256, 2, 32)[:nwg].cpu())
L.pero_set_option(b"gemm_e_var", 0); L.pero_set_option(b"gemm_policy", 0)
