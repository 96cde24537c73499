#!/usr/bin/env python3
"""Diagnostic build VAR 128 of the eight-phase kernel: cycles of every K-tile of a workgroup's third tile, of its epilogue and of the
next tile's first K-tiles (median over the workgroups; waves 0 and 4).
usage: python tools/gemm_e_ktiles.py [M=262144]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
torch.manual_seed(0)
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
RES = "--res" in sys.argv   # the residual epilogue
sys.argv = [a for a in sys.argv if a != "--res"]
VARX = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # extra variant bits (4: the epilogue without its stores)
CAP = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # at most this many workgroups (multiple of 8): fewer CUs share the memory system
for (N, K) in ([(512, 512), (512, 2048)] if RES else [(1536, 512), (2048, 512)]):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16()
    G = 256
    st = torch.zeros(G * 2 * 32, device="cuda", dtype=torch.int64)
    L.pero_set_option(b"gemm_policy", 20); L.pero_set_option(b"gemm_e_var", 128 | VARX | ((CAP // 8) << 8))
    for _ in range(5):
        if RES:
            ops.gemm_raw(x, w, y, M, N, K, K, K, N, bias=bias, residual=res, ldr=N, gate=st, ldg=8, flags=0,
                         in_dtype=_lib.PERO_BF16, out_dtype=_lib.PERO_BF16)
        else:
            ops.gemm(x, w, out=y, bias=bias, gate=st.view(torch.bfloat16).view(-1, 8))
    torch.cuda.synchronize()
    nk = K // 64
    ph = st.view(G, 2, 32)[:CAP or G].cpu().double()
    if nk + 11 > 32:
        continue
    d = (ph[:, :, 1:nk + 5] - ph[:, :, 0:nk + 4]).median(dim=0).values
    rel = (ph - ph[:, 0:1, nk - 1:nk]).median(dim=0).values   # relative to wave 0's last K-tile start
    for wv in range(2):
        names = [f"k{t}" for t in range(nk)] + ["epi", "gap", "k0'", "k1'"]
        print(f"[{M}x{N}x{K}] wave {4 * wv}: " + " ".join(f"{n}:{float(d[wv, i]):.0f}" for i, n in enumerate(names))
              + f" | tile {float((ph[:, wv, nk + 2] - ph[:, wv, 0]).median()):.0f}", flush=True)
        print("      timeline (0 = wave 0's last K-tile start): last K-tile start %.0f, P4 wait done %.0f, barrier %.0f, MFMA done %.0f, "
              "epilogue start %.0f, after the extra barrier %.0f, bias in registers %.0f, rows 0-63 stored %.0f, epilogue end %.0f, next tile K-tile 0 %.0f, K-tile 1 %.0f" % tuple(
                  float(rel[wv, i]) for i in (nk - 1, nk + 5, nk + 6, nk + 7, nk, nk + 8, nk + 9, nk + 10, nk + 1, nk + 2, nk + 3)), flush=True)
L.pero_set_option(b"gemm_e_var", 0); L.pero_set_option(b"gemm_policy", 0)
# split-K weight gradients (TT): the first 31 K-tiles of every work item
if "--tt" in sys.argv or True:
    L.pero_set_option(b"gemm_policy", 0); L.pero_set_option(b"gemm_e_var", 128)
    Mt = 262144
    for (N, K) in [(2048, 512), (1536, 512)]:
        x = (torch.randn(Mt, K, device="cuda") * 0.5).bfloat16()
        dy = (torch.randn(Mt, N, device="cuda") * 0.5).bfloat16()
        dw = torch.zeros(N, K, device="cuda")
        st = torch.zeros(256 * 2 * 32, device="cuda", dtype=torch.int64)
        for _ in range(5):
            ops.gemm_raw(dy, x, dw, N, K, Mt, N, K, K, gate=st, ldg=8, flags=_lib.GEMM_TRANS_A | _lib.GEMM_TRANS_B | _lib.GEMM_ATOMIC, k_split=0,
                         in_dtype=_lib.PERO_BF16, out_dtype=_lib.PERO_F32)
        torch.cuda.synchronize()
        ph = st.view(256, 2, 32).cpu().double()
        nwg = int((ph[:, 0, 1] != 0).sum())
        d = (ph[:nwg, :, 1:32] - ph[:nwg, :, 0:31]).median(dim=0).values
        for wv in range(2):
            print(f"TT dW [{N}x{K}] over {Mt} ({nwg} workgroups) wave {4 * wv}: K-tiles " + " ".join(f"{float(v):.0f}" for v in d[wv]), flush=True)
    L.pero_set_option(b"gemm_e_var", 0)
