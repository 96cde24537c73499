#!/usr/bin/env python3
"""Timing-only ablations of attn_bwd_lh_k (tools/abl/liblh_<mask>.so: 1 no LDS-DMA, 2 no MFMAs, 4 no fragment loads, 8 no output stores)."""
import ctypes, glob, os, re, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pero_pretraining_amd import ops
n, s, h, hd = 1024, 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
dqkv = torch.empty_like(qkv)
dbias = torch.zeros(3 * d, device="cuda")
work = torch.zeros(3 * n * h * 2 * 128, device="cuda")
V = ctypes.c_void_p
paths = sorted(glob.glob(os.path.join(R, "tools/abl/liblh_*.so")))
for path in paths + paths:
    lib = ctypes.CDLL(path)
    ctypes.c_int.in_dll(lib, "g_attn_lh").value = 1      # (the persistent kernel is an option: default off)
    f = lib.pero_attention_bwd
    f.argtypes = [V] * 8 + [ctypes.c_int64] * 4 + [ctypes.c_int, V]
    args = (qkv.data_ptr(), None, dout.data_ptr(), lse.data_ptr(), dvec.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), work.data_ptr(), n, s, h, hd, 1,
            torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        assert f(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f(*args)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(path):16s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
