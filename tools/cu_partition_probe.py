#!/usr/bin/env python3
"""Does a CU-partitioned pair of streams (hipExtStreamCreateWithCUMask) run the tile GEMM on most of the chip and the latency-bound
kernels (attention, LayerNorm) on the rest at the same time, and at what rates?  usage: python tools/cu_partition_probe.py [n_gemm_cus=192]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits: words[b >> 5] |= (1 << (b & 31))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

NG = int(sys.argv[1]) if len(sys.argv) > 1 else 192
torch.manual_seed(0)
M = 196608
x = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16(); w = (torch.randn(2048, 512, device="cuda") * 0.5).bfloat16(); bias = torch.randn(2048, device="cuda")
out = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16)
n, s, h, hd = 768, 256, 4, 128
d = h * hd
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16(); dout = torch.randn(n * s, d, device="cuda").bfloat16()
o, lse = ops.attention_fwd_fused(qkv, n, s, h)
db = torch.zeros(3 * d, device="cuda")
g1 = torch.ones(d, device="cuda"); b1 = torch.zeros(d, device="cuda"); xin = torch.randn(n * s, d, device="cuda").bfloat16()
_lib.lib().pero_set_option(b"gemm_policy", 20)
gemm = lambda: ops.gemm(x, w, bias=bias, relu=True, out=out)
attb = lambda: ops.attention_bwd_fused(qkv, o, dout, lse, n, s, h, dbias=db)
attf = lambda: ops.attention_fwd_fused(qkv, n, s, h)
lnf = lambda: ops.layernorm_fwd(xin, g1, b1, 1e-5)
torch.cuda.synchronize()

def run(stream, fn, iters):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters): fn()
        e1.record(stream)
    return e0, e1

def alone(tag, stream, fn, iters=200):
    e0, e1 = run(stream, fn, iters); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{tag:56s} {us:8.1f} us", flush=True)
    return us

full = torch.cuda.Stream()
t_g = alone("gemm 196608x2048x512 relu, whole chip", full, gemm)
t_ab = alone("attention backward 768 lines, whole chip", full, attb)
t_af = alone("attention forward 768 lines, whole chip", full, attf)
t_ln = alone("layernorm forward 196608 x 512, whole chip", full, lnf)
for layout in ("striped", "blocked"):
    if layout == "striped":   # bit b -> XCC b % 8: the first NG bits are NG / 8 CUs of every XCC
        gb = [b for b in range(256) if b < NG]
    else:                     # bit b -> XCC b // 32
        gb = [b for b in range(256) if (b % 32) < NG // 8]
    sb = [b for b in range(256) if b not in gb]
    G, S = masked_stream(gb), masked_stream(sb)
    _lib.lib().pero_set_option(b"gemm_e_var", (NG // 8) << 8)
    print(f"--- mask layout {layout}: {len(gb)} CUs for the GEMM stream, {len(sb)} for the other", flush=True)
    tg = alone("gemm on its partition, alone", G, gemm)
    tab = alone("attention backward on the small partition, alone", S, attb, 60)
    taf = alone("attention forward on the small partition, alone", S, attf, 100)
    tln = alone("layernorm forward on the small partition, alone", S, lnf, 200)
    for tag, fn, ts in (("attention backward", attb, tab), ("attention forward", attf, taf), ("layernorm forward", lnf, tln)):
        ng = 300
        ns = max(4, int(ng * tg / ts))
        eg = run(G, gemm, ng); es = run(S, fn, ns)
        torch.cuda.synchronize()
        ug = eg[0].elapsed_time(eg[1]) / ng * 1e3; us = es[0].elapsed_time(es[1]) / ns * 1e3
        print(f"  together with {tag:20s}: gemm {ug:7.1f} us ({t_g / ug:.2f} of whole-chip rate), {tag} {us:8.1f} us ({[t_ab, t_af, t_ln][['attention backward', 'attention forward', 'layernorm forward'].index(tag)] / us:.2f} of whole-chip rate)", flush=True)
    _lib.lib().pero_set_option(b"gemm_e_var", 0)
