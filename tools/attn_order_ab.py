#!/usr/bin/env python3
"""The paired attention backward by dispatch order of its blocks (pero_set_option("attn_order", n)), 2048 lines; bit equality + time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
n, s, h, hd = 2048, 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
del out
dbias = torch.zeros(3 * d, device="cuda")
ref = None
for rep in range(5):
    for order in (32, 132, 232, 332):
        _lib.call("pero_set_option", b"attn_order", order)
        for _ in range(3): g = ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=dbias, dvec=dvec)
        torch.cuda.synchronize()
        if ref is None: ref = g.clone()
        same = torch.equal(g, ref)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=dbias, dvec=dvec)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"order {order}: {sorted(ts)[1]:.0f} us per backward (2048 lines) identical {same}", flush=True)
_lib.call("pero_set_option", b"attn_order", 32)
