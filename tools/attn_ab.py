#!/usr/bin/env python3
"""Attention backward at the step's shape (N lines x 256 x 4 heads x 128): two launches vs the paired launch (A/B in one process).
usage: python tools/attn_ab.py [N=1024] [iters=20] [only=-1 (A/B) | 0 | 1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
from pero_pretraining_amd._lib import call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
s, h, hd = 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
db = torch.zeros(3 * d, device="cuda")
def run(): return ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec)
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {}
for mode in ((0, 1, 0, 1) if only < 0 else (only,)):
    call("pero_set_option", b"attn_bwd_pair", mode)
    db.zero_()
    g = run(); torch.cuda.synchronize()
    res.setdefault(mode, (g.clone(), db.clone()))
    t = bench(run)
    tf = bench(lambda: ops.attention_fwd_fused(qkv, n, s, h))
    print(f"pair={mode}: bwd {t:8.1f} us   fwd {tf:8.1f} us", flush=True)
for m in res:
    if m: print(m, "dqkv equal to mode 0:", torch.equal(res[0][0], res[m][0]), float((res[0][0].float() - res[m][0].float()).abs().max()),
                " dbias max diff:", float((res[0][1] - res[m][1]).abs().max()))
