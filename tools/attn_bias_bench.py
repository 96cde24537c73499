#!/usr/bin/env python3
"""Attention backward at the step's shape with and without the fused in_proj bias gradient (cost of the partial rows + reduce kernel)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from pero_pretraining_amd import ops
n, s, h, hd = 1024, 256, 4, 128
d = h * hd
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16(); dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
db = torch.zeros(3 * d, device="cuda")
def run(dbias): return ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=dbias, dvec=dvec)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
from pero_pretraining_amd._lib import call
bench(lambda: run(db), 40)   # the first timed loop of a process runs ~10 % slow (clocks): spend it here
for mode in (1, 0, 1, 0):
    call("pero_set_option", b"attn_bwd_pair", mode)
    print("one launch" if mode else "two launches", "with bias gradient", bench(lambda: run(db)), "without", bench(lambda: run(None)), flush=True)
call("pero_set_option", b"attn_bwd_pair", 1)
g = run(db.zero_()); print(float((db - g.float().sum(0)).abs().max()), float(db.abs().max()))
