#!/usr/bin/env python3
"""A/B of the attention kernels' bodies (pero_set_option "attn_pipe": 1 = software-pipelined operand reads, 0 = compiler-scheduled) at the
bench shape: 1024 lines x 256 positions x 4 heads x 128, realistic inputs (lse from the forward, D from dO * O).  Checks that both give
the same bits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
from pero_pretraining_amd._lib import call
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n, s, h, hd = int(os.environ.get("N", 1024)), 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
res = {}
for pipe in (0, 1, 0, 1):
    call("pero_set_option", b"attn_pipe", pipe)
    out, lse = ops.attention_fwd_fused(qkv, n, s, h)
    dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
    db = torch.zeros(3 * d, device="cuda")
    dq = ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec)
    dq2 = ops.attention_bwd_fused(qkv, out, dout, lse, n, s, h)   # two launches, D computed by the dQ kernel
    torch.cuda.synchronize()
    tf = bench(lambda: ops.attention_fwd_fused(qkv, n, s, h))
    tb = bench(lambda: ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec))
    tb2 = bench(lambda: ops.attention_bwd_fused(qkv, out, dout, lse, n, s, h))
    print(f"attn_pipe={pipe}: fwd {tf:7.1f} us | bwd paired (+bias grad) {tb:7.1f} us | bwd two launches {tb2:7.1f} us", flush=True)
    if pipe in res:
        continue
    res[pipe] = (out, lse, dq, dq2)
for name, i in (("out", 0), ("lse", 1), ("dqkv paired", 2), ("dqkv two launches", 3)):
    a, b = res[0][i], res[1][i]
    print(f"{name}: bit-identical = {torch.equal(a, b)}; max abs diff {float((a.float() - b.float()).abs().max()):.3e}; finite = {bool(torch.isfinite(b.float()).all())}")
