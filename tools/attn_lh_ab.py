#!/usr/bin/env python3
"""The persistent (line, head) attention backward (attn_bwd_lh_k, pero_set_option "attn_lh" 1) against the paired two-workgroups-per-CU
kernels (0): same bits in dqkv, bias gradient within rounding, and the time of both.  N from argv (default 1024 lines), S = 256, 4 heads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
from pero_pretraining_amd._lib import call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s, h, hd = 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
res = {}
for lh in (0, 1):
    call("pero_set_option", b"attn_lh", lh)
    db = torch.zeros(3 * d, device="cuda")
    dq = ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec)
    torch.cuda.synchronize()
    res[lh] = (dq, db)
    print(f"attn_lh={lh}: ran, finite={bool(torch.isfinite(dq.float()).all())}", flush=True)
a, b = res[0], res[1]
print("dqkv bit-identical:", torch.equal(a[0], b[0]), " max abs diff", float((a[0].float() - b[0].float()).abs().max()))
for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
    print(f"  {name}: equal={torch.equal(a[0][:, sl], b[0][:, sl])}")
print("dbias max abs diff", float((a[1] - b[1]).abs().max()), "of", float(a[1].abs().max()))
if iters > 0:
    for lh in (0, 1, 0, 1):
        call("pero_set_option", b"attn_lh", lh)
        db = torch.zeros(3 * d, device="cuda")
        for _ in range(5):
            ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec)
        e1.record(); torch.cuda.synchronize()
        print(f"attn_lh={lh}: {e0.elapsed_time(e1) / iters * 1e3:8.1f} us per backward ({n} lines)", flush=True)
call("pero_set_option", b"attn_lh", 1)
