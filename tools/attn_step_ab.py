#!/usr/bin/env python3
"""Attention backward at the step's shape (2048 lines, S = 256, 4 heads of 128; D handed in, bias gradient on) - for same-box A/B of two
library builds through tools/lib_ab.py:  python tools/lib_ab.py tools/ab_tmp/base.so tools/attn_step_ab.py ; python tools/attn_step_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
n, s, h, hd = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048), 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = (torch.randn(n * s, d, device="cuda") * 0.1).bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).view(n * s, h, hd).sum(-1).contiguous()
db = torch.zeros(3 * d, device="cuda")
def run(): return ops.attention_bwd_fused(qkv, out, dout, lse, n, s, h, dbias=db, dvec=dvec)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
db.zero_()
dq = run()
want = dq.float().sum(0)
print(f"attention backward {n} lines: {sorted(ts)[2]:.1f} us (min {min(ts):.1f}); bias gradient vs column sums of dqkv: max rel err {float((db - want).abs().max() / want.abs().max()):.2e}; checksum {float(dq.float().abs().sum()):.6e}")
