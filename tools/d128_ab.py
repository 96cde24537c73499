#!/usr/bin/env python3
"""gemm_bf16_d128 (256 x 128 tiles, two workgroups per CU) against gemm_bf16_e256 on the step's K = 512 products: bit equality and time.
usage: python tools/d128_ab.py [rows=524288]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
torch.manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[1]
def both(name, fn, fl):
    res = []
    for opt in (0, 8):
        _lib.call("pero_set_option", b"gemm_d128", opt)
        outs = fn()
        torch.cuda.synchronize()
        res.append(([o.clone() for o in outs], bench(fn)))
    _lib.call("pero_set_option", b"gemm_d128", 0)
    same = all(torch.equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    print(f"{name}: e256 {res[0][1]:.0f} us ({fl / res[0][1] / 1e6:.0f} TF/s) | d128 {res[1][1]:.0f} us ({fl / res[1][1] / 1e6:.0f} TF/s) | identical {same}", flush=True)
    if not same:
        for a, b in zip(res[0][0], res[1][0]):
            d = (a.float() - b.float()).abs()
            print("   max diff", d.max().item(), "mismatches", (a != b).sum().item(), "of", a.numel(), "first", (a != b).nonzero()[:4].tolist())
for N, K in ((2048, 512), (1536, 512), (512, 512)):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    fl = 2.0 * M * N * K
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bits = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
    both(f"N={N} K={K} plain+bias", lambda: (ops.gemm(x, w, out, bias=bias),), fl)
    both(f"N={N} K={K} relu", lambda: (ops.gemm(x, w, out, bias=bias, relu=True),), fl)
    both(f"N={N} K={K} relu+bits", lambda: (ops.gemm(x, w, out, bias=bias, relu=True, relu_bits=bits), bits), fl)
    gbits = torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8)
    cs = torch.zeros(N, device="cuda")
    def gate():
        cs.zero_()
        return (ops.gemm(x, w, out, relu_bits=gbits, colsum_into=cs), cs)
    both(f"N={N} K={K} gate+colsum", gate, fl)
    both(f"N={N} K={K} gate", lambda: (ops.gemm(x, w, out, relu_bits=gbits),), fl)
