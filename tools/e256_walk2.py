#!/usr/bin/env python3
"""The step's stored products on gemm_bf16_e256 by walk (gemm_e_var bits 16-19 = N-tiles of a row panel per workgroup, one after the other)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
M = 524288
torch.manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[2]
cases = []
x512 = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16()
for name, N, K, seqs in (("linear1 relu+bits", 2048, 512, (0, 2, 4, 18, 20)), ("linear1 relu+bits tiled", 2048, 512, (0, 2, 4)), ("linear2 dX gate+colsum tiled", 2048, 512, (0, 4)), ("linear1 relu", 2048, 512, (0, 4, 20)), ("linear2 dX gate+colsum", 2048, 512, (0, 4, 20)), ("linear2 dX gate", 2048, 512, (0, 4, 20)), ("in_proj bias", 1536, 512, (0, 3, 19)),
                         ("plain", 2048, 512, (0, 4, 20))):
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bits = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
    gbits = torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8)
    cs = torch.zeros(N, device="cuda")
    y = torch.randn(M, N, device="cuda").bfloat16() if "rowdot" in name else None
    dots = torch.zeros(M * (N // 128), device="cuda") if y is not None else None
    if name == "linear1 relu": fn = lambda: ops.gemm(x512, w, out, bias=bias, relu=True)
    elif name == "linear2 dX gate": fn = lambda: ops.gemm(x512, w, out, relu_bits=gbits)
    elif name == "linear1 relu+bits tiled": fn = lambda: ops.gemm(x512, w, out, bias=bias, relu=True, relu_bits=bits, bits_tiled=True)
    elif name == "linear2 dX gate+colsum tiled": fn = lambda: ops.gemm(x512, w, out, relu_bits=gbits, colsum_into=cs, bits_tiled=True)
    elif "relu" in name: fn = lambda: ops.gemm(x512, w, out, bias=bias, relu=True, relu_bits=bits)
    elif "gate" in name: fn = lambda: ops.gemm(x512, w, out, relu_bits=gbits, colsum_into=cs)
    elif "rowdot" in name: fn = lambda: ops.gemm(x512, w, out, rowdot=(y, dots))
    elif "bias" in name: fn = lambda: ops.gemm(x512, w, out, bias=bias)
    else: fn = lambda: ops.gemm(x512, w, out)
    fl = 2.0 * M * N * K
    ref = None
    line = []
    for rep in range(2):
        for seq in seqs:
            _lib.call("pero_set_option", b"gemm_e_walk", 0)
            _lib.call("pero_set_option", b"gemm_e_var", seq << 16)
            t = bench(fn)
            if ref is None: ref = out.clone()
            line.append(f"seq {seq}: {t:.0f} us ({fl / t / 1e6:.0f}){'' if torch.equal(out, ref) else ' DIFFERENT'}")
    print(f"{name} [{N} x {K}]: " + " | ".join(line), flush=True)
    del w, out, bits, gbits, y, dots
_lib.call("pero_set_option", b"gemm_e_var", 0); _lib.call("pero_set_option", b"gemm_e_walk", 1)
