#!/usr/bin/env python3
"""Does the tile GEMM's micro-benchmark rate survive cold operands?  The same product on ONE buffer set (repeated launches
find the 67 MB operand in the 256 MB Infinity Cache) against a rotation over NB distinct buffer sets (as in the step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib
from pero_pretraining_amd._lib import GEMM_TILE_V
M, NB = 65536, 12
def bench(fns, iters=24):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K, tag) in [(1536, 512, "qkv"), (512, 512, "out"), (2048, 512, "ffn1"), (512, 2048, "ffn2")]:
    sets = [((torch.randn(M, K, device="cuda") * 0.5).bfloat16(), (torch.randn(N, K, device="cuda") * 0.5).bfloat16(),
             torch.empty(M, N, device="cuda", dtype=torch.bfloat16), (torch.randn(M, N, device="cuda") * 0.5).bfloat16(),
             torch.empty(M, K, device="cuda", dtype=torch.bfloat16)) for _ in range(NB)]
    fl = 2.0 * M * N * K
    for name, mk in [("NT fwd (v256)", lambda s: (lambda: ops.gemm(s[0], s[1], out=s[2], extra_flags=GEMM_TILE_V))),
                     ("NN dX (r256)", lambda s: (lambda: ops.gemm(s[3], s[1], out=s[4], trans_b=True)))]:
        hot = bench([mk(sets[0])]); cold = bench([mk(s) for s in sets])
        print(f"{tag:5s} {name:14s}: one buffer set {hot:6.1f} us {fl/hot/1e6:5.0f} TF | {NB} sets in rotation {cold:6.1f} us {fl/cold/1e6:5.0f} TF")
    del sets
