import os, sys, torch
sys.path.insert(0, "/root/repo")
from pero_pretraining_amd import ops, _lib
M = 65536
_lib.lib().pero_set_option(b"gemm_policy", 8)
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
MF, GL, LD = 1 << 12, 1 << 13, 1 << 14
for N in (512, 2048):
  for K in (64, 256, 512, 1024, 2048):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a = bench(lambda: ops.gemm_raw(x, w, y, M, N, K, K, K, N, flags=0))
    b = bench(lambda: ops.gemm_raw(x, w, y, M, N, K, K, K, N, flags=MF | GL | LD))
    c = bench(lambda: ops.gemm_raw(x, w, y, M, N, K, K, K, N, flags=GL | LD))
    print(f"N={N} K={K}: full {a:7.1f} us | nothing {b:7.1f} us | mfma only {c:7.1f} us | mfma peak time {2.0*M*N*K/2.5e9:6.1f} us")
