#!/usr/bin/env python3
"""Times pero_attention_bwd (paired launch, D handed in, bias gradient on) of the ablation builds tools/abl/libattn_<mask>.so at the
bench shape (1024 lines x 256 positions x 4 heads x 128).  Mask bits: 1 no loop DMA, 2 no exp, 4 no gradient MFMAs, 8 no output
tiles, 16 no score MFMAs."""
import ctypes, glob, os, re, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, s, h, hd = 1024, 256, 4, 128
d = h * hd
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
sys.path.insert(0, R)
from pero_pretraining_amd import ops
out, lse = ops.attention_fwd_fused(qkv, n, s, h)          # realistic statistics: random ones make P overflow, and NaN / Inf rows skew the timing
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
del out
dqkv = torch.empty_like(qkv)
dbias = torch.zeros(3 * d, device="cuda")
work = torch.empty(3 * n * h * (s // 128) * 128 * 2, device="cuda")
V = ctypes.c_void_p
paths = sorted(glob.glob(os.path.join(R, os.environ.get("PERO_ABL_DIR", "tools/abl"), "libattn_*.so")), key=lambda p: int(re.findall(r"_(\d+)\.so", p)[0]))
# the production library as the reference point, then every build twice (clocks ramp up over the first seconds)
for _ in range(20):
    ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=dbias, dvec=dvec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=dbias, dvec=dvec)
e1.record(); torch.cuda.synchronize()
print(f"{'libpero_hip.so':22s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per backward (1024 lines)", flush=True)
for path in paths + paths:
    lib = ctypes.CDLL(path)
    f = lib.pero_attention_bwd
    f.argtypes = [V] * 8 + [ctypes.c_int64] * 4 + [ctypes.c_int, V]
    st = torch.cuda.current_stream().cuda_stream
    args = (qkv.data_ptr(), None, dout.data_ptr(), lse.data_ptr(), dvec.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), work.data_ptr(), n, s, h, hd, 1, st)
    for _ in range(3):
        assert f(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f(*args)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(path):22s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per backward (1024 lines)", flush=True)
