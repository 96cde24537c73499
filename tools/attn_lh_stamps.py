#!/usr/bin/env python3
"""Where a unit of attn_bwd_lh_k spends its time: s_memtime stamps (100 MHz) of wave 0 in each workgroup's third unit (diagnostic build
tools/abl/libattn_stamp.so = csrc/attention.hip with -DLH_STAMP).  Prints the median over workgroups of every interval."""
import ctypes, os, sys, torch, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from pero_pretraining_amd import ops
n, s, h, hd = 1024, 256, 4, 128
d = h * hd
torch.manual_seed(0)
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16()
dout = torch.randn(n * s, d, device="cuda").bfloat16()
out, lse = ops.attention_fwd_fused(qkv, n, s, h)
dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
dqkv = torch.empty_like(qkv)
dbias = torch.zeros(3 * d, device="cuda")
work = torch.zeros(3 * n * h * 2 * 128, device="cuda")
lib = ctypes.CDLL(os.path.join(R, "tools/abl/libattn_stamp.so"))
V = ctypes.c_void_p
ctypes.c_int.in_dll(lib, "g_attn_lh").value = 1
f = lib.pero_attention_bwd
f.argtypes = [V] * 8 + [ctypes.c_int64] * 4 + [ctypes.c_int, V]
args = (qkv.data_ptr(), None, dout.data_ptr(), lse.data_ptr(), dvec.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), work.data_ptr(), n, s, h, hd, 1,
        torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert f(*args) == 0
torch.cuda.synchronize()
G = 256
raw = work[3 * n * h * 128:3 * n * h * 128 + G * 128].view(torch.int32).cpu().numpy().view("uint64").reshape(G, 64)
names = {}
for hh in range(4):
    names[3 * hh] = f"half {hh}: top"; names[3 * hh + 1] = f"half {hh}: wait+barrier done"; names[3 * hh + 2] = f"half {hh}: MFMAs done"
names[12] = "end-Q barrier done"; names[13] = "epilogue Q done"
for j in range(8):
    names[14 + 3 * j] = f"stage {j}: top"; names[15 + 3 * j] = f"stage {j}: wait+barrier done"; names[16 + 3 * j] = f"stage {j}: MFMAs done"
names[38] = "end-K barrier done"; names[39] = "dK stored"; names[40] = "dV stored"; names[41] = "partials reduced"
idx = sorted(names)
print("median over workgroups, in s_memtime ticks (10 ns); wave 0 of the third unit")
prev = None
tot = statistics.median(int(raw[g, 41]) - int(raw[g, 0]) for g in range(G))
for i in idx:
    if prev is not None:
        dt = statistics.median(int(raw[g, i]) - int(raw[g, prev]) for g in range(G))
        print(f"  {names[prev]:32s} -> {names[i]:32s} {dt * 10:8.0f} ns")
    prev = i
print(f"unit total {tot * 10:.0f} ns")
