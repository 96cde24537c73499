#!/usr/bin/env python3
"""What the chip draws and clocks under each of the step's kernel families, and how the tile GEMM scales when it is given fewer CUs
(`gemm_e_var` bits 8-15 cap the persistent workgroups): is the step power-bound, and would GEMM on a CU subset beside attention /
LayerNorm on the rest use the board power better?   usage: python tools/power_probe.py"""
import os, re, subprocess, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops, _lib

def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return str(e)
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
    pw = re.search(r"Power \(W\): ([\d.]+)", t)
    return f"sclk {sclk.group(1) if sclk else '?'} MHz, {pw.group(1) if pw else '?'} W"

def sustained(tag, fn, work=None, secs=2.5):
    """run fn back to back for `secs`, sample smi while the queue is full, return us per call"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = time.perf_counter() - t0
    n = max(8, int(secs / max(one, 1e-5)))
    e0.record()
    for _ in range(n): fn()
    e1.record()
    time.sleep(secs * 0.5)
    s = smi()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    extra = f" {work / us / 1e6:7.1f} TF/s" if work else ""
    print(f"{tag:42s} {us:8.1f} us{extra} | {s}", flush=True)
    return us

torch.manual_seed(0)
M = 262144
print("idle:", smi(), flush=True)
# ---- tile GEMM by CU count
x = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16(); w = (torch.randn(2048, 512, device="cuda") * 0.5).bfloat16(); bias = torch.randn(2048, device="cuda")
out = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16)
x2 = (torch.randn(M, 2048, device="cuda") * 0.5).bfloat16(); w2 = (torch.randn(512, 2048, device="cuda") * 0.5).bfloat16(); out2 = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
_lib.lib().pero_set_option(b"gemm_policy", 20)
for cap in (0, 28, 24, 20, 16):
    _lib.lib().pero_set_option(b"gemm_e_var", cap << 8)
    sustained(f"gemm 262144x2048x512 relu, {cap * 8 or 256} workgroups", lambda: ops.gemm(x, w, bias=bias, relu=True, out=out), 2.0 * M * 2048 * 512)
    sustained(f"gemm 262144x512x2048, {cap * 8 or 256} workgroups", lambda: ops.gemm(x2, w2, out=out2), 2.0 * M * 2048 * 512)
_lib.lib().pero_set_option(b"gemm_e_var", 0)
del x2, w2, out2, out
# ---- attention, LayerNorm
n, s, h, hd = 1024, 256, 4, 128
d = h * hd
qkv = (torch.randn(n * s, 3 * d, device="cuda") * 0.7).bfloat16(); dout = torch.randn(n * s, d, device="cuda").bfloat16()
o, lse = ops.attention_fwd_fused(qkv, n, s, h)
fl = 4.0 * s * s * hd * n * h
sustained("attention forward, 1024 lines", lambda: ops.attention_fwd_fused(qkv, n, s, h), fl)
db = torch.zeros(3 * d, device="cuda")
sustained("attention backward, 1024 lines", lambda: ops.attention_bwd_fused(qkv, o, dout, lse, n, s, h, dbias=db), 3.5 * fl)
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
xin = torch.randn(n * s, d, device="cuda").bfloat16()
sustained("layernorm forward 262144 x 512", lambda: ops.layernorm_fwd(xin, g, b, 1e-5))
