#!/usr/bin/env python3
"""Front-end kernel (u8 NHWC lines -> bf16 patch rows, pitch 1024) timing at B lines of 40 x 2048 (default 256 and 1024)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import ops
def run(B):
  img = torch.randint(0, 256, (B, 40, 2048, 3), dtype=torch.uint8, device="cuda")
  mask = (torch.rand(B, 256, device="cuda") < 0.15).long()
  tile = torch.rand(3, 40, 8, device="cuda")
  def bench(fn, iters=20):
      for _ in range(3): fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters): fn()
      e1.record(); torch.cuda.synchronize()
      return e0.elapsed_time(e1) / iters * 1e3
  t = bench(lambda: ops.patches_from_u8(img, mask, tile, 8, torch.bfloat16, pitch=1024))
  byt = img.numel() + B * 256 * 1024 * 2
  print(f"B = {B}: patches_u8 (bf16, pitch 1024): {t:.1f} us  {byt / t / 1e6:.2f} TB/s ({byt / 1e6:.0f} MB)")

for B_ in ([int(a) for a in sys.argv[1:]] or [256, 1024]):
    run(B_)
