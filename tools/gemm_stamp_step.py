#!/usr/bin/env python3
"""Phase shares of gemm_bf16_s128 inside the real training step (library built with EXTRA=-DPERO_GEMM_STAMP)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib, functional as F
F.SIDE_STREAM_DW = False
h = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, 128, dev)
for i in range(3):
    sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
torch.cuda.synchronize(); h.pero_debug_read_stamps(buf, 1)
for i in range(2):
    sched.update_learning_rate(i); trainer.train_step_prepared(*batches[i % 2])
torch.cuda.synchronize(); h.pero_debug_read_stamps(buf, 1)
steps = buf[5] or 1
names = ["vmcnt wait", "barrier", "glds issue", "ds_read+land", "16 MFMA"]
tot = sum(buf[i] for i in range(5))
print("in-step s128 cycles per wave-k-step: " + ", ".join(f"{n} {buf[i]/steps:.0f} ({100*buf[i]/tot:.0f}%)" for i, n in enumerate(names)) + f" | total {tot/steps:.0f}")
