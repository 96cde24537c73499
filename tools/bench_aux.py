#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs on one GPU: VQ codebook argmin (config 3), VICReg and NT-Xent
joint-embedding steps (configs 4/5 per-GPU share).  usage: python tools/bench_aux.py [lines]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pero_pretraining_amd as P
from pero_pretraining_amd import ops
from pero_pretraining_amd.joint_embedding_pretraining import model as J
from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss, VICRegLoss
from pero_pretraining_amd.joint_embedding_pretraining.trainer import Trainer
from pero_pretraining_amd.optim import FusedAdam
from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# ---- config 3: codebook 8192 x 512, rows = lines * 256
M, K, D = 128 * 256, 8192, 512
x = torch.randn(M, D, device=dev); e = torch.randn(K, D, device=dev)
t = timeit(lambda: ops.vq_argmin(x, e))
print(f"vq_argmin M={M} K={K} D={D}: {t*1e3:.2f} ms  {2.0*M*K*D/t/1e12:.1f} TFLOP/s (f32 MFMA peak 157)  {M/256/t:.0f} lines/s")

# ---- joint embedding steps
rng = np.random.default_rng(0)
def joint(loss, shift):
    torch.manual_seed(0)
    bb = J.init_backbone({"num_blocks": 12, "model_dim": 512, "num_heads": 4, "feedforward_dim": 2048})
    hd = J.init_head({"type": "linear", "in_features": 512, "out_features": 4096})
    model = J.JointEmbeddingTransformerEncoder(bb, hd, loss).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    tr = Trainer(None, model, None, opt, WarmupSchleduler(opt, 1e-4, 100, 1), bfloat16=True)
    S = 256
    im1 = torch.from_numpy(rng.integers(0, 256, (B, 40, 2048, 3), dtype=np.uint8)).to(dev)
    im2 = torch.from_numpy(rng.integers(0, 256, (B, 40, 2048, 3), dtype=np.uint8)).to(dev)
    ones = torch.ones((B, S), dtype=torch.uint8, device=dev)
    sm1 = ones.clone(); sm2 = ones.clone()
    if shift:
        sm1[:, :4] = 0; sm2 = sm1.flip(1).contiguous()
    t = timeit(lambda: tr.train_step_prepared(im1, im2, ones, ones, sm1, sm2), n=4, warm=2)
    return t
t = joint(VICRegLoss(), True)
print(f"VICReg joint step (12-layer d512, linear head 4096, B={B} line pairs): {t*1e3:.2f} ms  {B/t:.0f} line-pairs/s")
t = joint(NTXentLoss(), False)
print(f"NT-Xent joint step (B={B} line pairs): {t*1e3:.2f} ms  {B/t:.0f} line-pairs/s")
