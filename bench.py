#!/usr/bin/env python3
"""Throughput of the masked-pretraining step (BASELINE.json configs[1]): 12-layer d=512 ViT over
40x2048 synthetic uint8 lines, V=4096, bf16 MFMA, one process per GPU.

  python bench.py                                   (1 GPU, SURVEY 8d protocol: 20 warm-up, 5 repeats of 100 steps, median)
  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...                      (no RANK in the environment: starts the N ranks itself, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W        (the driver's form; one rank per GPU over RCCL)

A step = BatchOperator.prepare_batch (host mask draw, labels / mask upload; the uint8 line images are ALREADY RESIDENT in HBM) ->
zero_grad -> front end (u8 -> /255 -> mask tile -> patches) -> 12 encoder layers -> head -> masked CE -> backward of all of it ->
[gradient all-reduce] -> fused Adam: the reference's Trainer.train_step (masked_pretraining/trainer.py:53-68).  ONE JSON line (rank 0).

  value               whole-job lines/s of that step: the MEDIAN over `--repeats` repeats of exactly --steps steps, each bracketed by
                      barrier + synchronize (max over ranks)
  resident_step       the same without prepare_batch in the timed region (labels and mask resident too: train_step_prepared only)
  with_h2d            prepare_batch INCLUDING the uint8 batch from pinned host memory over PCIe (copy stream, double-buffered) - never `value`
  roofline            the bf16 tile GEMM: algorithmic flops / HIP-event duration of every launch of two extra steps
  hbm_kernels         the bandwidth-bound kernels of the step alone at the step's shapes (HIP events): algorithmic GB, us, TB/s, fraction of 8 TB/s
  batch_sweep         the same step at 16 / 64 / 128 / 1024 lines per GPU (the reference's default --batch-size is 16), eager and as a hipGraph
  legs                configs[2] (codebook argmin alone, and the V = 8192 masked step with its labels from the argmin inside the step), configs[3]
                      (VICReg step; under N > 1 with data-parallel gradients, per-rank and exact global statistics), configs[4] (NT-Xent step,
                      512 lines per GPU; under N > 1 with cross-rank negatives)
  cpu_baseline        the CPU oracle (oracle/pero_oracle.py, a "port") on the host cores, a bounded sample of the same workload
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
F32_MFMA_PEAK_TFLOPS = 157.3    # same table, "Peak FP32 (matrix)"

CFG = dict(num_blocks=12, model_dim=512, num_heads=4, feedforward_dim=2048, vocab=4096, width=2048, height=40,
           patch=8, channels=3)


def flops_per_line(c=CFG, masked_frac=None, last_layer_rows=False):
    """forward F = patch + L*(qkv + attn + out + ffn) + head; step = 3F.  With `masked_frac` the EXECUTED count:
    the head's two backward products run on the masked rows only (the other rows of dlogits are exact zeros); with `last_layer_rows` so do the
    backward products of the LAST layer's feed-forward block and out-projection (functional.ROW_SPARSE_LAST_LAYER)."""
    S, d, L, ff, V = c["width"] // c["patch"], c["model_dim"], c["num_blocks"], c["feedforward_dim"], c["vocab"]
    kp = c["channels"] * c["height"] * c["patch"]
    head = 2 * S * d * V
    F = 2 * S * kp * d + L * (2 * S * d * 3 * d + 4 * S * S * d + 2 * S * d * d + 4 * S * d * ff) + head
    if masked_frac is None:
        return 3 * F
    rows = 2 * (2 * S * d * d + 4 * S * d * ff) * (1.0 - masked_frac) if last_layer_rows else 0.0
    return 3 * F - 2 * head * (1.0 - masked_frac) - rows


def csrc_hash():
    """sha256 over the kernel sources: a committed PMC profile is only quoted while it describes THIS code."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pero_pretraining_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(batch):
    """HBM bytes per GEMM launch from the committed rocprofv3 --pmc summary (profiles/, separate FETCH_SIZE / WRITE_SIZE passes of
    this command: tools/pmc_traffic.sh).  bench.py cannot run the profiler on itself, so the number is quoted ONLY when the summary
    was collected for this batch size from exactly these kernel sources (`csrc_sha256`); otherwise null."""
    here = csrc_hash()
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (name.endswith("pmc_gemm_traffic.json")):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if int(d.get("lines_per_gpu", -1)) == int(batch) and d.get("csrc_sha256") == here:
            return d.get("hbm_bytes_per_launch"), name
    return None, None


def pmc_leg_traffic(leg):
    """HBM bytes per launch of a leg's dominant kernel family from the committed profiles/*pmc_legs_traffic.json (tools/pmc_legs.sh: separate
    FETCH_SIZE / WRITE_SIZE passes of `bench.py --legs-only --leg <name>`), quoted only for exactly these kernel sources; else (None, None)."""
    here = csrc_hash()
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not name.endswith("pmc_legs_traffic.json"):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("csrc_sha256") == here and leg in d.get("legs", {}):
            return d["legs"][leg].get("hbm_bytes_per_launch"), name
    return None, None


# ------------------------------------------------------------------------------------------------ launching
def self_launch(args, argv):
    """`python bench.py --gpus N` without RANK in the environment: start the N ranks as a CHILD process
    (python -m torch.distributed.run ... bench.py ...) BEFORE anything here touches the GPU, relay its one JSON line and exit with its
    code.  The parent never initialises HIP (a process that has must not exec or be replaced on this pool)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    child_argv = [a for a in argv if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + child_argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PERO_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.exit(proc.returncode if proc.returncode or line is not None else 1)


# ------------------------------------------------------------------------------------------------ model / data
def build(device, bf16):
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining import model as M
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    torch.manual_seed(0)
    bb = M.init_backbone({"type": "vit", "num_blocks": CFG["num_blocks"], "model_dim": CFG["model_dim"],
                          "num_heads": CFG["num_heads"], "feedforward_dim": CFG["feedforward_dim"]})
    hd = M.init_head({"in_features": CFG["model_dim"], "out_features": CFG["vocab"]})
    model = M.MaskedTransformerEncoder(bb, hd).to(device).train()
    opt = FusedAdam(model.parameters(), lr=2e-4)
    sched = WarmupSchleduler(opt, 2e-4, 10000, 1)
    trainer = Trainer(BatchOperator(device, 0.15), model, None, opt, sched, bfloat16=bf16)
    return model, opt, sched, trainer


def synthetic(rank, B, device, nbatches=2):
    """SURVEY 8d synthetic inputs.  Per batch: the uint8 images on the device (resident in HBM), the labels on the host AND on the
    device, a pre-drawn mask on the device (for the resident_step leg only; `value` draws its masks in prepare_batch)."""
    rng = np.random.default_rng(1234 + rank)
    W, S = CFG["width"], CFG["width"] // CFG["patch"]
    out = []
    for _ in range(nbatches):
        images = torch.from_numpy(rng.integers(0, 256, (B, CFG["height"], W, CFG["channels"]), dtype=np.uint8)).to(device)
        labels_h = rng.integers(0, CFG["vocab"], (B, S)).astype(np.int64)
        mask_h = (rng.random((B, S)) < 0.15).astype(np.int64)
        mask = torch.from_numpy(mask_h).to(device)
        mask._pero_host = mask_h   # what BatchOperator / DevicePrefetcher attach: the host original of an uploaded mask
        out.append({"images": images, "labels": labels_h, "labels_dev": torch.from_numpy(labels_h).to(device), "mask_dev": mask})
    return out


def cpu_baseline(budget_s=20.0, B=8):
    """The CPU oracle's train step (f32, torch CPU primitive ops + autograd) at the config-2 shape."""
    from oracle import pero_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))  # the one-GPU box grants a 16-CPU share; more threads only thrash
    from pero_pretraining_amd.masked_pretraining import model as M
    torch.manual_seed(0)
    bb = M.init_backbone({"type": "vit", "num_blocks": CFG["num_blocks"], "model_dim": CFG["model_dim"],
                          "num_heads": CFG["num_heads"], "feedforward_dim": CFG["feedforward_dim"]})
    hd = M.init_head({"in_features": CFG["model_dim"], "out_features": CFG["vocab"]})
    sd = {k: v.clone() for k, v in M.MaskedTransformerEncoder(bb, hd).state_dict().items()}
    orc = O.MaskedStepOracle(sd, CFG["num_heads"])
    rng = np.random.default_rng(1234)
    S = CFG["width"] // CFG["patch"]
    images = rng.integers(0, 256, (B, CFG["height"], CFG["width"], CFG["channels"]), dtype=np.uint8)
    labels = rng.integers(0, CFG["vocab"], (B, S)).astype(np.int64)
    mask = (rng.random((B, S)) < 0.15).astype(np.int64)
    offsets = rng.integers(0, 4096 - S, B)
    orc.step(images, labels, mask, 2e-4, offsets)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.step(images, labels, mask, 2e-4, offsets)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": round(n * B / el, 3), "unit": "lines/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of B={B} lines (40x2048, 12-layer d=512, f32) through oracle/pero_oracle.py MaskedStepOracle"}


class Timer:
    """K steps bracketed by barrier + synchronize on both sides, max over ranks; median over repeats."""

    def __init__(self, device):
        self.device = device
        self.cuda = device.type == "cuda"

    def fence(self):
        if self.cuda:
            torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        if self.cuda:
            torch.cuda.synchronize()

    def run(self, step, steps, first=0):
        self.fence()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = step(first + i)
        self.fence()
        el = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([el], device=self.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el, out

    def median(self, step, steps, repeats, first=0):
        els, out = [], None
        for r in range(repeats):
            el, out = self.run(step, steps, first + r * steps)
            els.append(el)
        return statistics.median(els), els, out


def timed_gemms(run_steps, nsteps):
    """Every pero_gemm launch of `nsteps` steps bracketed by HIP events on the stream it is launched on (the weight gradients are
    put on the main stream for this leg, so no two kernels share the chip while timed) -> {tag: [seconds, flops, launches]}."""
    from pero_pretraining_amd import functional as F
    from pero_pretraining_amd import ops
    side_was = F.SIDE_STREAM_DW
    F.SIDE_STREAM_DW = False
    run_steps(1)
    torch.cuda.synchronize()
    ops.gemm_timeline = []
    run_steps(nsteps)
    torch.cuda.synchronize()
    tl, ops.gemm_timeline = ops.gemm_timeline, None
    F.SIDE_STREAM_DW = side_was
    per = {}
    for e0, e1, fl, tag in tl:
        d = per.setdefault(tag, [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1) * 1e-3
        d[1] += fl
        d[2] += 1
    return per


def gemm_roofline(per, nsteps, step_seconds, traffic):
    fast = {k: v for k, v in per.items() if k.startswith("gemm_bf16_tile")}
    tsum = sum(v[0] for v in fast.values()) or 1e-30
    fsum = sum(v[1] for v in fast.values())
    nl = sum(v[2] for v in fast.values()) or 1
    ach = fsum / tsum / 1e12
    return {"bound": "mfma",
            "kernel": "bf16 MFMA tile GEMM gemm_bf16_e256 (eight-phase persistent 256x256x64: forward, input gradients with fused "
                      "epilogues, split-K weight gradients) + its row-complete 128x512 form gemm_bf16_n512 (out-projection / linear2 with the "
                      "residual LayerNorm in the epilogue, linear1's / in_proj's input gradients with the LayerNorm BACKWARD in the epilogue: "
                      "by_layout tags *_ln_fwd / *_ln_bwd; the product's flops over the WHOLE launch, LayerNorm work and the column-sum reduce "
                      "included - since round 4 the family carries 47 of the step's 49 LayerNorm passes)",
            "achieved": round(ach, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic, "launches_per_step": nl // nsteps, "avg_launch_us": round(tsum / nl * 1e6, 2),
            "gflop_per_launch": round(fsum / nl / 1e9, 3),
            "gemm_time_share_of_step": round((tsum / nsteps) / step_seconds, 3),
            "by_layout": {k: {"tflops": round(v[1] / v[0] / 1e12, 1), "ms_per_step": round(v[0] / nsteps * 1e3, 3),
                              "launches_per_step": v[2] // nsteps} for k, v in sorted(per.items())}}


HBM_PEAK_TBS = 8.0   # MI355X_MICROARCH.md "HBM3E peak BW" (spec; 6.3 TB/s is what a float4 copy reaches)


def hbm_kernels(device, batch):
    """The bandwidth-bound kernels of the step (front end, LayerNorm, Adam, attention at S = 256), each alone at the step's shapes,
    timed with HIP events on the launching stream: algorithmic bytes / median duration, against the 8 TB/s of the data sheet."""
    from pero_pretraining_amd import ops
    S, d, h = CFG["width"] // CFG["patch"], CFG["model_dim"], CFG["num_heads"]
    rows = batch * S
    g = torch.Generator(device=device).manual_seed(11)

    def timed(fn, n=8, rounds=3):
        """median over `rounds` of (n back-to-back launches between one pair of HIP events) / n"""
        for _ in range(3):
            fn()
        evs = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return statistics.median(a.elapsed_time(b) for a, b in evs) * 1e-3 / n

    out = {}

    def add(name, seconds, nbytes, note):
        out[name] = {"gb": round(nbytes / 1e9, 3), "us": round(seconds * 1e6, 1), "tb_per_s": round(nbytes / seconds / 1e12, 2),
                     "frac_of_8_tb_per_s": round(nbytes / seconds / 1e12 / HBM_PEAK_TBS, 3), "bytes": note}

    # front end: u8 images -> patch rows (bf16, pitch 1024)
    images = torch.randint(0, 256, (batch, CFG["height"], CFG["width"], CFG["channels"]), device=device, dtype=torch.uint8, generator=g)
    mask = (torch.rand(batch, S, device=device, generator=g) < 0.15).long()
    tile = torch.rand(CFG["channels"], CFG["height"], CFG["patch"], device=device, generator=g)
    kp = CFG["channels"] * CFG["height"] * CFG["patch"]
    pitch = ((kp + 127) // 128) * 128
    t = timed(lambda: ops.patches_from_u8(images, mask, tile, CFG["patch"], torch.bfloat16, pitch))
    add("patches_u8 (front end)", t, images.numel() + rows * pitch * 2, "u8 image read + bf16 patch rows (pitch 1024) written")
    del images
    x = torch.randn(rows, d, device=device, generator=g).bfloat16()
    dy = torch.randn(rows, d, device=device, generator=g).bfloat16()
    gamma = torch.rand(d, device=device, generator=g) + 0.5
    beta = torch.randn(d, device=device, generator=g) * 0.1
    pe = torch.randn(4096, d, device=device, generator=g)
    offs = torch.randint(0, 4096 - S, (batch,), device=device, generator=g)
    t = timed(lambda: ops.layernorm_fwd(x, gamma, beta, 1e-5, pe=pe, offsets=offs, S=S))
    add("layernorm_fwd (+ positional rows)", t, rows * d * 2 * 2, "bf16 rows read + written (the f32 positional rows come from the Infinity Cache)")
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    dg, db, dxs = (torch.zeros(d, device=device) for _ in range(3))
    t = timed(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dxs))
    add("layernorm_bwd (from the input rows) + reduce", t, rows * d * 2 * 3, "dy, x read + dx written")
    t = timed(lambda: ops.layernorm_bwd_out(dy, y, rstd, gamma, beta, dg, db, dxs))
    add("layernorm_bwd_out (from the output rows) + reduce", t, rows * d * 2 * 3, "dy, t read + dx written")
    del x, dy, y
    # fused Adam over the flat parameter buffer (config-2 size)
    n = 40422912
    pflat, gflat, m, v = (torch.randn(n, device=device, generator=g) * 0.01 for _ in range(4))
    v.abs_()
    pb = torch.empty(n, device=device, dtype=torch.bfloat16)
    t = timed(lambda: ops.adam_step(pflat, gflat, m, v, pb, 1e-4, 0.9, 0.999, 1e-8, 10))
    add("adam_k (fused, + bf16 weight copy)", t, n * (16 + 14), "p, g, m, v read (16 B) + p, m, v, bf16 copy written (14 B) per parameter")
    del pflat, gflat, m, v, pb
    # attention of one layer
    qkv = (torch.randn(rows, 3 * d, device=device, generator=g) * 0.5).bfloat16()
    t = timed(lambda: ops.attention_fwd_fused(qkv, batch, S, h))
    add("attention forward (one layer)", t, rows * (3 * d + d) * 2, "qkv read + out written (lse: 2 %)")
    out_, lse = ops.attention_fwd_fused(qkv, batch, S, h)
    dout = (torch.randn(rows, d, device=device, generator=g) * 0.1).bfloat16()
    dvec = (out_.float() * dout.float()).view(rows, h, d // h).sum(-1).contiguous()
    dbias = torch.zeros(3 * d, device=device)
    t = timed(lambda: ops.attention_bwd_fused(qkv, out_, dout, lse, batch, S, h, dbias=dbias, dvec=dvec))
    add("attention backward (one layer, D handed in)", t, rows * (3 * d + d + 3 * d) * 2, "qkv, dout read + dqkv written")
    return out


def batch_sweep(device, model, opt, sched, bf16, rank, timer, world):
    """The headline step at small per-GPU batches (the reference's default is --batch-size 16, masked_pretraining/train.py:30):
    labels and mask resident, eager launches and - where a step is launch-bound - the same step replayed as a hipGraph."""
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    res = {}
    for B in (16, 64, 128, 1024):
        batches = synthetic(rank, B, device)
        entry = {}
        for mode in (("eager", "hip_graph") if B <= 128 else ("eager",)):
            tr = Trainer(BatchOperator(device, 0.15), model, None, opt, sched, bfloat16=bf16, hip_graph=(mode == "hip_graph"))

            def step(i):
                sched.update_learning_rate(i)
                b = batches[i % len(batches)]
                return tr.train_step_prepared(b["images"], b["labels_dev"], b["mask_dev"])
            for i in range(3):
                step(i)
            steps = 30 if B <= 128 else 10
            el, rp, _ = timer.median(step, steps, 3, first=3)
            lps = world * B * steps / el
            entry[mode] = {"value": round(lps, 1), "unit": "lines/s", "ms_per_step": round(el / steps * 1e3, 3),
                           "step_mfma_frac_formula_flops": round(lps / world * flops_per_line() / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4)}
            del tr
        res[f"B={B}"] = entry
        del batches
    res["note"] = ("labels and mask resident (train_step_prepared); 3 repeats of 30 (B <= 128) / 10 steps, median; hip_graph = Trainer(hip_graph=True): "
                   "zero_grad + forward + backward replayed as one graph, dense head backward")
    return res


# ------------------------------------------------------------------------------------------------ legs
def leg_config3_step(timer, device, batch, steps, repeats, rank, world):
    """BASELINE.json configs[2] as a STEP: the masked step with a V = 8192 head whose labels are produced INSIDE the step by the codebook argmin
    (pero_vq_argmin, K = 8192 x D = 512, exact f32) from resident encoder features (synthetic standard-normal features and codebook;
    the tokenizer's VGG encoder is out of scope)."""
    from pero_pretraining_amd import ops
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining import model as M
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    torch.manual_seed(0)
    V = 8192
    bb = M.init_backbone({"type": "vit", "num_blocks": CFG["num_blocks"], "model_dim": CFG["model_dim"], "num_heads": CFG["num_heads"],
                          "feedforward_dim": CFG["feedforward_dim"]})
    hd = M.init_head({"in_features": CFG["model_dim"], "out_features": V})
    model = M.MaskedTransformerEncoder(bb, hd).to(device).train()
    opt = FusedAdam(model.parameters(), lr=2e-4)
    sched = WarmupSchleduler(opt, 2e-4, 10000, 1)
    tr = Trainer(BatchOperator(device, 0.15), model, None, opt, sched, bfloat16=True)
    S = CFG["width"] // CFG["patch"]
    rng = np.random.default_rng(77 + rank)
    images = torch.from_numpy(rng.integers(0, 256, (batch, CFG["height"], CFG["width"], CFG["channels"]), dtype=np.uint8)).to(device)
    g = torch.Generator(device=device).manual_seed(3 + rank)
    feats = torch.randn(batch * S, 512, device=device, generator=g)
    code = torch.randn(V, 512, device=device, generator=g)
    mask_h = (rng.random((batch, S)) < 0.15).astype(np.int64)
    mask = torch.from_numpy(mask_h).to(device)
    mask._pero_host = mask_h

    def step(i):
        sched.update_learning_rate(i)
        labels = ops.vq_argmin(feats, code).view(batch, S)      # the tokenizer's quantizer: labels of this batch, on the device
        return tr.train_step_prepared(images, labels, mask)
    for i in range(2):
        step(i)
    med, els, loss_v = timer.median(step, steps, repeats, first=2)
    evs = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.vq_argmin(feats, code); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t_arg = statistics.median(a.elapsed_time(b) for a, b in evs) * 1e-3
    cfg3 = dict(CFG, vocab=V)
    fl = flops_per_line(cfg3) + 2.0 * S * V * 512
    lps = world * batch * steps / med
    del model, opt, tr
    return {"workload": f"masked step with a V = 8192 head, labels = codebook argmin (8192 x 512, f32 exact) of resident features INSIDE the step, "
                        f"{batch} lines of 40x2048 per GPU, bf16 (BASELINE.json configs[2])",
            "ms_per_step": round(med / steps * 1e3, 3), "lines_per_s": round(lps, 1), "repeats_ms_per_step": [round(e / steps * 1e3, 3) for e in els],
            "argmin_ms_of_it": round(t_arg * 1e3, 3), "loss": round(float(loss_v.detach()), 5),
            "gflop_per_line_step": round(fl / 1e9, 3),
            "note": "the argmin's 2 S K D flops per line run on the exact f32 MFMA (157 TFLOP/s peak): the step's bf16 MFMA fraction is not quoted for the sum"}


def leg_config3(timer, device):
    """BASELINE.json configs[2]: codebook argmin of the VQ tokenizer, K = 8192 codes x D = 512, the rows of 128 lines."""
    from pero_pretraining_amd import ops
    M, K, D = 128 * 256, 8192, 512
    g = torch.Generator(device=device).manual_seed(3)
    x = torch.randn(M, D, device=device, generator=g)
    e = torch.randn(K, D, device=device, generator=g)
    for _ in range(3):
        ops.vq_argmin(x, e)
    evs = []
    for _ in range(9):   # one launch per call: HIP events on the launching stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.vq_argmin(x, e); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = statistics.median(a.elapsed_time(b) for a, b in evs) * 1e-3
    ach = 2.0 * M * K * D / t / 1e12
    traffic, tsrc = pmc_leg_traffic("config3_vq_argmin")
    return {"workload": "VQ codebook argmin (BASELINE.json configs[2]): 32768 rows (128 lines x 256) x 8192 codes x 512, f32 exact",
            "ms": round(t * 1e3, 3), "lines_per_s": round(M / 256 / t, 1), "rows_per_s": round(M / t, 1),
            "roofline": {"bound": "mfma", "kernel": "vq_argmin_fast_k (v_mfma_f32_32x32x2_f32, running argmin in registers)",
                         "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": f"profiles/{tsrc}" if tsrc else None}}


def leg_joint(timer, device, kind, pairs, steps, repeats, world, rank, variant=None):
    """BASELINE.json configs[3] / [4], one GPU's share: joint-embedding step (two views through the 12-layer backbone as one 2N
    batch + linear head 4096 + VICReg / NT-Xent + backward + [gradient all-reduce] + fused Adam) on `pairs` line pairs of 40x2048 per
    GPU.  variant: VICReg "global" = exact global statistics over all ranks; NT-Xent "cross" = cross-rank negatives (all-gather)."""
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.joint_embedding_pretraining import model as J
    from pero_pretraining_amd.joint_embedding_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss, VICRegLoss
    from pero_pretraining_amd.joint_embedding_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    from pero_pretraining_amd.parallel import DataParallel
    torch.manual_seed(0)
    bb = J.init_backbone({"num_blocks": 12, "model_dim": 512, "num_heads": 4, "feedforward_dim": 2048})
    hd = J.init_head({"type": "linear", "in_features": 512, "out_features": 4096})
    if kind == "vicreg":
        loss = VICRegLoss(global_statistics=(variant == "global"))
    else:
        loss = NTXentLoss(cross_rank_negatives=(variant == "cross"))
    model = J.JointEmbeddingTransformerEncoder(bb, hd, loss).to(device).train()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    tr = Trainer(BatchOperator(device), model, None, opt, WarmupSchleduler(opt, 1e-4, 100, 1), bfloat16=True)
    if dist.is_initialized():
        tr.data_parallel = DataParallel(model, opt)
    rng = np.random.default_rng(5 + rank)
    S = 256
    ones = np.ones((pairs, S), np.uint8)
    sm1 = ones.copy()
    sm2 = ones.copy()
    if kind == "vicreg":   # image-shift augmentation: the views overlap on all but 4 positions (SURVEY.md 8d synthetic inputs)
        sm1[:, :4] = 0
        sm2 = sm1[:, ::-1].copy()
    batch = {"images": rng.integers(0, 256, (pairs, 40, 2048, 3), dtype=np.uint8),
             "images2": rng.integers(0, 256, (pairs, 40, 2048, 3), dtype=np.uint8),
             "image_masks": ones, "image_masks2": ones, "shift_masks": sm1, "shift_masks2": sm2}
    prepared = tr.batch_operator.prepare_batch(batch)   # resident in HBM (masks keep their host copies: no device sync)
    for i in range(2):
        tr.train_step_prepared(*prepared)
    med, els, loss_v = timer.median(lambda i: tr.train_step_prepared(*prepared), steps, repeats)
    per = timed_gemms(lambda n: [tr.train_step_prepared(*prepared) for _ in range(n)], 1)
    leg_name = ("config4_vicreg_step" if kind == "vicreg" else "config5_ntxent_step") if world == 1 else None
    traffic, tsrc = pmc_leg_traffic(leg_name) if leg_name else (None, None)
    roof = gemm_roofline(per, 1, med / steps, traffic)
    roof["traffic_source"] = f"profiles/{tsrc}" if tsrc else None
    del model, opt, tr
    name = "VICReg" if kind == "vicreg" else "NT-Xent"
    return {"workload": f"{name} joint-embedding step, 12-layer d=512 ViT + linear head 4096, {pairs} line pairs of 40x2048 per GPU, bf16 "
                        f"(BASELINE.json configs[{3 if kind == 'vicreg' else 4}]" + (", one GPU's share)" if world == 1 else f", {world} GPUs)"),
            "variant": {"vicreg": {None: "per-rank statistics", "global": "exact global statistics (two extra all-reduces: D + 1 floats, D x D f32)"},
                        "ntxent": {None: "per-line negatives (the reference's loss)", "cross": "cross-rank negatives: pooled embeddings all-gathered"}}[kind][variant],
            "ms_per_step": round(med / steps * 1e3, 3), "line_pairs_per_s": round(world * pairs * steps / med, 1),
            "repeats_ms_per_step": [round(e / steps * 1e3, 3) for e in els], "loss": round(float(loss_v.detach()), 5),
            "distributed": {"world_size": world, "backend": dist.get_backend() if dist.is_initialized() else None},
            "roofline": roof}


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # warm-up 20 steps, time >= 100 steps, median of 5 repeats
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PERO_BENCH_BATCH", 2048)), help="lines per GPU")
    ap.add_argument("--no-side-stream", action="store_true", help="weight gradients on the main stream (the default since round 3; kept for the profiling scripts)")
    ap.add_argument("--side-stream", action="store_true", help="weight gradients on a second HIP stream (rounds 1-2's default; measured slower now)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the config 3 / 4 / 5 legs")
    ap.add_argument("--legs-only", action="store_true", help="only the config 3 / 4 / 5 legs (profiling)")
    ap.add_argument("--leg", default=None, help="with --legs-only: only these legs, comma-separated (config3_vq_argmin, config3_masked_step, config4_vicreg_step, config5_ntxent_step)")
    ap.add_argument("--leg-pairs", default="128,512", help="line pairs per GPU of the config 4 / config 5 legs (the tests' rehearsal shrinks them)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the small-batch sweep and the bandwidth-bound kernel block")
    ap.add_argument("--dp-no-overlap", action="store_true", help="data parallel: reduce all gradients after the backward pass (A/B of the bucket hooks)")
    ap.add_argument("--dp-layers-per-bucket", type=int, default=2)
    ap.add_argument("--no-options", action="store_true", help="skip the extra legs (resident step, PCIe-inclusive step, masked head)")
    ap.add_argument("--head-backward", default="masked", choices=["masked", "dense"],
                    help="head backward on the masked rows only (default: the other rows of dlogits are exact zeros) or over all rows")
    ap.add_argument("--dense-last-layer", action="store_true",
                    help="functional.ROW_SPARSE_LAST_LAYER = False for the whole run (the headline then equals option_dense_last_layer of a default run)")
    ap.add_argument("--masked-head", action="store_true",
                    help="OPTION, not the headline: head FORWARD + loss on the masked positions only (model.head_rows = 'masked'); "
                         "the default evaluates the head on every position like the reference")
    ap.add_argument("--spawn", action="store_true", help="start the rank(s) through torch.distributed.run even for --gpus 1 (RCCL with one rank)")
    ap.add_argument("--plumbing-test", action="store_true", help=argparse.SUPPRESS)   # CPU / gloo rehearsal of launch + timing + relay (tests)
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        self_launch(args, sys.argv[1:])     # never returns

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    # Under torch.distributed.run RCCL prints a version banner on the C-level stdout at communicator creation: everything but the ONE
    # JSON line goes to stderr (file descriptor 1 is pointed at stderr; the line is written to the saved descriptor at the end)
    real_stdout = None
    if "RANK" in os.environ:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)

    def emit(out):
        if rank != 0:
            return
        if real_stdout is None:
            print(json.dumps(out))
        else:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.plumbing_test:
        return plumbing_test(args, world, rank, emit)
    # PERO_BENCH_REHEARSE_GLOO=1: rehearsal of the world > 1 code paths on a box with fewer GPUs than ranks - the ranks share the
    # devices that exist and exchange over gloo (RCCL needs one device per rank); the timings mean nothing and the line says so
    rehearse = os.environ.get("PERO_BENCH_REHEARSE_GLOO") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: same code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend() + (" (REHEARSAL: ranks share devices, timings not meaningful)" if rehearse else "")

    from pero_pretraining_amd import functional as F
    if args.dense_last_layer:
        F.ROW_SPARSE_LAST_LAYER = False
    from pero_pretraining_amd.parallel import DataParallel
    F.SIDE_STREAM_DW = bool(args.side_stream) and not args.no_side_stream
    timer = Timer(device)
    bf16 = args.dtype == "bf16"
    out = {"metric": "text-line images/sec (masked-ViT step)", "unit": "lines/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}
    S = CFG["width"] // CFG["patch"]

    if not args.legs_only:
        model, opt, sched, trainer = build(device, bf16)
        model.head_backward = args.head_backward
        if args.masked_head:
            model.head_rows = "masked"
        if dist.is_initialized():
            trainer.data_parallel = DataParallel(model, opt, overlap=not args.dp_no_overlap, layers_per_bucket=args.dp_layers_per_bucket)
        batches = synthetic(rank, args.batch, device)
        np.random.seed(4321 + rank)   # prepare_batch draws its masks from numpy's global stream (the reference's call)

        # the reference's train_step = prepare_batch + the step.  prepare_batch (numpy mask draw, labels and mask to the device) runs
        # for every step inside the timed region, one batch ahead on a copy stream from pinned staging buffers
        # (common/dataloader.DevicePrefetcher: a pageable upload would stall the launching thread); the uint8 images it is
        # handed are already device tensors - resident in HBM - and pass through untouched
        import itertools
        from pero_pretraining_amd.common.dataloader import DevicePrefetcher
        feed = iter(DevicePrefetcher(({"images": b["images"], "labels": b["labels"]} for b in itertools.cycle(batches)),
                                     trainer.batch_operator, device))

        def step(i):
            sched.update_learning_rate(i)
            images, labels, mask = next(feed)
            return trainer.train_step_prepared(images, labels, mask)

        def step_resident(i):
            sched.update_learning_rate(i)
            b = batches[i % len(batches)]
            return trainer.train_step_prepared(b["images"], b["labels_dev"], b["mask_dev"])

        # ---- headline
        for i in range(args.warmup):
            step(i)
        elapsed, repeats, loss = timer.median(step, args.steps, args.repeats, first=args.warmup)
        final_loss = float(loss.detach()) if hasattr(loss, "detach") else float(loss)
        lines_per_s = world * args.batch * args.steps / elapsed
        step_flops = flops_per_line()
        exec_flops = flops_per_line(masked_frac=0.15, last_layer_rows=F.ROW_SPARSE_LAST_LAYER) if (args.head_backward == "masked" and not args.masked_head) else step_flops
        out.update({
            "value": round(lines_per_s, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "config": {"workload": "masked pretraining step incl. prepare_batch, 12-layer d=512 h=4 ff=2048 ViT, V=4096, 40x2048 u8 lines "
                                   "(BASELINE.json configs[1]); uint8 images resident in HBM, the PCIe-inclusive rate is `with_h2d`",
                       "lines_per_gpu": args.batch, "global_batch": args.batch * world, "seq_len": S,
                       "parallelism": f"dp{world}", "optimizer": "fused Adam (f32 master weights)",
                       "weight_gradients_on_side_stream": bool(F.SIDE_STREAM_DW), "head_rows": model.head_rows,
                       "head_backward": args.head_backward,
                       "last_layer_backward_rows": "masked (exact zeros elsewhere; the all-positions form is timed in option_dense_last_layer)" if F.ROW_SPARSE_LAST_LAYER else "all",
                       "gflop_per_line_step": round(step_flops / 1e9, 3), "gflop_per_line_step_executed": round(exec_flops / 1e9, 3)},
            "protocol": {"repeats": args.repeats, "statistic": "median", "repeats_ms_per_step": [round(e / args.steps * 1e3, 3) for e in repeats],
                         "inputs": "uint8 line images resident in HBM; labels from the host and the mask drawn on the host inside the timed "
                                   "step (BatchOperator.prepare_batch)"},
            "distributed": {"world_size": dist.get_world_size() if dist.is_initialized() else 1, "backend": backend,
                            "self_launched": bool(os.environ.get("PERO_BENCH_SELF_LAUNCHED")),
                            "gradient_all_reduce": "bucketed, overlapped with backward" if dist.is_initialized() and not args.dp_no_overlap else
                                                   ("after backward" if dist.is_initialized() else None)},
            # formula FLOPs (3F per line) and EXECUTED FLOPs (head backward on the masked rows only): the MFMA fraction of the
            # step is computed from the executed count - multiplying by exact zeros is not work
            "step_mfma_frac": round(lines_per_s / world * exec_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
            "step_mfma_frac_formula_flops": round(lines_per_s / world * step_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
            "final_loss": round(final_loss, 5)})

        if not args.no_options and not args.masked_head:
            reps = min(args.repeats, 3)
            osteps = min(args.steps, 30)   # the side legs: three repeats of at most 30 steps each (the headline keeps the full protocol)
            # ---- the same step with labels and mask resident too (train_step_prepared only: rounds 1-2's `value`)
            for i in range(2):
                step_resident(i)
            el, rp, _ = timer.median(step_resident, osteps, reps, first=2)
            out["resident_step"] = {"value": round(world * args.batch * osteps / el, 2), "unit": "lines/s", "ms_per_step": round(el / osteps * 1e3, 3),
                                    "repeats_ms_per_step": [round(e / osteps * 1e3, 3) for e in rp],
                                    "note": "prepare_batch outside the timed region (mask pre-drawn, labels resident)"}
            # ---- PCIe-inclusive: the uint8 batch from pinned host memory on a copy stream, double-buffered against the previous step
            from pero_pretraining_amd.common.dataloader import DevicePrefetcher
            rng = np.random.default_rng(99 + rank)
            host = [{"images": rng.integers(0, 256, (args.batch, CFG["height"], CFG["width"], CFG["channels"]), dtype=np.uint8),
                     "labels": rng.integers(0, CFG["vocab"], (args.batch, S)).astype(np.int64)} for _ in range(2)]
            n_total = 2 + reps * osteps
            it = iter(DevicePrefetcher((host[i % 2] for i in range(n_total + 1)), trainer.batch_operator, device))

            def step_h2d(i):
                sched.update_learning_rate(i)
                images, labels, mask = next(it)
                return trainer.train_step_prepared(images, labels, mask)

            for i in range(2):
                step_h2d(i)
            el, rp, _ = timer.median(step_h2d, osteps, reps, first=2)
            out["with_h2d"] = {"value": round(world * args.batch * osteps / el, 2), "unit": "lines/s", "ms_per_step": round(el / osteps * 1e3, 3),
                               "repeats_ms_per_step": [round(e / osteps * 1e3, 3) for e in rp],
                               "h2d_bytes_per_step": int(host[0]["images"].nbytes + host[0]["labels"].nbytes + args.batch * S * 8),
                               "note": "PCIe-inclusive (never `value`): uint8 batch + labels + mask from pinned host memory on a copy stream, "
                                       "double-buffered (common/dataloader.DevicePrefetcher)"}
            del it, host
            # ---- option: head forward + loss on the masked positions only
            model.head_rows = "masked"
            for i in range(3):
                step(i)
            el, rp, _ = timer.median(step, osteps, reps, first=3)
            out["option_masked_head"] = {"head_rows": "masked", "value": round(world * args.batch * osteps / el, 2), "unit": "lines/s",
                                         "ms_per_step": round(el / osteps * 1e3, 3),
                                         "note": "same loss, gradients and update; the head FORWARD too runs on the ~15 % masked positions only "
                                                 "(the reference computes and discards the rest); NOT the headline value"}
            model.head_rows = "all"
            # ---- option: the last layer's row-wise backward over ALL positions (the headline runs it on the masked rows: exact zeros elsewhere)
            from pero_pretraining_amd import functional as Fn
            Fn.ROW_SPARSE_LAST_LAYER = False
            for i in range(3):
                step(i)
            el, rp, _ = timer.median(step, osteps, reps, first=3)
            out["option_dense_last_layer"] = {"value": round(world * args.batch * osteps / el, 2), "unit": "lines/s", "ms_per_step": round(el / osteps * 1e3, 3),
                                              "note": "functional.ROW_SPARSE_LAST_LAYER = False: norm2 / linear2 / linear1 / norm1 / out-projection of the LAST layer "
                                                      "run their backward over all positions, 85 % of which carry an exactly zero gradient (the loss reads the masked "
                                                      "positions); same gradients to 1e-5 (tests/test_gpu_full_size.py); the headline works from the masked rows, as the "
                                                      "head's backward has since round 2"}
            Fn.ROW_SPARSE_LAST_LAYER = not args.dense_last_layer

        if not args.no_roofline:
            per = timed_gemms(lambda n: [step(args.warmup + 7 + k) for k in range(n)], 2)
            traffic, src = pmc_traffic(args.batch)
            out["roofline"] = gemm_roofline(per, 2, elapsed / args.steps, traffic)
            out["roofline"]["traffic_source"] = f"profiles/{src} (csrc_sha256 {csrc_hash()})" if src else None
        del batches
        if not args.no_sweep and not args.masked_head:
            out["batch_sweep"] = batch_sweep(device, model, opt, sched, bf16, rank, timer, world)
        del model, opt, trainer
        torch.cuda.empty_cache()   # the masked model's 48 GB of saved activations
        if not args.no_sweep and rank == 0:
            out["hbm_kernels"] = hbm_kernels(device, args.batch)
            torch.cuda.empty_cache()

    if not args.no_legs:
        reps = min(args.repeats, 3)
        legs = {}
        vp, npairs = (int(v) for v in args.leg_pairs.split(","))
        if world == 1:
            want = lambda name: args.leg is None or name in args.leg.split(",")
            if want("config3_vq_argmin"):
                legs["config3_vq_argmin"] = leg_config3(timer, device)
            if want("config3_masked_step"):
                legs["config3_masked_step"] = leg_config3_step(timer, device, args.batch, 5, reps, rank, world)
                torch.cuda.empty_cache()
            if want("config4_vicreg_step"):
                legs["config4_vicreg_step"] = leg_joint(timer, device, "vicreg", vp, 5, reps, world, rank)
            if want("config5_ntxent_step"):
                legs["config5_ntxent_step"] = leg_joint(timer, device, "ntxent", npairs, 3, reps, world, rank)
        else:
            legs["config4_vicreg_dp"] = leg_joint(timer, device, "vicreg", vp, 5, reps, world, rank)
            legs["config4_vicreg_dp_global_statistics"] = leg_joint(timer, device, "vicreg", vp, 5, reps, world, rank, variant="global")
            legs["config5_ntxent_dp_cross_rank_negatives"] = leg_joint(timer, device, "ntxent", npairs, 3, reps, world, rank, variant="cross")
        out["legs"] = legs

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.legs_only:
        out["cpu_baseline"] = cpu_baseline()
    emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


def plumbing_test(args, world, rank, emit):
    """CPU rehearsal of everything around the GPU work (tests/test_parallel_gloo.py): rank environment, gloo process group, the
    barrier-bracketed max-over-ranks timing, the one JSON line through the saved descriptor and the self-launch relay."""
    dev = torch.device("cpu")
    if "RANK" in os.environ:
        dist.init_process_group("gloo")
    timer = Timer(dev)
    x = torch.zeros(4)

    def step(i):
        y = x + 1.0
        if dist.is_initialized():
            dist.all_reduce(y)
        if rank == world - 1:
            time.sleep(0.002)   # the slowest rank sets the time
        return y

    for i in range(args.warmup):
        step(i)
    el, reps, y = timer.median(step, args.steps, args.repeats, first=args.warmup)
    emit({"metric": "plumbing", "value": round(world * args.steps / el, 2), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
          "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
          "distributed": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "backend": dist.get_backend() if dist.is_initialized() else None,
                          "self_launched": bool(os.environ.get("PERO_BENCH_SELF_LAUNCHED"))},
          "all_reduce_check": float(y[0]), "repeats": len(reps)})
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
