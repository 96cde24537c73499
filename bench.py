#!/usr/bin/env python3
"""Throughput of the masked-pretraining step (BASELINE.json configs[1]): 12-layer d=512 ViT over
40x2048 synthetic uint8 lines, V=4096, bf16 MFMA, one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A step = zero_grad -> front end (u8 -> patches, mask tile) -> 12 encoder layers -> head -> masked CE ->
backward of all of it -> [gradient all-reduce] -> fused Adam.  Prints ONE JSON line (rank 0).

  value               whole-job lines/s with the batch resident in HBM when the timed region starts: the MEDIAN of three
                      repeats of exactly --steps steps, each bracketed by barrier + synchronize (max over ranks)
  with_prepare_batch  the same step INCLUDING BatchOperator.prepare_batch: host mask draw + the uint8 batch from pinned host
                      memory over PCIe on a copy stream, double-buffered against the previous step (never `value`)
  roofline            the bf16 tile GEMM: algorithmic flops / HIP-event duration of every launch of two extra steps
  legs                driver-visible numbers of the other single-GPU configs: config 3 (codebook argmin, its own roofline vs the
                      f32 MFMA peak), configs 4 / 5 (VICReg / NT-Xent joint steps, per-GPU share), medians of three repeats
  cpu_baseline        the CPU oracle (oracle/pero_oracle.py, a "port") on the host cores, a bounded sample of the same workload
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
F32_MFMA_PEAK_TFLOPS = 157.3    # same table, "Peak FP32 (matrix)"
REPEATS = 3

CFG = dict(num_blocks=12, model_dim=512, num_heads=4, feedforward_dim=2048, vocab=4096, width=2048, height=40,
           patch=8, channels=3)


def flops_per_line(c=CFG):
    """forward F = patch + L*(qkv + attn + out + ffn) + head; step = 3F."""
    S, d, L, ff, V = c["width"] // c["patch"], c["model_dim"], c["num_blocks"], c["feedforward_dim"], c["vocab"]
    kp = c["channels"] * c["height"] * c["patch"]
    F = 2 * S * kp * d + L * (2 * S * d * 3 * d + 4 * S * S * d + 2 * S * d * d + 4 * S * d * ff) + 2 * S * d * V
    return 3 * F


def build(device, bf16):
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining import model as M
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    torch.manual_seed(0)
    bb = M.init_backbone({"type": "vit", "num_blocks": CFG["num_blocks"], "model_dim": CFG["model_dim"],
                          "num_heads": CFG["num_heads"], "feedforward_dim": CFG["feedforward_dim"]})
    hd = M.init_head({"in_features": CFG["model_dim"], "out_features": CFG["vocab"]})
    model = M.MaskedTransformerEncoder(bb, hd).to(device).train()
    opt = FusedAdam(model.parameters(), lr=2e-4)
    sched = WarmupSchleduler(opt, 2e-4, 10000, 1)
    trainer = Trainer(None, model, None, opt, sched, bfloat16=bf16)
    return model, opt, sched, trainer


def synthetic(rank, B, device, nbatches=2):
    rng = np.random.default_rng(1234 + rank)
    W, S = CFG["width"], CFG["width"] // CFG["patch"]
    out = []
    for _ in range(nbatches):
        images = torch.from_numpy(rng.integers(0, 256, (B, CFG["height"], W, CFG["channels"]), dtype=np.uint8)).to(device)
        labels = torch.from_numpy(rng.integers(0, CFG["vocab"], (B, S)).astype(np.int64)).to(device)
        mask = torch.from_numpy((rng.random((B, S)) < 0.15).astype(np.int64)).to(device)
        out.append((images, labels, mask))
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


def pmc_traffic(batch):
    """HBM bytes per GEMM launch from the committed rocprofv3 --pmc summary of this command (profiles/), if it was
    collected for this batch size; None otherwise (bench.py cannot run the profiler on itself)."""
    for name in ("r02_pmc_gemm_traffic.json", "r01_pmc_gemm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if int(d.get("lines_per_gpu", -1)) == int(batch):
                return d.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
    return None


class Timer:
    """K steps bracketed by barrier + synchronize on both sides, max over ranks; median over repeats."""

    def __init__(self, device):
        self.device = device

    def fence(self):
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
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

    def median(self, step, steps, repeats=REPEATS, first=0):
        els, out = [], None
        for r in range(repeats):
            el, out = self.run(step, steps, first + r * steps)
            els.append(el)
        return statistics.median(els), els, out


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
    return {"workload": "VQ codebook argmin (BASELINE.json configs[2]): 32768 rows (128 lines x 256) x 8192 codes x 512, f32 exact",
            "ms": round(t * 1e3, 3), "lines_per_s": round(M / 256 / t, 1), "rows_per_s": round(M / t, 1),
            "roofline": {"bound": "mfma", "kernel": "vq_argmin_fast_k (v_mfma_f32_32x32x2_f32, running argmin in registers)",
                         "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None}}


def leg_joint(timer, device, kind, pairs, steps):
    """BASELINE.json configs[3] / [4], one GPU's share: joint-embedding step (two views through the 12-layer backbone as one 2N
    batch + linear head 4096 + VICReg / NT-Xent + backward + fused Adam) on `pairs` line pairs of 40x2048."""
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.joint_embedding_pretraining import model as J
    from pero_pretraining_amd.joint_embedding_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss, VICRegLoss
    from pero_pretraining_amd.joint_embedding_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    torch.manual_seed(0)
    bb = J.init_backbone({"num_blocks": 12, "model_dim": 512, "num_heads": 4, "feedforward_dim": 2048})
    hd = J.init_head({"type": "linear", "in_features": 512, "out_features": 4096})
    model = J.JointEmbeddingTransformerEncoder(bb, hd, VICRegLoss() if kind == "vicreg" else NTXentLoss()).to(device).train()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    tr = Trainer(BatchOperator(device), model, None, opt, WarmupSchleduler(opt, 1e-4, 100, 1), bfloat16=True)
    rng = np.random.default_rng(5)
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
    med, els, loss = timer.median(lambda i: tr.train_step_prepared(*prepared), steps)
    del model, opt, tr
    return {"workload": f"{'VICReg' if kind == 'vicreg' else 'NT-Xent'} joint-embedding step, 12-layer d=512 ViT + linear head 4096, {pairs} line pairs "
                        f"of 40x2048 per GPU, bf16 (BASELINE.json configs[{3 if kind == 'vicreg' else 4}], one GPU's share)",
            "ms_per_step": round(med / steps * 1e3, 3), "line_pairs_per_s": round(pairs * steps / med, 1),
            "repeats_ms_per_step": [round(e / steps * 1e3, 3) for e in els], "loss": round(float(loss), 5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PERO_BENCH_BATCH", 1024)), help="lines per GPU")
    ap.add_argument("--no-side-stream", action="store_true", help="weight gradients on the main stream (clean per-kernel profiles)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the config 3 / 4 / 5 legs")
    ap.add_argument("--dp-no-overlap", action="store_true", help="data parallel: reduce all gradients after the backward pass (A/B of the bucket hooks)")
    ap.add_argument("--dp-layers-per-bucket", type=int, default=2)
    ap.add_argument("--no-options", action="store_true", help="skip the extra legs (masked head, prepare_batch pipeline)")
    ap.add_argument("--masked-head", action="store_true",
                    help="OPTION, not the headline: head + loss on the masked positions only (model.head_rows = 'masked'); "
                         "the default evaluates the head on every position like the reference")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # Under torch.distributed.run RCCL prints a version banner on the C-level stdout at communicator creation: everything but the ONE
    # JSON line goes to stderr (file descriptor 1 is pointed at stderr; the line is written to the saved descriptor at the end)
    real_stdout = None
    if "RANK" in os.environ:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: same code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend()

    from pero_pretraining_amd import functional as F
    from pero_pretraining_amd import ops
    from pero_pretraining_amd.parallel import DataParallel
    if args.no_side_stream:
        F.SIDE_STREAM_DW = False
    bf16 = args.dtype == "bf16"
    model, opt, sched, trainer = build(device, bf16)
    if args.masked_head:
        model.head_rows = "masked"
    if dist.is_initialized():
        trainer.data_parallel = DataParallel(model, opt, overlap=not args.dp_no_overlap, layers_per_bucket=args.dp_layers_per_bucket)
    batches = synthetic(rank, args.batch, device)
    timer = Timer(device)

    # --masked-head: the positions are listed once per resident batch (in training the mask is drawn on the host, so the
    # list costs no device sync there either)
    row_lists = [torch.nonzero(b[2].reshape(-1) == 1).reshape(-1) for b in batches]

    def step(i):
        sched.update_learning_rate(i)
        images, labels, mask = batches[i % len(batches)]
        rows = row_lists[i % len(batches)] if model.head_rows == "masked" else None
        return trainer.train_step_prepared(images, labels, mask, rows=rows)

    # ---- headline: batch resident in HBM
    for i in range(args.warmup):
        step(i)
    elapsed, repeats, loss = timer.median(step, args.steps, first=args.warmup)
    final_loss = float(loss)
    lines_per_s = world * args.batch * args.steps / elapsed
    step_flops = flops_per_line()
    head_rows = model.head_rows

    # ---- the same step including BatchOperator.prepare_batch: host mask draw (numpy, the reference's call) + u8 batch over PCIe
    # from pinned host memory on a copy stream, double-buffered against the previous step
    with_prepare = None
    if not args.no_options and not args.masked_head:
        from pero_pretraining_amd.common.dataloader import DevicePrefetcher
        from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
        bop = BatchOperator(device, 0.15)
        rng = np.random.default_rng(99 + rank)
        S = CFG["width"] // CFG["patch"]
        host = [{"images": rng.integers(0, 256, (args.batch, CFG["height"], CFG["width"], CFG["channels"]), dtype=np.uint8),
                 "labels": rng.integers(0, CFG["vocab"], (args.batch, S)).astype(np.int64)} for _ in range(2)]
        n_total = 2 + REPEATS * args.steps
        pf = DevicePrefetcher((host[i % 2] for i in range(n_total + 1)), bop, device)
        it = iter(pf)

        def step_h2d(i):
            sched.update_learning_rate(i)
            images, labels, mask = next(it)
            return trainer.train_step_prepared(images, labels, mask)

        for i in range(2):
            step_h2d(i)
        el, reps, _ = timer.median(step_h2d, args.steps, first=2)
        with_prepare = {"value": round(world * args.batch * args.steps / el, 2), "unit": "lines/s", "ms_per_step": round(el / args.steps * 1e3, 3),
                        "repeats_ms_per_step": [round(e / args.steps * 1e3, 3) for e in reps],
                        "h2d_bytes_per_step": int(host[0]["images"].nbytes + host[0]["labels"].nbytes + args.batch * S * 8),
                        "note": "prepare_batch inside the timed step: numpy mask draw on the host (masked_pretraining/batch_operator.py:27-32), "
                                "uint8 batch + labels + mask from pinned host memory on a copy stream, double-buffered (common/dataloader.DevicePrefetcher)"}
        del pf, it, host

    # ---- reported beside the headline, never as `value`: the same step with the head and the loss on the masked positions only
    option = None
    if not args.masked_head and not args.no_options:
        model.head_rows = "masked"
        for i in range(3):
            step(i)
        el, reps, _ = timer.median(step, args.steps, first=3)
        option = {"head_rows": "masked", "value": round(world * args.batch * args.steps / el, 2), "unit": "lines/s",
                  "ms_per_step": round(el / args.steps * 1e3, 3),
                  "note": "same loss, gradients and update; the head (d -> V) and the cross entropy run on the ~15 % masked "
                          "positions instead of all (the reference computes and discards the rest); NOT the headline value"}
        model.head_rows = "all"

    roofline = None
    if not args.no_roofline:
        # every GEMM launch of two extra steps bracketed by HIP events on the stream it is launched on; the weight
        # gradients are put back on the main stream for this leg so that no two kernels share the chip while timed
        side_was = F.SIDE_STREAM_DW
        F.SIDE_STREAM_DW = False
        step(args.warmup + args.steps)
        torch.cuda.synchronize()
        ops.gemm_timeline = []
        for i in range(2):
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        tl, ops.gemm_timeline = ops.gemm_timeline, None
        F.SIDE_STREAM_DW = side_was
        per = {}
        for e0, e1, fl, tag in tl:
            d = per.setdefault(tag, [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += fl
            d[2] += 1
        fast = {k: v for k, v in per.items() if k.startswith("gemm_bf16_tile")}
        tsum = sum(v[0] for v in fast.values()) or 1e-30
        fsum = sum(v[1] for v in fast.values())
        nl = sum(v[2] for v in fast.values()) or 1
        ach = fsum / tsum / 1e12
        roofline = {"bound": "mfma",
                    "kernel": "bf16 MFMA tile GEMM gemm_bf16_e256 (eight-phase persistent 256x256x64: forward, input gradients with fused "
                              "epilogues, split-K weight gradients)",
                    "achieved": round(ach, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(args.batch), "launches_per_step": nl // 2, "avg_launch_us": round(tsum / nl * 1e6, 2),
                    "gflop_per_launch": round(fsum / nl / 1e9, 3),
                    "gemm_time_share_of_step": round((tsum / 2) / (elapsed / args.steps), 3),
                    "by_layout": {k: {"tflops": round(v[1] / v[0] / 1e12, 1), "ms_per_step": round(v[0] / 2 * 1e3, 3),
                                      "launches_per_step": v[2] // 2} for k, v in sorted(per.items())}}

    legs = None
    if not args.no_legs and world == 1:
        # free the masked model's 48 GB of saved activations first
        del batches, row_lists
        torch.cuda.empty_cache()
        legs = {"config3_vq_argmin": leg_config3(timer, device),
                "config4_vicreg_step": leg_joint(timer, device, "vicreg", 128, 5),
                "config5_ntxent_step": leg_joint(timer, device, "ntxent", 128, 5)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "text-line images/sec (masked-ViT step)", "value": round(lines_per_s, 2), "unit": "lines/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "masked pretraining step, 12-layer d=512 h=4 ff=2048 ViT, V=4096, 40x2048 u8 lines "
                                   "(BASELINE.json configs[1])",
                       "lines_per_gpu": args.batch, "global_batch": args.batch * world, "seq_len": CFG["width"] // CFG["patch"],
                       "parallelism": f"dp{world}", "optimizer": "fused Adam (f32 master weights)",
                       "weight_gradients_on_side_stream": bool(F.SIDE_STREAM_DW), "head_rows": head_rows,
                       "gflop_per_line_step": round(step_flops / 1e9, 3)},
            "protocol": {"repeats": REPEATS, "statistic": "median", "repeats_ms_per_step": [round(e / args.steps * 1e3, 3) for e in repeats],
                         "inputs": "resident in HBM"},
            "distributed": {"world_size": dist.get_world_size() if dist.is_initialized() else 1, "backend": backend,
                            "gradient_all_reduce": "bucketed, overlapped with backward" if dist.is_initialized() and not args.dp_no_overlap else
                                                   ("after backward" if dist.is_initialized() else None)},
            "step_mfma_frac": round(lines_per_s / world * step_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
            "final_loss": round(final_loss, 5),
            "roofline": roofline, "cpu_baseline": cpu, "with_prepare_batch": with_prepare, "option_masked_head": option, "legs": legs,
        }
        if real_stdout is None:
            print(json.dumps(out))
        else:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
