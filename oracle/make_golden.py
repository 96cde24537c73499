#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container: it imports the reference's own modules from /root/reference
(read-only, never copied) on CPU, feeds them seeded inputs and writes small `.npz` fixtures to
tests/golden/.  Only the fixtures (data) travel to the GPU box.

The reference's backbone constructor moves its mask pattern `.to("cuda")`
(models/transformers.py:34), which cannot run on this GPU-less host; construction (only) is
wrapped in a context manager that makes `Tensor.to("cuda")` a no-op.  The reference files are
untouched and no arithmetic is affected.

usage:  python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import contextlib
import os
import sys

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


@contextlib.contextmanager
def cuda_to_is_noop():
    orig = torch.Tensor.to

    def to(self, *args, **kwargs):
        if args and isinstance(args[0], str) and args[0].startswith("cuda"):
            return self
        return orig(self, *args, **kwargs)

    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


def np_sd(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def synthetic_batch(rng, n, w, vocab, pad_from=None):
    images = rng.integers(0, 256, (n, 40, w, 3), dtype=np.uint8)
    labels = rng.integers(0, vocab, (n, w // 8)).astype(np.int64)
    if pad_from is not None:  # trailing padding positions carry label -1 (common/dataloader.py:61)
        for i, p in enumerate(pad_from):
            labels[i, p:] = -1
    return images, labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, REFERENCE_ROOT)
    torch.set_num_threads(8)

    from pero_pretraining.models import transformers as R_tr
    from pero_pretraining.models import autoencoders as R_ae
    from pero_pretraining.masked_pretraining import model as R_mm
    from pero_pretraining.masked_pretraining import trainer as R_mt
    from pero_pretraining.masked_pretraining import batch_operator as R_mb
    from pero_pretraining.joint_embedding_pretraining import losses as R_jl
    from pero_pretraining.joint_embedding_pretraining import model as R_jm
    from pero_pretraining.common import lr_scheduler as R_lr
    from pero_pretraining.common import dataloader as R_dl

    def build_masked(backbone_def, head_def, seed):
        torch.manual_seed(seed)
        with cuda_to_is_noop():
            backbone = R_mm.init_backbone(dict(backbone_def))
        head = R_mm.init_head(dict(head_def))
        return R_mm.MaskedTransformerEncoder(backbone, head)

    # ---- G1: mask tile + positional table samples -----------------------------------------
    with cuda_to_is_noop():
        bb = R_tr.VisionTransformerEncoder(num_blocks=1, model_dim=64, num_heads=4, feedforward_dim=128)
    tile = bb.mask_pattern[0, :, :, :8].numpy().copy()
    assert np.array_equal(bb.mask_pattern[0, :, :, 8:16].numpy(), tile)
    pe = bb.position_model.pe[:, 0, :].numpy()
    np.savez_compressed(os.path.join(out, "g1_tables.npz"), mask_tile=tile,
                        mask_pattern_sum=np.float64(bb.mask_pattern.double().sum().item()),
                        pe64_rows=pe[[0, 1, 7, 100, 4095]], pe64_row_index=np.array([0, 1, 7, 100, 4095]))

    # ---- G2/G3/G4: tiny masked model, eval + train mode, outputs, loss, grads ------------------
    tiny_bb = {"type": "vit", "num_blocks": 2, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128}
    tiny_hd = {"type": "linear", "in_features": 64, "out_features": 96}
    rng = np.random.default_rng(1234)
    model = build_masked(tiny_bb, tiny_hd, seed=0)
    # perturb the identically-initialised clone layers / zero biases so that every tensor matters
    torch.manual_seed(1)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    images, labels = synthetic_batch(rng, 3, 128, 96, pad_from=[16, 12, 16])
    mask = ((rng.random((3, 16)) < 0.3).astype(np.int64)) * (labels >= 0)
    mask[0, 0] = 1
    bop = R_mb.BatchOperator(torch.device("cpu"), 0.15)
    x = bop._prepare_batch_images({"images": images})
    lab_t = bop._prepare_batch_labels({"labels": labels})

    model.eval()
    x_eval = x.clone()
    res = model.forward(x_eval, lab_t, mask.copy())
    model.zero_grad()
    res["loss"].backward()
    fix = {"images": images, "labels": labels, "mask": mask,
           "masked_images_sample": x_eval[:, :, :, :64].detach().numpy(),  # backbone.mask mutates x in place
           "eval_output": res["output"].detach().numpy(), "eval_loss": np.float64(res["loss"].item()),
           "backbone_eval": model.backbone(x.clone(), mask.copy()).detach().numpy()}
    for k, v in np_sd(model).items():
        fix["sd." + k] = v
    for k, p in model.named_parameters():
        fix["grad." + k] = p.grad.detach().numpy().copy()
    # eval without mask (joint-embedding style encode)
    fix["eval_output_nomask"] = model.forward(x.clone())["output"].detach().numpy()

    # train mode: positional offsets come from torch.randint on the global RNG (transformers.py:182);
    # record them by replaying the same RNG state.
    model.train()
    torch.manual_seed(77)
    state = torch.get_rng_state()
    offsets = torch.randint(0, 4096 - 16, (3,))
    torch.set_rng_state(state)
    res = model.forward(x.clone(), lab_t, mask.copy())
    fix["train_offsets"] = offsets.numpy()
    fix["train_output"] = res["output"].detach().numpy()
    fix["train_loss"] = np.float64(res["loss"].item())
    # unmasked-weight variant of the loss (model.py:84-93)
    lossw = R_mm.MaskedCrossEntropyLoss(unmasked_weight=0.25)
    fix["train_loss_unmasked_w025"] = np.float64(lossw(res["output"], lab_t, torch.from_numpy(mask)).item())
    np.savez_compressed(os.path.join(out, "g4_masked_tiny.npz"), **fix)

    # ---- G5: 3-step train_step trajectory through the reference Trainer (Adam + warm-up) -----------
    model = build_masked(tiny_bb, tiny_hd, seed=3)
    sd0 = np_sd(model)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    sched = R_lr.WarmupSchleduler(opt, 2e-3, 2, 1)
    trainer = R_mt.Trainer(bop, model, None, opt, sched, bfloat16=False)
    model.train()
    traj = {"lr": [], "loss": [], "offsets": [], "mask": [], "images": [], "labels": []}
    np.random.seed(5)
    torch.manual_seed(11)
    for it in range(1, 4):
        images, labels = synthetic_batch(rng, 2, 64, 96)
        batch = {"images": images, "labels": labels}
        sched.update_learning_rate(it)
        np_state = np.random.get_state()
        m = bop._create_mask(batch)
        np.random.set_state(np_state)
        st = torch.get_rng_state()
        offs = torch.randint(0, 4096 - 8, (2,))
        torch.set_rng_state(st)
        loss = trainer.train_step(batch)
        traj["lr"].append(sched.current_lr); traj["loss"].append(loss.item()); traj["offsets"].append(offs.numpy())
        traj["mask"].append(m); traj["images"].append(images); traj["labels"].append(labels)
    fix = {k: np.stack(v) for k, v in traj.items()}
    for k, v in sd0.items():
        fix["sd0." + k] = v
    for k, v in np_sd(model).items():
        fix["sd3." + k] = v
    np.savez_compressed(os.path.join(out, "g5_trajectory.npz"), **fix)

    # ---- G4c: config-1 scale model from the seed recipe (weights not stored) -------------------------
    c1_bb = {"type": "vit", "num_blocks": 4, "model_dim": 256, "num_heads": 4, "feedforward_dim": 1024}
    c1_hd = {"in_features": 256, "out_features": 4096}
    model = build_masked(c1_bb, c1_hd, seed=0)
    rng1 = np.random.default_rng(1234)
    images, labels = synthetic_batch(rng1, 8, 512, 4096)
    mask = (rng1.random((8, 64)) < 0.15).astype(np.int64)
    offsets = rng1.integers(0, 4096 - 64, 8)
    model.eval()
    x = bop._prepare_batch_images({"images": images})
    res = model.forward(x.clone(), torch.from_numpy(labels), mask.copy())
    model.zero_grad(); res["loss"].backward()
    sd = model.state_dict()
    samp = np.random.default_rng(0).integers(0, 8 * 64 * 4096, 4096)
    fix = {"eval_loss": np.float64(res["loss"].item()),
           "logit_sample_index": samp, "logit_samples": res["output"].detach().numpy().reshape(-1)[samp],
           "param_checksums": np.array([float(v.double().sum()) for v in sd.values()]),
           "param_abs_checksums": np.array([float(v.double().abs().sum()) for v in sd.values()]),
           "param_names": np.array(list(sd.keys())),
           "grad_norms": np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()]),
           "offsets": offsets}
    # train mode with explicit offsets: replay through torch.randint by monkeypatch-free trick = set pe shift
    model.train()
    torch.manual_seed(99)
    st = torch.get_rng_state(); offs = torch.randint(0, 4096 - 64, (8,)); torch.set_rng_state(st)
    res = model.forward(x.clone(), torch.from_numpy(labels), mask.copy())
    fix["train_offsets"] = offs.numpy(); fix["train_loss"] = np.float64(res["loss"].item())
    np.savez_compressed(os.path.join(out, "g4c_config1.npz"), **fix)

    # ---- G6/G7: quantizers -------------------------------------------------------------------------
    fix = {}
    for name, (K, D, M) in {"small": (64, 32, 200), "cb8192": (8192, 512, 256)}.items():
        torch.manual_seed(5)
        vq = R_ae.VectorQuantizer(K, D, 0.25, 0.99)
        vq.eval()
        g = np.random.default_rng(7)
        feats = g.standard_normal((2, D, 1, M // 2)).astype(np.float32)
        with torch.no_grad():
            q, idx = vq(torch.from_numpy(feats))
            flat = torch.from_numpy(feats).permute(0, 2, 3, 1).reshape(-1, D)
            w = vq.embedding.weight
            dist = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1) - 2 * torch.matmul(flat, w.t()))
            two = torch.topk(dist, 2, dim=1, largest=False).values
            cd = torch.cdist(flat, w)
            km = torch.argmin(cd, dim=1)
        fix[f"{name}.codebook_seed"] = np.int64(5)
        if K <= 64:
            fix[f"{name}.codebook"] = w.detach().numpy().copy()
        else:  # large codebook: store the seed recipe + checksum (normal_() on the embedding weight)
            fix[f"{name}.codebook_checksum"] = np.float64(w.double().sum().item())
            fix[f"{name}.codebook_head"] = w[:4, :8].detach().numpy().copy()
        fix[f"{name}.features"] = feats
        fix[f"{name}.indices"] = idx.numpy()
        fix[f"{name}.quantized_sample"] = q[:, :8, :, :8].numpy()
        fix[f"{name}.best"] = two[:, 0].numpy(); fix[f"{name}.second"] = two[:, 1].numpy()
        fix[f"{name}.kmeans_indices"] = km.numpy()
    np.savez_compressed(os.path.join(out, "g6_quantizers.npz"), **fix)

    # ---- G8: VICReg with the real three-valued masks of BatchCreator ---------------------------------
    bc = R_dl.BatchCreator()
    g = np.random.default_rng(11)
    lines = [g.integers(0, 256, (40, w, 3), dtype=np.uint8) for w in (480, 512, 400, 512)]
    data = [{"image": im, "image2": im, "labels": None, "image_id": str(i)} for i, im in enumerate(lines)]
    for seed in range(3, 50):  # first numpy seed whose paddings give equally many invariance rows in both views
        np.random.seed(seed)
        st = bc.stack_images(data)
        masks = (st[2], st[3], st[9], st[10])
        if int((masks[2] == 1).sum()) == int((masks[3] == 1).sum()) and (masks[2] == 2).any():
            break
    stack_seed, shifts = seed, np.array(st[8])
    masks = tuple(np.ascontiguousarray(m) for m in masks)
    n, S = masks[0].shape
    D = 48
    xv = torch.from_numpy(g.standard_normal((n, S, D)).astype(np.float32)).requires_grad_(True)
    yv = torch.from_numpy((g.standard_normal((n, S, D)) * 0.7 + 0.1).astype(np.float32)).requires_grad_(True)
    tm = tuple(torch.from_numpy(m) for m in masks)
    res = R_jl.VICRegLoss()(xv, yv, *tm)
    res["loss"].backward()
    fix = {"x": xv.detach().numpy(), "y": yv.detach().numpy(), "image_masks1": masks[0], "image_masks2": masks[1],
           "shift_masks1": masks[2], "shift_masks2": masks[3], "grad_x": xv.grad.numpy(), "grad_y": yv.grad.numpy(),
           "stack_seed": np.int64(stack_seed), "shifts": shifts}
    for k, v in res.items():
        fix[k] = np.float64(v.item())
    res2 = R_jl.VICRegLoss(variance_weight=25.0, invariance_weight=25.0, covariance_weight=1.0)(xv, yv, *tm)
    fix["loss_w25_25_1"] = np.float64(res2["loss"].item())
    np.savez_compressed(os.path.join(out, "g8_vicreg.npz"), **fix)

    # ---- G9: NT-Xent (all-ones masks: the only inputs the reference accepts) --------------------------
    n, S, D = 3, 24, 40
    xv = torch.from_numpy(g.standard_normal((n, S, D)).astype(np.float32)).requires_grad_(True)
    yv = torch.from_numpy((xv.detach().numpy() + 0.5 * g.standard_normal((n, S, D))).astype(np.float32)).requires_grad_(True)
    ones = torch.ones((n, S), dtype=torch.uint8)
    res = R_jl.NTXentLoss()(xv, yv, ones, ones, ones, ones)
    res["loss"].backward()
    raised = False
    try:
        bad = ones.clone(); bad[:, :3] = 0
        R_jl.NTXentLoss()(xv, yv, ones, ones, bad, bad.flip(1))
    except IndexError:
        raised = True
    np.savez_compressed(os.path.join(out, "g9_ntxent.npz"), x=xv.detach().numpy(), y=yv.detach().numpy(),
                        loss=np.float64(res["loss"].item()), grad_x=xv.grad.numpy(), grad_y=yv.grad.numpy(),
                        nontrivial_shift_mask_raises_indexerror=np.bool_(raised))

    # ---- G10: learning-rate schedule -------------------------------------------------------------------
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = R_lr.WarmupSchleduler(opt, 2e-4, 10000, 1)
    its = [0, 1, 5000, 10000, 10001]
    lrs = []
    for it in its:
        sched.update_learning_rate(it); lrs.append(opt.param_groups[0]["lr"])
    sched2 = R_lr.WarmupSchleduler(opt, 1e-3, 100, 2)
    lrs2 = []
    for it in [0, 10, 50, 100, 101]:
        sched2.update_learning_rate(it); lrs2.append(opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(out, "g10_lr.npz"), iterations=np.array(its), lr=np.array(lrs),
                        iterations2=np.array([0, 10, 50, 100, 101]), lr2=np.array(lrs2))

    # ---- G11: joint-embedding model (default L6 d512 backbone is too big to store: tiny via direct ctor)
    torch.manual_seed(21)
    with cuda_to_is_noop():
        jb = R_tr.VisionTransformerEncoder(num_blocks=2, model_dim=64, num_heads=4, feedforward_dim=128)
    jh = R_jm.LinearHead(in_features=64, out_features=80)
    jmodel = R_jm.JointEmbeddingTransformerEncoder(jb, jh, R_jl.VICRegLoss())
    torch.manual_seed(2)
    with torch.no_grad():
        for p in jmodel.parameters():
            p.add_(0.05 * torch.randn_like(p))
    jmodel.eval()
    W = masks[0].shape[1] * 8
    im1 = g.integers(0, 256, (4, 40, W, 3), dtype=np.uint8)
    im2 = g.integers(0, 256, (4, 40, W, 3), dtype=np.uint8)
    x1 = torch.from_numpy(im1).float().permute(0, 3, 1, 2) / 255.0
    x2 = torch.from_numpy(im2).float().permute(0, 3, 1, 2) / 255.0
    res = jmodel.forward(x1, x2, *tm)
    jmodel.zero_grad(); res["loss"].backward()
    fix = {"images1": im1, "images2": im2, "image_masks1": masks[0], "image_masks2": masks[1],
           "shift_masks1": masks[2], "shift_masks2": masks[3], "output1": res["output1"].numpy(),
           "output2": res["output2"].numpy()}
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        fix[k] = np.float64(res[k].item())
    for k, v in np_sd(jmodel).items():
        fix["sd." + k] = v
    for k, p in jmodel.named_parameters():
        fix["gradnorm." + k] = np.float64(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(out, "g11_joint_tiny.npz"), **fix)

    sizes = {f: os.path.getsize(os.path.join(out, f)) for f in sorted(os.listdir(out))}
    print("wrote:", sizes)


if __name__ == "__main__":
    main()
