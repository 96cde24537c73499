#!/usr/bin/env python3
"""Round-2 golden vectors, produced by the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (imports /root/reference read-only on CPU, same `.to("cuda")` construction shim as
oracle/make_golden.py); writes small fixtures under tests/golden/:

  g17_vq_large.npz          `VectorQuantizer(8192, 512)` in eval mode on 4096 feature rows: indices, best / second-best
                            distance (the margin that identifies near-ties), and the k-means labelling
                            `torch.cdist(...).argmin` of scripts/produce_kmeans_labels.py:72-76 against 4096 centroids with its
                            own margins.  The 8 MB of features / 8 + 16 MB of code vectors are NOT stored: they come from
                            seed recipes (numpy Generator / torch.manual_seed) and are pinned by checksums + head samples.
  g18_joint_trajectory.npz  three `Trainer.train_step` calls of the reference's OWN joint-embedding Trainer
                            (joint_embedding_pretraining/trainer.py:46-61) with its BatchOperator, a VICReg loss, Adam and the
                            warm-up schedule on a tiny ViT + LinearHead, batches from the reference BatchCreator (three-valued
                            shift masks): losses, loss parts, positional offsets of both encodes, initial / final weights.
  g19_mlp_head.npz          the reference's `MLPHead` (joint_embedding_pretraining/model.py:79-115) at small dims: weights,
                            input, output, gradients.

usage:  python oracle/make_golden_r2.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REFERENCE_ROOT, cuda_to_is_noop, np_sd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    sys.path.insert(0, REFERENCE_ROOT)
    torch.set_num_threads(8)
    from pero_pretraining.models import autoencoders as R_ae
    from pero_pretraining.models import transformers as R_tr
    from pero_pretraining.joint_embedding_pretraining import model as R_jm
    from pero_pretraining.joint_embedding_pretraining import losses as R_jl
    from pero_pretraining.joint_embedding_pretraining import trainer as R_jt
    from pero_pretraining.joint_embedding_pretraining import batch_operator as R_jb
    from pero_pretraining.common import dataloader as R_dl
    from pero_pretraining.common import lr_scheduler as R_lr

    # ---- G17: the config-3 quantizer on 4096 rows, and the k-means labelling at its default K = 4096 --------------
    K, D, M = 8192, 512, 4096
    torch.manual_seed(5)
    vq = R_ae.VectorQuantizer(K, D, 0.25, 0.99)   # embedding.weight.data.normal_() (models/autoencoders.py:180)
    vq.eval()
    g = np.random.default_rng(1717)
    feats = g.standard_normal((4, D, 1, M // 4)).astype(np.float32)   # (N, D_e, 1, T) as VQVAE.quantize hands it over
    with torch.no_grad():
        q, idx = vq(torch.from_numpy(feats))
        flat = torch.from_numpy(feats).permute(0, 2, 3, 1).reshape(-1, D)
        w = vq.embedding.weight
        dist = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1) - 2 * torch.matmul(flat, w.t()))
        two = torch.topk(dist, 2, dim=1, largest=False).values
        assert torch.equal(torch.argmin(dist, dim=1), idx)
        # scripts/produce_kmeans_labels.py:72-76: centroids (1, K, F), features (M, F) -> cdist -> squeeze -> argmin(dim=1)
        torch.manual_seed(6)
        cent = torch.randn(4096, D)
        cd = torch.cdist(flat, cent.unsqueeze(0)).squeeze()
        km = torch.argmin(cd, dim=1)
        km2 = torch.topk(cd, 2, dim=1, largest=False).values
    fix = {"K": np.int64(K), "D": np.int64(D), "rows": np.int64(M),
           "feature_seed": np.int64(1717), "feature_shape": np.array(feats.shape), "feature_checksum": np.float64(feats.astype(np.float64).sum()),
           "feature_head": feats[0, :8, 0, :8].copy(),
           "codebook_seed": np.int64(5), "codebook_checksum": np.float64(w.double().sum().item()), "codebook_head": w[:4, :8].detach().numpy().copy(),
           "indices": idx.numpy(), "best": two[:, 0].numpy(), "second": two[:, 1].numpy(),
           "quantized_sample": q[:, :8, :, :8].numpy(),
           "centroid_seed": np.int64(6), "centroid_checksum": np.float64(cent.double().sum().item()), "centroid_head": cent[:4, :8].numpy().copy(),
           "kmeans_indices": km.numpy(), "kmeans_best": km2[:, 0].numpy(), "kmeans_second": km2[:, 1].numpy()}
    np.savez_compressed(os.path.join(out, "g17_vq_large.npz"), **fix)
    rel = (two[:, 1] - two[:, 0]) / two[:, 0].abs()
    print("g17: smallest relative margins", np.sort(rel.numpy())[:6], " k-means:", np.sort(((km2[:, 1] - km2[:, 0]) / km2[:, 0]).numpy())[:4])

    # ---- G18: the reference's joint Trainer.train_step, three steps ------------------------------------------------
    torch.manual_seed(31)
    with cuda_to_is_noop():
        jb = R_tr.VisionTransformerEncoder(num_blocks=2, model_dim=64, num_heads=4, feedforward_dim=128)
    jh = R_jm.LinearHead(in_features=64, out_features=80)
    jmodel = R_jm.JointEmbeddingTransformerEncoder(jb, jh, R_jl.VICRegLoss())
    torch.manual_seed(32)
    with torch.no_grad():
        for p in jmodel.parameters():
            p.add_(0.05 * torch.randn_like(p))
    sd0 = np_sd(jmodel)
    opt = torch.optim.Adam(jmodel.parameters(), lr=1e-3)
    sched = R_lr.WarmupSchleduler(opt, 1e-3, 2, 1)
    bop = R_jb.BatchOperator(torch.device("cpu"))
    trainer = R_jt.Trainer(bop, jmodel, None, opt, sched, bfloat16=False)
    jmodel.train()
    bc = R_dl.BatchCreator()
    rng = np.random.default_rng(77)
    traj = {k: [] for k in ("lr", "loss", "loss.variance", "loss.invariance", "loss.covariance", "offsets1", "offsets2")}
    batches = []
    torch.manual_seed(33)
    step_seeds = []
    for it in range(1, 4):
        lines = [rng.integers(0, 256, (40, wd, 3), dtype=np.uint8) for wd in (256, 224, 248, 256)]
        data = [{"image": im, "image2": im, "labels": None, "image_id": str(i)} for i, im in enumerate(lines)]
        for seed in range(100 * it, 100 * it + 80):   # a numpy seed whose paddings give equally many invariance rows in both views
            np.random.seed(seed)
            batch = bc.create_batch(data)
            if int((batch["shift_masks"] == 1).sum()) == int((batch["shift_masks2"] == 1).sum()):
                break
        else:
            raise RuntimeError("no usable padding seed")
        step_seeds.append(seed)
        sched.update_learning_rate(it)
        n, S = batch["image_masks"].shape
        st = torch.get_rng_state()
        o1 = torch.randint(0, 4096 - S, (n,)); o2 = torch.randint(0, 4096 - S, (n,))   # the two encodes' draws (transformers.py:182)
        torch.set_rng_state(st)
        # the reference returns the loss tensor only; the parts come from a forward of the same batch BEFORE the step
        # (same RNG state -> same offsets; no parameter change)
        with torch.no_grad():
            parts = jmodel.forward(*bop.prepare_batch(batch))
        torch.set_rng_state(st)
        loss = trainer.train_step(batch)
        assert abs(float(loss) - float(parts["loss"])) < 1e-6 * abs(float(loss))
        traj["lr"].append(sched.current_lr); traj["loss"].append(loss.item())
        for k in ("loss.variance", "loss.invariance", "loss.covariance"):
            traj[k].append(parts[k].item())
        traj["offsets1"].append(o1.numpy()); traj["offsets2"].append(o2.numpy())
        batches.append(batch)
    fix = {k: np.stack(v) for k, v in traj.items()}
    fix["padding_seeds"] = np.array(step_seeds)
    for i, b in enumerate(batches):
        for k in ("images", "images2", "image_masks", "image_masks2", "shift_masks", "shift_masks2"):
            fix[f"b{i}.{k}"] = np.ascontiguousarray(b[k])
    for k, v in sd0.items():
        fix["sd0." + k] = v
    for k, v in np_sd(jmodel).items():
        fix["sd3." + k] = v
    np.savez_compressed(os.path.join(out, "g18_joint_trajectory.npz"), **fix)
    print("g18: losses", traj["loss"], "shapes", batches[0]["images"].shape)

    # ---- G19: MLPHead ----------------------------------------------------------------------------------------------
    torch.manual_seed(41)
    head = R_jm.MLPHead(in_dim=48, hidden_dim=96, num_layers=3)
    x = torch.randn(3, 20, 48, requires_grad=True)
    y = head(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    fix = {"x": x.detach().numpy(), "y": y.detach().numpy(), "gy": gy.numpy(), "grad_x": x.grad.numpy()}
    for k, v in np_sd(head).items():
        fix["sd." + k] = v
    for k, p in head.named_parameters():
        fix["grad." + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(out, "g19_mlp_head.npz"), **fix)

    sizes = {f: os.path.getsize(os.path.join(out, f)) for f in ("g17_vq_large.npz", "g18_joint_trajectory.npz", "g19_mlp_head.npz")}
    print("wrote:", sizes)


if __name__ == "__main__":
    main()
