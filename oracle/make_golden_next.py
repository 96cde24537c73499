#!/usr/bin/env python3
"""Golden vectors for the SURVEY.md section 8(f) "next" rows, produced by the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (imports /root/reference read-only on CPU, same `.to("cuda")` construction shim as
oracle/make_golden.py); writes small fixtures under tests/golden/:

  g12_tester.npz                 reference masked Tester.test() on three synthetic batches (loss, errors_1/3/10, masks)
  g13_reference_checkpoint.pth   `MaskedTransformerEncoder.save()` of a tiny reference model (plain tensor state_dict;
                                 load it with torch.load(weights_only=True) only)
  g13_checkpoint.npz             inputs + eval output of that model, state_dict key / shape listing
  g14_labels.txt / g14_labels.npz  label text file written by the reference's own `save_labels` (scripts/common.py:51-54;
                                 the module cannot be imported - it needs lmdb / cv2 - so that ONE function is compiled
                                 from the module's syntax tree and run; nothing is copied)
  g16_vq_ema.npz                 reference VectorQuantizer in TRAINING mode, two steps: indices, quantized, straight-through
                                 gradient, EMA cluster sizes / ema_w / codebook after each step
  g15_batch_creator.npz          `BatchCreator.stack_images` / `create_batch` (common/dataloader.py:32-155) outputs for
                                 seeded ragged lines: padded images, image masks, shifts, three-valued shift masks,
                                 labels; padded and crop mode

usage:  python oracle/make_golden_next.py [--out tests/golden]
"""
import argparse
import ast
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REFERENCE_ROOT, cuda_to_is_noop, np_sd, synthetic_batch  # noqa: E402


def reference_function(relpath, name, extra_globals=None):
    """Compile ONE top-level function of a reference module that cannot be imported as a whole."""
    path = os.path.join(REFERENCE_ROOT, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1, (relpath, name)
    mod = ast.Module(body=fn, type_ignores=[])
    g = dict(extra_globals or {})
    exec(compile(mod, path, "exec"), g)
    return g[name]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    sys.path.insert(0, REFERENCE_ROOT)
    torch.set_num_threads(8)
    from pero_pretraining.masked_pretraining import model as R_mm
    from pero_pretraining.masked_pretraining import batch_operator as R_mb
    from pero_pretraining.masked_pretraining import tester as R_mtest
    from pero_pretraining.common import dataloader as R_dl
    from pero_pretraining.common import helpers as R_h

    def build_masked(backbone_def, head_def, seed, perturb_seed):
        torch.manual_seed(seed)
        with cuda_to_is_noop():
            backbone = R_mm.init_backbone(dict(backbone_def))
        head = R_mm.init_head(dict(head_def))
        model = R_mm.MaskedTransformerEncoder(backbone, head)
        torch.manual_seed(perturb_seed)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.05 * torch.randn_like(p))
        return model

    # ---- G12: Tester ------------------------------------------------------------------------------
    bb = {"type": "vit", "num_blocks": 2, "model_dim": 32, "num_heads": 4, "feedforward_dim": 64}
    hd = {"type": "linear", "in_features": 32, "out_features": 24}
    model = build_masked(bb, hd, seed=4, perturb_seed=6)
    rng = np.random.default_rng(99)
    batches = []
    for n, w, pad in ((3, 128, [16, 10, 16]), (2, 128, None), (4, 64, [8, 8, 5, 8])):
        images, labels = synthetic_batch(rng, n, w, 24, pad_from=pad)
        batches.append({"images": images, "labels": labels})
    bop = R_mb.BatchOperator(torch.device("cpu"), 0.4)
    np.random.seed(17)
    tester = R_mtest.Tester(bop, model, batches, measured_errors=(1, 3, 10))
    res = tester.test()
    fix = {"loss": np.float64(float(res["loss"])), "errors_1": np.float64(res["errors_1"]),
           "errors_3": np.float64(res["errors_3"]), "errors_10": np.float64(res["errors_10"]),
           "numpy_seed": np.int64(17), "masking_prob": np.float64(0.4), "n_batches": np.int64(len(batches))}
    # max_lines variant (tester.py:40: stops after the batch that EXCEEDS max_lines)
    np.random.seed(17)
    res2 = R_mtest.Tester(bop, model, [dict(images=b["images"], labels=b["labels"]) for b in batches], max_lines=3,
                          measured_errors=(1, 5)).test()
    fix["maxlines3.loss"] = np.float64(float(res2["loss"])); fix["maxlines3.errors_1"] = np.float64(res2["errors_1"])
    fix["maxlines3.errors_5"] = np.float64(res2["errors_5"])
    for i, b in enumerate(batches):
        fix[f"b{i}.images"] = b["images"]; fix[f"b{i}.labels"] = b["labels"]; fix[f"b{i}.mask"] = np.asarray(b["mask"])
        with torch.no_grad():
            model.eval()
            x = bop._prepare_batch_images(b)
            fix[f"b{i}.output"] = model.forward(x, None, np.asarray(b["mask"]).copy())["output"].numpy()
    for k, v in np_sd(model).items():
        fix["sd." + k] = v
    np.savez_compressed(os.path.join(out, "g12_tester.npz"), **fix)

    # ---- G13: reference-written checkpoint -----------------------------------------------------------
    model = build_masked(bb, hd, seed=8, perturb_seed=9)
    ckpt = os.path.join(out, "g13_reference_checkpoint.pth")
    model.save(ckpt)
    images, labels = synthetic_batch(rng, 2, 64, 24)
    mask = (rng.random((2, 8)) < 0.4).astype(np.int64); mask[0, 0] = 1
    model.eval()
    with torch.no_grad():
        res = model.forward(bop._prepare_batch_images({"images": images}), torch.from_numpy(labels), mask.copy())
    sd = model.state_dict()
    np.savez_compressed(os.path.join(out, "g13_checkpoint.npz"), images=images, labels=labels, mask=mask,
                        output=res["output"].numpy(), loss=np.float64(res["loss"].item()),
                        keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
                        dtypes=np.array([str(v.dtype) for v in sd.values()]),
                        checkpoint_path_7=np.array(R_h.get_checkpoint_path("ckpts", 7)),
                        visualization_path_7=np.array(R_h.get_visualization_path("vis", 7, "trn")))

    # ---- G14: label text file ------------------------------------------------------------------------
    save_labels = reference_function("pero_pretraining/scripts/common.py", "save_labels")
    g = np.random.default_rng(3)
    ids = ["line_000.jpg", "b/line 1".replace(" ", "_"), "c-2.png", "empty.jpg"]
    lens = [5, 12, 1, 0]
    data = {i: g.integers(0, 8192, n).tolist() for i, n in zip(ids, lens)}
    save_labels(data, os.path.join(out, "g14_labels.txt"))
    np.savez_compressed(os.path.join(out, "g14_labels.npz"), ids=np.array(ids), lens=np.array(lens),
                        labels=np.concatenate([np.asarray(data[i], dtype=np.int64) for i in ids]))

    # ---- G15: BatchCreator ---------------------------------------------------------------------------
    fix = {}
    g = np.random.default_rng(21)
    widths = (480, 512, 400, 512, 97, 256)
    lines = [g.integers(0, 256, (10, w, 3), dtype=np.uint8) for w in widths]
    lines2 = [g.integers(0, 256, (10, w, 3), dtype=np.uint8) for w in widths]
    line_labels = [g.integers(0, 4096, int(np.ceil(w / 8))).tolist() for w in widths]
    fix["widths"] = np.array(widths)
    fix["lines_flat"] = np.concatenate([l.reshape(-1) for l in lines]); fix["lines2_flat"] = np.concatenate([l.reshape(-1) for l in lines2])
    fix["labels_flat"] = np.concatenate([np.asarray(l, dtype=np.int64) for l in line_labels])
    for name, kwargs, paired, seed in (("pad", {}, True, 5), ("pad_same", {"same_left_paddings": True}, True, 6),
                                      ("single", {}, False, 7), ("crop", {"crop_width": 256, "crop_step": 8}, True, 8)):
        bc = R_dl.BatchCreator(**kwargs)
        data = [{"image": a, "image2": (b if paired else None), "labels": (None if name == "crop" else l), "image_id": f"id{i}"}
                for i, (a, b, l) in enumerate(zip(lines, lines2, line_labels))]   # crop mode: unlabeled (joint training)
        np.random.seed(seed)
        batch = bc.create_batch(data)
        fix[f"{name}.seed"] = np.int64(seed)
        for k in ("images", "images2", "image_masks", "image_masks2", "shifts", "shift_masks", "shift_masks2", "labels"):
            if batch[k] is not None:
                fix[f"{name}.{k}"] = np.asarray(batch[k])
        fix[f"{name}.ids"] = np.array(batch["ids"])
        if batch["original_images"] is not None:
            fix[f"{name}.original_images_shape"] = np.array(batch["original_images"].shape)
            fix[f"{name}.original_images_sum"] = np.int64(batch["original_images"].astype(np.int64).sum())
    np.savez_compressed(os.path.join(out, "g15_batch_creator.npz"), **fix)
    # ---- G16: VectorQuantizer training mode (EMA codebook update, autoencoders.py:225-237) ------------------------
    from pero_pretraining.models import autoencoders as R_ae
    torch.manual_seed(13)
    vq = R_ae.VectorQuantizer(64, 32, 0.25, 0.99)
    vq.train()
    g = np.random.default_rng(31)
    fix = {"codebook0": vq.embedding.weight.detach().numpy().copy(), "ema_w0": vq.ema_w.detach().numpy().copy(),
           "ema_cluster_size0": vq.ema_cluster_size.numpy().copy(), "decay": np.float64(0.99), "commitment_cost": np.float64(0.25)}
    for step in range(2):
        feats = torch.from_numpy((g.standard_normal((2, 32, 1, 150)) * (1.0 + step)).astype(np.float32)).requires_grad_(True)
        q, idx = vq(feats)
        loss = vq.calculate_loss(q, feats) + (q * q).mean()
        loss.backward()
        fix[f"s{step}.features"] = feats.detach().numpy(); fix[f"s{step}.indices"] = idx.numpy()
        fix[f"s{step}.quantized"] = q.detach().numpy(); fix[f"s{step}.loss"] = np.float64(loss.item())
        fix[f"s{step}.grad_features"] = feats.grad.numpy().copy()
        fix[f"s{step}.codebook"] = vq.embedding.weight.detach().numpy().copy()
        fix[f"s{step}.ema_w"] = vq.ema_w.detach().numpy().copy()
        fix[f"s{step}.ema_cluster_size"] = vq.ema_cluster_size.numpy().copy()
    np.savez_compressed(os.path.join(out, "g16_vq_ema.npz"), **fix)

    print({f: os.path.getsize(os.path.join(out, f)) for f in sorted(os.listdir(out)) if f.startswith(("g12", "g13", "g14", "g15", "g16"))})


if __name__ == "__main__":
    main()
