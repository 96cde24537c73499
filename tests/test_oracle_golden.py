"""CPU: pin the oracle (oracle/pero_oracle.py) against golden vectors produced by the reference
itself (oracle/make_golden.py, run in the build container).  Tolerances: float32 arithmetic in a
different association order than torch's fused kernels -> 2e-5 absolute on O(1) activations,
1e-6 relative on losses; integer outputs exact."""
import numpy as np
import torch

from oracle import pero_oracle as O


def sd_from(fix, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(fix[k]) for k in fix.files if k.startswith(prefix)}


def test_tables(golden):
    g = golden("g1_tables.npz")
    tile = O.mask_tile()
    assert np.array_equal(tile.numpy(), g["mask_tile"])
    assert abs(float(tile.double().sum()) * 512 - float(g["mask_pattern_sum"])) < 1e-3
    pe = O.positional_table(64, 4096).numpy()
    assert np.array_equal(pe[g["pe64_row_index"]], g["pe64_rows"])


def test_lr_schedule(golden):
    g = golden("g10_lr.npz")
    for it, lr in zip(g["iterations"], g["lr"]):
        assert O.warmup_lr(int(it), 2e-4, 10000, 1) == lr
    for it, lr in zip(g["iterations2"], g["lr2"]):
        assert O.warmup_lr(int(it), 1e-3, 100, 2) == lr


def test_masked_tiny_forward_eval_train(golden):
    g = golden("g4_masked_tiny.npz")
    sd = sd_from(g)
    x = O.prepare_images(torch.from_numpy(g["images"]))
    labels = torch.from_numpy(g["labels"])
    mask = g["mask"]
    # masking semantic (in-place overwrite in the reference)
    xm = O.apply_mask(x, mask, O.mask_tile())
    assert np.array_equal(xm[:, :, :, :64].numpy(), g["masked_images_sample"])
    out, loss = O.masked_model_forward(sd, x, labels, mask, 4)
    assert np.abs(out.numpy() - g["eval_output"]).max() < 2e-5
    assert abs(float(loss) - float(g["eval_loss"])) < 1e-6 * abs(float(g["eval_loss"])) + 1e-6
    tok = O.backbone_tokens(sd, x, 4, mask)
    assert np.abs(tok.reshape(3, 16, 64).permute(0, 2, 1).numpy() - g["backbone_eval"]).max() < 2e-5
    out_nm, _ = O.masked_model_forward(sd, x, None, None, 4)
    assert np.abs(out_nm.numpy() - g["eval_output_nomask"]).max() < 2e-5
    out_t, loss_t = O.masked_model_forward(sd, x, labels, mask, 4, offsets=g["train_offsets"])
    assert np.abs(out_t.numpy() - g["train_output"]).max() < 2e-5
    assert abs(float(loss_t) - float(g["train_loss"])) < 2e-6 * abs(float(g["train_loss"]))
    lw = O.masked_cross_entropy(torch.from_numpy(g["train_output"]), labels, mask, unmasked_weight=0.25)
    assert abs(float(lw) - float(g["train_loss_unmasked_w025"])) < 2e-6 * float(g["train_loss_unmasked_w025"])


def test_masked_tiny_gradients(golden):
    g = golden("g4_masked_tiny.npz")
    sd = {k: v.requires_grad_(True) for k, v in sd_from(g).items()}
    x = O.prepare_images(torch.from_numpy(g["images"]))
    _, loss = O.masked_model_forward(sd, x, torch.from_numpy(g["labels"]), g["mask"], 4)
    grads = torch.autograd.grad(loss, list(sd.values()))
    for (k, _), gr in zip(sd.items(), grads):
        ref = g["grad." + k]
        assert np.abs(gr.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k


def test_trajectory_three_steps(golden):
    g = golden("g5_trajectory.npz")
    orc = O.MaskedStepOracle(sd_from(g, "sd0."), 4)
    for i in range(3):
        loss = orc.step(g["images"][i], g["labels"][i], g["mask"][i], float(g["lr"][i]), g["offsets"][i])
        assert abs(loss - float(g["loss"][i])) < 2e-5 * abs(float(g["loss"][i])), (i, loss, g["loss"][i])
    for k, v in sd_from(g, "sd3.").items():
        got, ref = orc.sd[k].detach().numpy(), v.numpy()
        if k.endswith("in_proj_bias"):
            # the key bias has a mathematically zero gradient (softmax is shift invariant); Adam
            # normalises its rounding-noise gradient to +-lr per step, so that slice is not comparable
            d = got.shape[0] // 3
            got, ref = np.delete(got, np.s_[d:2 * d]), np.delete(ref, np.s_[d:2 * d])
        assert np.abs(got - ref).max() < 2e-4, k


def test_quantizers(golden):
    g = golden("g6_quantizers.npz")
    # small codebook: stored
    idx, dist = O.vq_nearest(np.ascontiguousarray(g["small.features"].transpose(0, 2, 3, 1)).reshape(-1, 32), g["small.codebook"])
    assert np.array_equal(idx, g["small.indices"])
    q, idx2 = O.vq_quantize(g["small.features"], g["small.codebook"])
    assert np.array_equal(idx2, g["small.indices"])
    assert np.array_equal(q[:, :8, :, :8], g["small.quantized_sample"])
    km, _ = O.kmeans_assign(np.ascontiguousarray(g["small.features"].transpose(0, 2, 3, 1)).reshape(-1, 32), g["small.codebook"])
    assert np.array_equal(km, g["small.kmeans_indices"])
    # 8192 x 512 codebook from the seed recipe (autoencoders.py:177-180: Embedding init then normal_())
    torch.manual_seed(5)
    w = torch.nn.Embedding(8192, 512).weight.data
    w.normal_()
    assert abs(float(w.double().sum()) - float(g["cb8192.codebook_checksum"])) < 1e-6
    assert np.array_equal(w[:4, :8].numpy(), g["cb8192.codebook_head"])
    flat = np.ascontiguousarray(g["cb8192.features"].transpose(0, 2, 3, 1)).reshape(-1, 512)
    idx, dist = O.vq_nearest(flat, w.numpy())
    best, second = O.margins(dist)
    near_tie = (second - best) < 1e-4 * np.abs(best)
    assert np.array_equal(idx[~near_tie], g["cb8192.indices"][~near_tie])
    assert near_tie.sum() <= 8  # rows whose best two codes are within 1e-4 relative: not comparable across BLAS orders
    km, _ = O.kmeans_assign(flat[:64], w.numpy())
    assert np.array_equal(km, g["cb8192.kmeans_indices"][:64])


def test_vicreg(golden):
    g = golden("g8_vicreg.npz")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = torch.from_numpy(g["y"]).requires_grad_(True)
    masks = [g[k] for k in ("image_masks1", "image_masks2", "shift_masks1", "shift_masks2")]
    assert set(np.unique(masks[2])) == {0, 1, 2}
    res = O.vicreg_loss(x, y, *masks)
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        assert abs(float(res[k]) - float(g[k])) < 2e-6 * abs(float(g[k])) + 1e-7, k
    gx, gy = torch.autograd.grad(res["loss"], [x, y])
    assert np.abs(gx.numpy() - g["grad_x"]).max() < 1e-6
    assert np.abs(gy.numpy() - g["grad_y"]).max() < 1e-6
    res2 = O.vicreg_loss(x, y, *masks, variance_weight=25.0, invariance_weight=25.0)
    assert abs(float(res2["loss"]) - float(g["loss_w25_25_1"])) < 2e-6 * float(g["loss_w25_25_1"])


def test_ntxent(golden):
    g = golden("g9_ntxent.npz")
    assert bool(g["nontrivial_shift_mask_raises_indexerror"])
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = torch.from_numpy(g["y"]).requires_grad_(True)
    ones = np.ones(x.shape[:2], np.uint8)
    res = O.ntxent_loss(x, y, ones, ones, ones, ones)
    assert abs(float(res["loss"]) - float(g["loss"])) < 2e-6 * float(g["loss"])
    gx, gy = torch.autograd.grad(res["loss"], [x, y])
    assert np.abs(gx.numpy() - g["grad_x"]).max() < 1e-6
    assert np.abs(gy.numpy() - g["grad_y"]).max() < 1e-6


def test_joint_tiny(golden):
    g = golden("g11_joint_tiny.npz")
    sd = sd_from(g)
    x1 = O.prepare_images(torch.from_numpy(g["images1"]))
    x2 = O.prepare_images(torch.from_numpy(g["images2"]))
    masks = tuple(g[k] for k in ("image_masks1", "image_masks2", "shift_masks1", "shift_masks2"))
    o1, o2, res = O.joint_model_forward(sd, x1, x2, masks, 4)
    assert np.abs(o1.numpy() - g["output1"]).max() < 2e-5
    assert np.abs(o2.numpy() - g["output2"]).max() < 2e-5
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        assert abs(float(res[k]) - float(g[k])) < 5e-6 * abs(float(g[k])) + 1e-7, k


def test_vqvae_quantize_oracle_matches_reference(golden):
    """g20: the reference's own VQVAE.quantize (1x1 projection -> VectorQuantizer -> 1x1 projection, models/autoencoders.py:142-146)."""
    g = golden("g20_vqvae_quantize.npz")
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    rows, labels, tok = O.vqvae_quantize(g["features"], sd)
    want = torch.from_numpy(g["projected"]).permute(0, 2, 3, 1).reshape(-1, int(g["embeddings_dim"]))
    assert float((rows - want).abs().max()) < 1e-5
    assert np.array_equal(labels.numpy(), g["labels"])
    assert float((tok - torch.from_numpy(g["tokens"])).abs().max()) < 1e-5
