"""GPU parity of the joint-embedding path (VICReg, NT-Xent, heads, model) against reference-generated golden
vectors (g8, g9, g11) and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pero_oracle as O  # noqa: E402


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_vicreg_matches_reference_golden(golden):
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    g = golden("g8_vicreg.npz")
    x, y = cu(g["x"]).requires_grad_(True), cu(g["y"]).requires_grad_(True)
    masks = [cu(g[k]) for k in ("image_masks1", "image_masks2", "shift_masks1", "shift_masks2")]
    res = VICRegLoss()(x, y, *masks)
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        assert abs(float(res[k]) - float(g[k])) < 1e-4 * abs(float(g[k])) + 1e-7, (k, float(res[k]), float(g[k]))
    res["loss"].backward()
    assert np.abs(x.grad.cpu().numpy() - g["grad_x"]).max() < 1e-4 * np.abs(g["grad_x"]).max() + 1e-8
    assert np.abs(y.grad.cpu().numpy() - g["grad_y"]).max() < 1e-4 * np.abs(g["grad_y"]).max() + 1e-8
    res2 = VICRegLoss(variance_weight=25.0, invariance_weight=25.0, covariance_weight=1.0)(x, y, *masks)
    assert abs(float(res2["loss"]) - float(g["loss_w25_25_1"])) < 1e-4 * float(g["loss_w25_25_1"])


def test_vicreg_bf16_large_uses_fast_syrk():
    """D = 256, ~500 rows: the SYRK and its backward take the bf16 tile kernel; compare with the oracle in f64."""
    import pero_pretraining_amd as P
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    rng = np.random.default_rng(0)
    n, s, d = 4, 64, 256
    x = (rng.standard_normal((n, s, d)) * 0.8).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((n, s, d))).astype(np.float32)
    im = np.ones((n, s), np.uint8)
    im[:, :3] = 0
    sm = np.ones((n, s), np.uint8)
    sm[:, -2:] = 0
    xb, yb = torch.from_numpy(x).bfloat16(), torch.from_numpy(y).bfloat16()
    xo, yo = xb.double().requires_grad_(True), yb.double().requires_grad_(True)
    ref = O.vicreg_loss(xo, yo, im, im, sm, sm[:, ::-1].copy())
    ref["loss"].backward()
    xg, yg = xb.cuda().requires_grad_(True), yb.cuda().requires_grad_(True)
    with P.autocast(True):
        res = VICRegLoss()(xg, yg, cu(im), cu(im), cu(sm), cu(sm[:, ::-1].copy()))
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        assert abs(float(res[k]) - float(ref[k])) < 2e-2 * abs(float(ref[k])) + 1e-6, k
    res["loss"].backward()
    gx = xg.grad.float().cpu().double()
    cos = float((gx.flatten() @ xo.grad.flatten()) / (gx.norm() * xo.grad.norm()))
    assert cos > 0.995, cos


def test_ntxent_matches_reference_golden(golden):
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss
    g = golden("g9_ntxent.npz")
    x, y = cu(g["x"]).requires_grad_(True), cu(g["y"]).requires_grad_(True)
    ones = torch.ones(x.shape[:2], dtype=torch.uint8).cuda()
    res = NTXentLoss()(x, y, ones, ones, ones, ones)
    assert abs(float(res["loss"]) - float(g["loss"])) < 1e-4 * float(g["loss"])
    res["loss"].backward()
    assert np.abs(x.grad.cpu().numpy() - g["grad_x"]).max() < 1e-4 * np.abs(g["grad_x"]).max() + 1e-8
    assert np.abs(y.grad.cpu().numpy() - g["grad_y"]).max() < 1e-4 * np.abs(g["grad_y"]).max() + 1e-8
    bad = ones.clone()
    bad[:, :3] = 0
    with pytest.raises(IndexError):  # the reference raises IndexError for any non-trivial mask (g9 records it)
        NTXentLoss()(x, y, ones, ones, bad, bad.flip(1))
    assert bool(g["nontrivial_shift_mask_raises_indexerror"])


def test_ntxent_bf16_batched_fast_path():
    import pero_pretraining_amd as P
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss
    rng = np.random.default_rng(1)
    n, s, d = 3, 128, 256
    x = rng.standard_normal((n, s, d)).astype(np.float32)
    y = (x + 0.5 * rng.standard_normal((n, s, d))).astype(np.float32)
    xb, yb = torch.from_numpy(x).bfloat16(), torch.from_numpy(y).bfloat16()
    xo, yo = xb.double().requires_grad_(True), yb.double().requires_grad_(True)
    ones = np.ones((n, s), np.uint8)
    ref = O.ntxent_loss(xo, yo, ones, ones, ones, ones)
    ref["loss"].backward()
    xg, yg = xb.cuda().requires_grad_(True), yb.cuda().requires_grad_(True)
    with P.autocast(True):
        res = NTXentLoss()(xg, yg, cu(ones), cu(ones), cu(ones), cu(ones))
    assert abs(float(res["loss"]) - float(ref["loss"])) < 2e-2 * float(ref["loss"])
    res["loss"].backward()
    gx = xg.grad.float().cpu().double()
    cos = float((gx.flatten() @ xo.grad.flatten()) / (gx.norm() * xo.grad.norm()))
    assert cos > 0.99, cos


def sd_from(fix, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(fix[k]) for k in fix.files if k.startswith(prefix)}


def test_joint_model_tiny_matches_reference_golden(golden):
    from pero_pretraining_amd.joint_embedding_pretraining import model as J
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    g = golden("g11_joint_tiny.npz")
    bb = J.init_backbone({"num_blocks": 2, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    hd = J.init_head({"type": "linear", "in_features": 64, "out_features": 80})
    model = J.JointEmbeddingTransformerEncoder(bb, hd, VICRegLoss())
    model.load_state_dict(sd_from(g))
    model = model.cuda().eval()
    masks = [cu(g[k]) for k in ("image_masks1", "image_masks2", "shift_masks1", "shift_masks2")]
    res = model(cu(g["images1"]), cu(g["images2"]), *masks)
    assert np.abs(res["output1"].float().cpu().numpy() - g["output1"]).max() < 1e-4
    assert np.abs(res["output2"].float().cpu().numpy() - g["output2"]).max() < 1e-4
    for k in ("loss", "loss.variance", "loss.invariance", "loss.covariance"):
        assert abs(float(res[k]) - float(g[k])) < 1e-4 * abs(float(g[k])) + 1e-7, k
    model.zero_grad()
    res["loss"].backward()
    for k, p in model.named_parameters():
        ref = float(g["gradnorm." + k])
        assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-7, (k, float(p.grad.norm()), ref)


def test_mlp_head_matches_oracle_and_reference_layout():
    from pero_pretraining_amd.joint_embedding_pretraining.model import MLPHead, init_head
    torch.manual_seed(0)
    head = init_head({"type": "mlp", "in_dim": 64, "hidden_dim": 128, "num_layers": 3})
    assert isinstance(head, MLPHead)
    assert list(head.state_dict().keys()) == ["layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias",
                                              "layers.4.weight", "layers.4.bias"]
    x = torch.randn(2, 10, 64)
    xr = x.clone().requires_grad_(True)
    ref = head.layers(xr.reshape(20, 64)).reshape(2, 10, -1)  # torch CPU reference of the same Sequential
    ref.square().sum().backward()
    gref = {k: p.grad.clone() for k, p in head.named_parameters()}
    head.zero_grad()
    head = head.cuda()
    xg = x.cuda().requires_grad_(True)
    out = head(xg)
    assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < 1e-4
    out.square().sum().backward()
    assert np.abs(xg.grad.cpu().numpy() - xr.grad.numpy()).max() < 1e-3
    for k, p in head.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - gref[k].numpy()).max() < 1e-3 * max(1.0, float(gref[k].abs().max())), k


def test_vector_quantizer_module_matches_reference_golden(golden):
    """Config 3 tokenizer: VectorQuantizer(8192, 512) from the reference's seed recipe, indices bit-exact (outside
    recorded near-ties), quantized output bit-exact for the small codebook; k-means label helper."""
    from pero_pretraining_amd.models.autoencoders import VectorQuantizer, kmeans_labels
    g = golden("g6_quantizers.npz")
    torch.manual_seed(5)
    vq = VectorQuantizer(8192, 512, 0.25, 0.99)
    assert abs(float(vq.embedding.weight.double().sum()) - float(g["cb8192.codebook_checksum"])) < 1e-6
    assert set(vq.state_dict().keys()) == {"ema_w", "ema_cluster_size", "embedding.weight"}
    vq = vq.cuda().eval()
    q, idx = vq(cu(g["cb8192.features"]))
    near_tie = (g["cb8192.second"] - g["cb8192.best"]) < 1e-4 * np.abs(g["cb8192.best"])
    got = idx.cpu().numpy()
    assert np.array_equal(got[~near_tie], g["cb8192.indices"][~near_tie])
    assert q.shape == g["cb8192.features"].shape
    same = got == g["cb8192.indices"]
    qs = q[:, :8, :, :8].cpu().numpy()
    ok_cols = same.reshape(2, -1)[:, :8]
    assert np.array_equal(qs[np.broadcast_to(ok_cols[:, None, None, :], qs.shape)],
                          g["cb8192.quantized_sample"][np.broadcast_to(ok_cols[:, None, None, :], qs.shape)])
    torch.manual_seed(5)
    small = VectorQuantizer(64, 32, 0.25, 0.99)
    small.embedding.weight.data.copy_(torch.from_numpy(g["small.codebook"]))
    small = small.cuda().eval()
    q2, idx2 = small(cu(g["small.features"]))
    assert np.array_equal(idx2.cpu().numpy(), g["small.indices"])
    assert np.array_equal(q2[:, :8, :, :8].cpu().numpy(), g["small.quantized_sample"])
    km = kmeans_labels(cu(g["small.features"][:, :, 0, :]), cu(g["small.codebook"]))
    assert np.array_equal(km.cpu().numpy().reshape(-1), g["small.kmeans_indices"])
    # training mode (EMA codebook update) is covered by tests/test_gpu_next_rows.py::test_vq_training_mode_matches_reference


def test_ntxent_cross_rank_negatives_single_process_matches_the_oracle():
    """world of one: the pooled embeddings of the OTHER lines of the same batch are the negatives (f32 parity mode and bf16)."""
    import pero_pretraining_amd as P
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss
    rng = np.random.default_rng(3)
    n, s, D = 5, 32, 128
    x = rng.standard_normal((n, s, D)).astype(np.float32)
    y = (x + 0.4 * rng.standard_normal((n, s, D))).astype(np.float32)
    xo, yo = torch.from_numpy(x).double().requires_grad_(True), torch.from_numpy(y).double().requires_grad_(True)
    ref, _ = O.ntxent_cross_loss(xo, yo, n)
    ref.backward()
    ones = torch.ones((n, s), dtype=torch.uint8)
    xg, yg = cu(x).requires_grad_(True), cu(y).requires_grad_(True)
    res = NTXentLoss(cross_rank_negatives=True)(xg, yg, ones, ones, ones, ones)
    assert abs(float(res["loss"]) - float(ref)) < 1e-4 * float(ref)
    res["loss"].backward()
    assert np.abs(xg.grad.cpu().numpy() - xo.grad.numpy()).max() < 1e-4 * np.abs(xo.grad.numpy()).max() + 1e-8
    assert np.abs(yg.grad.cpu().numpy() - yo.grad.numpy()).max() < 1e-4 * np.abs(yo.grad.numpy()).max() + 1e-8
    plain = NTXentLoss()(cu(x), cu(y), ones, ones, ones, ones)["loss"]
    assert float(res["loss"]) > float(plain)        # more negatives in every normaliser
    xb, yb = cu(x).bfloat16().requires_grad_(True), cu(y).bfloat16().requires_grad_(True)
    with P.autocast(True):
        rb = NTXentLoss(cross_rank_negatives=True)(xb, yb, ones, ones, ones, ones)
    assert abs(float(rb["loss"]) - float(ref)) < 3e-2 * float(ref)
