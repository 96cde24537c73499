"""GPU parity of the drop-in modules (pero_pretraining_amd.masked_pretraining / models) against golden
vectors produced by the reference itself (tests/golden, oracle/make_golden.py) and against the CPU oracle.
Bars: f32 mode - losses within 1e-4 relative (north_star), outputs 1e-4 absolute on O(1) logits,
gradients 1e-3 relative to the tensor's max; bf16 mode - 3e-2 relative on the loss (bf16 operands)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pero_oracle as O  # noqa: E402


def sd_from(fix, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(fix[k]) for k in fix.files if k.startswith(prefix)}


def build_tiny(sd=None):
    from pero_pretraining_amd.masked_pretraining import model as M
    bb = M.init_backbone({"type": "vit", "num_blocks": 2, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    hd = M.init_head({"type": "linear", "in_features": 64, "out_features": 96})
    model = M.MaskedTransformerEncoder(bb, hd)
    if sd is not None:
        model.load_state_dict(sd)
    return model.cuda()


def test_state_dict_keys_match_reference(golden):
    g = golden("g4_masked_tiny.npz")
    model = build_tiny()
    assert list(model.state_dict().keys()) == list(sd_from(g).keys())


@pytest.mark.parametrize("input_kind", ["u8_nhwc", "f32_nchw", "batch_operator_float_images"])
def test_masked_tiny_eval_forward_backward(golden, input_kind):
    g = golden("g4_masked_tiny.npz")
    model = build_tiny(sd_from(g)).eval()
    labels = torch.from_numpy(g["labels"]).cuda()
    if input_kind == "u8_nhwc":
        x = torch.from_numpy(g["images"]).cuda()
    elif input_kind == "batch_operator_float_images":
        # BatchOperator(float_images=True) hands over the reference's `.float().permute(0, 3, 1, 2) / 255` tensor: NCHW shape,
        # NHWC strides (not contiguous) - INTEGRATION.md documents this path
        from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
        bop = BatchOperator(torch.device("cuda", 0), 0.15, float_images=True)
        x, lab2, _ = bop.prepare_batch({"images": g["images"], "labels": g["labels"]})
        assert x.dtype == torch.float32 and x.shape[1] == 3 and not x.is_contiguous() and torch.equal(lab2, labels)
    else:
        x = O.prepare_images(torch.from_numpy(g["images"])).contiguous().cuda()
    res = model(x, labels, g["mask"].copy())
    assert res["output"].shape == g["eval_output"].shape
    assert np.abs(res["output"].detach().float().cpu().numpy() - g["eval_output"]).max() < 1e-4
    assert abs(float(res["loss"]) - float(g["eval_loss"])) < 1e-4 * abs(float(g["eval_loss"]))
    if input_kind == "f32_nchw":  # reference semantic: backbone.mask overwrites the caller's tensor
        assert np.array_equal(x[:, :, :, :64].cpu().numpy(), g["masked_images_sample"])
    elif input_kind == "batch_operator_float_images":
        # same, through the non-contiguous tensor; its x / 255 ran on the GPU (torch's device division is within 1 ulp of the
        # CPU's, not bit-equal), the overwritten noise-tile columns are exact copies
        got = x[:, :, :, :64].cpu().numpy()
        assert np.abs(got - g["masked_images_sample"]).max() < 1e-7
        cols = np.repeat(g["mask"][:, :8].astype(bool), 8, axis=1)
        sel = np.broadcast_to(cols[:, None, None, :], got.shape)
        assert sel.any() and np.array_equal(got[sel], g["masked_images_sample"][sel])
    model.zero_grad()
    res["loss"].backward()
    for k, p in model.named_parameters():
        ref = g["grad." + k]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 1e-3 * max(np.abs(ref).max(), 1e-3), (k, err, np.abs(ref).max())
    # (N, d, S) backbone output, as the reference returns it
    xb = torch.from_numpy(g["images"]).cuda()
    bo = model.backbone(xb, g["mask"].copy())
    assert bo.shape == g["backbone_eval"].shape
    assert np.abs(bo.detach().float().cpu().numpy() - g["backbone_eval"]).max() < 1e-4
    out_nm = model(torch.from_numpy(g["images"]).cuda())["output"]
    assert np.abs(out_nm.detach().float().cpu().numpy() - g["eval_output_nomask"]).max() < 1e-4


def test_masked_tiny_train_mode_offsets_and_unmasked_weight(golden):
    from pero_pretraining_amd.masked_pretraining.model import MaskedCrossEntropyLoss
    g = golden("g4_masked_tiny.npz")
    model = build_tiny(sd_from(g)).train()
    x = torch.from_numpy(g["images"]).cuda()
    labels = torch.from_numpy(g["labels"]).cuda()
    model.backbone.set_offsets(g["train_offsets"])
    res = model(x, labels, g["mask"].copy())
    assert np.abs(res["output"].detach().float().cpu().numpy() - g["train_output"]).max() < 1e-4
    assert abs(float(res["loss"]) - float(g["train_loss"])) < 1e-4 * float(g["train_loss"])
    lw = MaskedCrossEntropyLoss(unmasked_weight=0.25)(res["output"], labels, torch.from_numpy(g["mask"]).cuda())
    assert abs(float(lw) - float(g["train_loss_unmasked_w025"])) < 1e-4 * float(g["train_loss_unmasked_w025"])
    # train-mode offsets drawn like the reference: same torch.randint call on the same device RNG stream
    model.backbone.set_offsets(None)
    torch.manual_seed(123)
    expect = torch.randint(0, 4096 - 16, (3,), device="cuda")
    torch.manual_seed(123)
    got = model.backbone.position_model.draw_offsets(3, 16, torch.device("cuda", 0))
    assert torch.equal(expect, got)


def test_bf16_mode_close_to_f32(golden):
    import pero_pretraining_amd as P
    g = golden("g4_masked_tiny.npz")
    model = build_tiny(sd_from(g)).eval()
    x = torch.from_numpy(g["images"]).cuda()
    labels = torch.from_numpy(g["labels"]).cuda()
    with P.autocast(True):
        res = model(x, labels, g["mask"].copy())
    assert res["output"].dtype == torch.bfloat16
    assert abs(float(res["loss"]) - float(g["eval_loss"])) < 3e-2 * float(g["eval_loss"])
    model.zero_grad()
    res["loss"].backward()
    for k, p in model.named_parameters():
        ref = g["grad." + k]
        if np.abs(ref).max() < 1e-6:
            continue
        cos = float((p.grad.cpu().double().flatten() @ torch.from_numpy(ref).double().flatten()) /
                    (p.grad.cpu().double().norm() * np.linalg.norm(ref.astype(np.float64)) + 1e-30))
        assert cos > 0.98, (k, cos)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):  # the reference Trainer's own switch
        res2 = model(x, labels, g["mask"].copy())
    assert res2["output"].dtype == torch.bfloat16 and float(res2["loss"]) == float(res["loss"])


def test_trajectory_three_steps_fused_adam(golden):
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    g = golden("g5_trajectory.npz")
    model = build_tiny(sd_from(g, "sd0.")).train()
    opt = FusedAdam(model.parameters(), lr=2e-3)
    sched = WarmupSchleduler(opt, 2e-3, 2, 1)
    trainer = Trainer(BatchOperator(torch.device("cuda", 0), 0.15), model, None, opt, sched, bfloat16=False)
    np.random.seed(5)  # the mask sequence of the reference run (BatchOperator draws from the global numpy RNG)
    for i in range(3):
        sched.update_learning_rate(i + 1)
        assert sched.current_lr == float(g["lr"][i])
        model.backbone.set_offsets(g["offsets"][i])
        batch = {"images": g["images"][i], "labels": g["labels"][i]}
        st = np.random.get_state()
        assert np.array_equal(trainer.batch_operator._create_mask(batch), g["mask"][i])
        np.random.set_state(st)
        loss = trainer.train_step(batch)
        assert abs(float(loss) - float(g["loss"][i])) < 1e-4 * float(g["loss"][i]), (i, float(loss), g["loss"][i])
    sd3 = sd_from(g, "sd3.")
    for k, v in model.state_dict().items():
        got, ref = v.cpu().numpy(), sd3[k].numpy()
        if k.endswith("in_proj_bias"):  # key-bias slice: mathematically zero gradient, Adam amplifies rounding noise
            d = got.shape[0] // 3
            got, ref = np.delete(got, np.s_[d:2 * d]), np.delete(ref, np.s_[d:2 * d])
        assert np.abs(got - ref).max() < 5e-4, k


def test_config1_seed_recipe_matches_reference(golden):
    """Config 1 (4 layers, d=256, B=8, 40x512): weights from torch.manual_seed(0) + the constructors (same RNG
    draws as the reference), loss and sampled logits against the reference's CPU run."""
    from pero_pretraining_amd.masked_pretraining import model as M
    g = golden("g4c_config1.npz")
    torch.manual_seed(0)
    bb = M.init_backbone({"type": "vit", "num_blocks": 4, "model_dim": 256, "num_heads": 4, "feedforward_dim": 1024})
    hd = M.init_head({"in_features": 256, "out_features": 4096})
    model = M.MaskedTransformerEncoder(bb, hd)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["param_names"])
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    assert np.allclose(sums, g["param_checksums"], rtol=0, atol=1e-9)
    model = model.cuda().eval()
    rng1 = np.random.default_rng(1234)
    images = rng1.integers(0, 256, (8, 40, 512, 3), dtype=np.uint8)
    labels = rng1.integers(0, 4096, (8, 64)).astype(np.int64)
    mask = (rng1.random((8, 64)) < 0.15).astype(np.int64)
    res = model(torch.from_numpy(images).cuda(), torch.from_numpy(labels).cuda(), mask)
    assert abs(float(res["loss"]) - float(g["eval_loss"])) < 1e-4 * float(g["eval_loss"])
    samples = res["output"].detach().float().cpu().numpy().reshape(-1)[g["logit_sample_index"]]
    assert np.abs(samples - g["logit_samples"]).max() < 1e-4
    model.zero_grad()
    res["loss"].backward()
    norms = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    assert np.allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    model.train()
    model.backbone.set_offsets(g["train_offsets"])
    res = model(torch.from_numpy(images).cuda(), torch.from_numpy(labels).cuda(), mask)
    assert abs(float(res["loss"]) - float(g["train_loss"])) < 1e-4 * float(g["train_loss"])


def test_head_on_masked_rows_only_gives_the_same_step(golden):
    """model.head_rows = "masked": head, loss and their backward on the masked positions alone - the loss and every
    gradient equal the all-positions path (which is pinned to the reference by g4 / g5)."""
    import copy
    from pero_pretraining_amd.masked_pretraining import model as M
    torch.manual_seed(3)
    bb = M.init_backbone({"type": "vit", "num_blocks": 2, "model_dim": 128, "num_heads": 1, "feedforward_dim": 256})
    hd = M.init_head({"type": "linear", "in_features": 128, "out_features": 4096})
    dense = M.MaskedTransformerEncoder(bb, hd).cuda().train()
    dense.head_backward = "dense"          # separate head / loss nodes, backward over every position
    sparse = copy.deepcopy(dense)
    sparse.head_rows = "masked"
    hybrid = copy.deepcopy(dense)          # the default: head forward on every position, head BACKWARD on the masked rows (_HeadCEFn)
    hybrid.head_backward = "masked"
    assert M.MaskedTransformerEncoder.head_backward == "masked" and M.MaskedTransformerEncoder.head_rows == "all"
    rng = np.random.default_rng(5)
    images = torch.from_numpy(rng.integers(0, 256, (5, 40, 256, 3), dtype=np.uint8)).cuda()
    labels = rng.integers(0, 4096, (5, 32)).astype(np.int64)
    labels[4, 20:] = -1
    mask = ((rng.random((5, 32)) < 0.2) & (labels >= 0)).astype(np.int64)
    offs = rng.integers(0, 4096 - 32, 5)
    res = {}
    row_list = torch.from_numpy(np.flatnonzero(mask.reshape(-1) == 1)).cuda()
    dev_mask = torch.from_numpy(mask).cuda()                     # exists on the device only: its row list needs a sync ...
    tagged = torch.from_numpy(mask).cuda()
    tagged._pero_host = mask.copy()                              # ... unless it carries its host original (BatchOperator / DevicePrefetcher)
    for name, model, m, rows in (("dense", dense, mask, None), ("sparse", sparse, mask, None), ("sparse_rows", sparse, dev_mask, row_list),
                                 ("sparse_tagged", sparse, tagged, None), ("sparse_dev", sparse, dev_mask, None),
                                 ("hybrid", hybrid, mask, None), ("hybrid_tagged", hybrid, tagged, None), ("hybrid_dev", hybrid, dev_mask, None)):
        model.zero_grad()
        model.backbone.set_offsets(offs)
        out = model(images, torch.from_numpy(labels).cuda(), m) if rows is None else model(images, torch.from_numpy(labels).cuda(), m, rows=rows)
        out["loss"].backward()
        torch.cuda.synchronize()
        res[name] = (float(out["loss"]), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, out)
    n = int(mask.sum())
    # head backward on the masked rows: logits of every position bit for bit, the same loss, the same gradients
    for name in ("hybrid", "hybrid_tagged", "hybrid_dev", "sparse_dev"):
        out = res[name][2]
        assert torch.equal(out["output"], res["dense"][2]["output"]), name
        assert res[name][0] == res["dense"][0], name
        if name in ("hybrid", "hybrid_tagged"):
            assert not out["output"].requires_grad        # _HeadCEFn ran (its logits are not differentiable) ...
        else:
            assert out["output"].requires_grad            # ... a device-only mask takes the dense path: no device sync for the count
        for k, g in res["dense"][1].items():
            assert (res[name][1][k] - g).abs().max() <= 1e-5 * max(float(g.abs().max()), 1e-6), (name, k)
    for name in ("sparse", "sparse_rows", "sparse_tagged"):
        out = res[name][2]
        assert out["output"] is None and out["output_rows"].shape == (n, 4096) and out["rows"].numel() == n
        assert abs(res[name][0] - res["dense"][0]) <= 1e-6 * abs(res["dense"][0])
        rows = res["dense"][2]["output"].reshape(-1, 4096)[out["rows"]]
        assert torch.equal(rows, out["output_rows"])  # the same dot products, row for row
        for k, g in res["dense"][1].items():
            assert (res[name][1][k] - g).abs().max() <= 1e-5 * max(float(g.abs().max()), 1e-6), k
    # evaluation and the unmasked-weight loss keep the all-positions path
    sparse.eval()
    assert sparse(images, torch.from_numpy(labels).cuda(), mask)["output"].shape == (5, 32, 4096)
    # nothing masked: the reference's NaN (mean over an empty selection)
    sparse.train()
    assert np.isnan(float(sparse(images, torch.from_numpy(labels).cuda(), np.zeros_like(mask))["loss"]))


def test_hip_graph_step_equals_eager_step(golden):
    """Trainer(hip_graph=True): zero_grad -> forward -> backward replayed from a captured hipGraph gives the losses and
    the weights of the eager Trainer over three optimizer steps with changing batches (eval mode: no offset draws, so
    both runs see the same inputs; the kernels and their order are the same, atomics aside)."""
    import copy
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    g = golden("g5_trajectory.npz")
    eager = build_tiny(sd_from(g, "sd0.")).eval()
    graph = copy.deepcopy(eager)
    runs = {}
    for name, model, flag in (("eager", eager, False), ("graph", graph, True)):
        opt = FusedAdam(model.parameters(), lr=2e-3)
        sched = WarmupSchleduler(opt, 2e-3, 2, 1)
        trainer = Trainer(None, model, None, opt, sched, bfloat16=True, hip_graph=flag)
        losses = []
        for i in range(3):
            sched.update_learning_rate(i + 1)
            images = torch.from_numpy(g["images"][i]).cuda()
            labels = torch.from_numpy(g["labels"][i]).cuda()
            mask = torch.from_numpy(g["mask"][i]).cuda()
            losses.append(float(trainer.train_step_prepared(images, labels, mask)))
        torch.cuda.synchronize()
        runs[name] = (losses, {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()})
        if flag:
            assert len(trainer._graphs) == 1   # one capture, three replays
    for a, b in zip(runs["eager"][0], runs["graph"][0]):
        assert np.isfinite(a) and abs(a - b) <= 1e-5 * abs(a), runs
    for k, v in runs["eager"][1].items():
        w = runs["graph"][1][k]
        if k.endswith("in_proj_bias"):  # key-bias slice: zero gradient up to rounding noise, which Adam turns into +-lr steps
            d = v.shape[0] // 3
            v, w = np.delete(v, np.s_[d:2 * d]), np.delete(w, np.s_[d:2 * d])
        assert np.abs(v - w).max() <= 1e-4, k


def test_weight_gradients_on_a_side_stream_give_the_same_gradients(golden):
    """functional.SIDE_STREAM_DW (off by default since round 3: two whole-chip kernels side by side only share the CUs): the weight / bias
    gradients of the backward pass on a second HIP stream - same loss, same gradients (the option stays covered)."""
    import copy
    from pero_pretraining_amd import functional as F
    torch.manual_seed(11)
    base = build_tiny().train()
    rng = np.random.default_rng(12)
    images = torch.from_numpy(rng.integers(0, 256, (6, 40, 256, 3), dtype=np.uint8)).cuda()
    labels = torch.from_numpy(rng.integers(0, 96, (6, 32)).astype(np.int64)).cuda()
    mask = (rng.random((6, 32)) < 0.25).astype(np.int64)
    mask[0, 0] = 1
    offs = rng.integers(0, 4096 - 32, 6)
    res = {}
    assert F.SIDE_STREAM_DW is False
    try:
        for side in (False, True):
            F.SIDE_STREAM_DW = side
            model = copy.deepcopy(base)
            model.backbone.set_offsets(offs)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(images, labels, mask)
            out["loss"].backward()
            torch.cuda.synchronize()
            res[side] = (float(out["loss"]), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    finally:
        F.SIDE_STREAM_DW = False
    assert res[True][0] == res[False][0]
    for k, g in res[False][1].items():
        assert (res[True][1][k] - g).abs().max() <= 1e-5 * max(float(g.abs().max()), 1e-6), k
