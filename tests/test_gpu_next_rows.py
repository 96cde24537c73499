"""GPU parity of the SURVEY.md section 8(f) "next" rows (evaluation, checkpoints, label production, batch collation)
against the reference's own outputs (tests/golden g12-g15) and the CPU oracle.  Integer results are bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pero_oracle as O  # noqa: E402


def sd_from(fix, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(fix[k]) for k in fix.files if k.startswith(prefix)}


def build_masked32(sd=None):
    from pero_pretraining_amd.masked_pretraining import model as M
    bb = M.init_backbone({"type": "vit", "num_blocks": 2, "model_dim": 32, "num_heads": 4, "feedforward_dim": 64})
    hd = M.init_head({"type": "linear", "in_features": 32, "out_features": 24})
    model = M.MaskedTransformerEncoder(bb, hd)
    if sd is not None:
        model.load_state_dict(sd)
    return model.cuda()


# ---- (f1) evaluation -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,V", [(37, 24), (300, 4096), (5, 1)])
def test_label_rank_kernel_bit_exact(dtype, rows, V):
    from pero_pretraining_amd import ops
    g = np.random.default_rng(rows * 7 + V)
    lg = torch.from_numpy(g.standard_normal((rows, V)).astype(np.float32)).to(dtype)
    if V > 4:  # exact ties on purpose (quantise a few rows hard)
        lg[::3] = (lg[::3] * 2).round() / 2
    labels = g.integers(0, V, rows)
    mask = (g.random(rows) < 0.6).astype(np.int64)
    mask[0] = 1
    ks = [1, 2, 3, 10]
    counters = torch.zeros(1 + len(ks), dtype=torch.int64, device="cuda")
    kt = torch.tensor(ks, dtype=torch.int32, device="cuda")
    lab_t, msk_t = torch.from_numpy(labels).cuda(), torch.from_numpy(mask).cuda()
    _, ranks = ops.label_rank(lg.cuda(), lab_t, msk_t, kt, counters, want_ranks=True)
    ops.label_rank(lg.cuda(), lab_t, msk_t, kt, counters)  # accumulates: second pass doubles every counter
    ref = O.label_ranks(lg.float().numpy(), labels, mask)
    assert np.array_equal(ranks.cpu().numpy(), ref)
    exp, n = O.errors_from_ranks(ref, ks)
    assert counters.cpu().tolist() == [2 * n] + [2 * exp[f"errors_{k}"] for k in ks]


def test_masked_tester_matches_reference(golden):
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.tester import Tester
    g = golden("g12_tester.npz")
    model = build_masked32(sd_from(g))
    batches = [{"images": g[f"b{i}.images"], "labels": g[f"b{i}.labels"]} for i in range(int(g["n_batches"]))]
    bop = BatchOperator(torch.device("cuda"), float(g["masking_prob"]))
    np.random.seed(int(g["numpy_seed"]))
    tester = Tester(bop, model, batches, measured_errors=(1, 3, 10))
    res = tester.test()
    assert model.training  # test() leaves the model in train mode like the reference
    for i, b in enumerate(batches):  # same host RNG stream -> same masks as the reference run
        assert np.array_equal(b["mask"].cpu().numpy(), g[f"b{i}.mask"])
    assert abs(float(res["loss"]) - float(g["loss"])) < 1e-4 * float(g["loss"])
    for k in ("errors_1", "errors_3", "errors_10"):
        assert res[k] == float(g[k]), (k, res[k], float(g[k]))
    np.random.seed(int(g["numpy_seed"]))
    res2 = Tester(bop, model, [dict(images=b["images"], labels=b["labels"]) for b in batches], max_lines=3,
                  measured_errors=(1, 5)).test()
    assert abs(float(res2["loss"]) - float(g["maxlines3.loss"])) < 1e-4 * float(g["maxlines3.loss"])
    assert res2["errors_1"] == float(g["maxlines3.errors_1"]) and res2["errors_5"] == float(g["maxlines3.errors_5"])


def test_masked_tester_bf16_and_joint_tester(golden):
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.masked_pretraining.tester import Tester
    g = golden("g12_tester.npz")
    model = build_masked32(sd_from(g))
    batches = [{"images": g[f"b{i}.images"], "labels": g[f"b{i}.labels"]} for i in range(int(g["n_batches"]))]
    np.random.seed(int(g["numpy_seed"]))
    res = Tester(BatchOperator(torch.device("cuda"), float(g["masking_prob"])), model, batches, bfloat16=True).test()
    assert abs(float(res["loss"]) - float(g["loss"])) < 3e-2 * float(g["loss"])
    assert abs(res["errors_10"] - float(g["errors_10"])) < 0.15

    from pero_pretraining_amd.joint_embedding_pretraining import model as JM
    from pero_pretraining_amd.joint_embedding_pretraining.batch_operator import BatchOperator as JBop
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    from pero_pretraining_amd.joint_embedding_pretraining.tester import Tester as JTester
    from pero_pretraining_amd.models.transformers import VisionTransformerEncoder
    j = golden("g11_joint_tiny.npz")
    jb = VisionTransformerEncoder(num_blocks=2, model_dim=64, num_heads=4, feedforward_dim=128)
    jmodel = JM.JointEmbeddingTransformerEncoder(jb, JM.LinearHead(in_features=64, out_features=80), VICRegLoss())
    jmodel.load_state_dict(sd_from(j))
    jmodel.cuda()
    batch = {"images": j["images1"], "images2": j["images2"], "image_masks": j["image_masks1"], "image_masks2": j["image_masks2"],
             "shift_masks": j["shift_masks1"], "shift_masks2": j["shift_masks2"]}
    out = JTester(JBop(torch.device("cuda")), jmodel, [batch, batch]).test()
    assert abs(float(out["loss"]) - float(j["loss"])) < 1e-4 * abs(float(j["loss"]))
